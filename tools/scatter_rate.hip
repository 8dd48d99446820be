// scatter_rate.hip -- how fast can one wavefront per cell add its 27 x (27 blocks of 72 bytes) to 27 rows of a 64 GB block-CSR value
// array, by kind of memory operation?  The assembly kernel's scatter (assemble3.hip) issues f64 atomics and sits on the memory-side
// atomic rate (~25 G requests of <= 64 bytes per second = 1.6 TB/s of values).  If the cells of one launch never share a row (a cell
// colouring), the same update is a plain read-modify-write.  Rates here decide whether that is worth building.
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/scatter_rate.hip -o tools/scatter_rate && tools/scatter_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int ROW = 1125; // doubles per matrix row (125 blocks of 9)
constexpr int SPAN = 243; // doubles one cell adds to a row (27 blocks), contiguous here: the best case of the row order

__device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// MODE 0: f64 atomic add, 8 bytes per lane   1: plain store 8 B / lane   2: load + add + store 8 B / lane
//      3: load + add + store 16 B / lane     4: plain store 16 B / lane  5: as 3 with the loads of the next row issued before the stores
// LOCAL: rows of a cell are neighbours (8 apart) instead of random
template <int MODE, bool LOCAL>
__global__ __launch_bounds__(128) void k_scatter(double *buf, unsigned n_rows, unsigned n_cells) {
  const unsigned cell = blockIdx.x * 2 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (cell >= n_cells) return;
  const double v = 1.0 + lane;
  d2 nxt = {0, 0};
  for (int a = 0; a < 27; ++a) {
    const unsigned h = hash32(cell * 27u + a);
    const unsigned row = LOCAL ? (hash32(cell) % (n_rows - 256) + 8u * a + (h & 7u)) : h % n_rows;
    const unsigned off = (hash32(h) % 98u) * 9u; // first block of the cell in the row
    double *p = buf + size_t(row) * ROW + off;
    if constexpr (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (64 * k + lane < SPAN) unsafeAtomicAdd(p + 64 * k + lane, v);
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (64 * k + lane < SPAN) p[64 * k + lane] = v;
    } else if constexpr (MODE == 2) {
      double o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (64 * k + lane < SPAN) ? p[64 * k + lane] : 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) if (64 * k + lane < SPAN) p[64 * k + lane] = o[k] + v;
    } else if constexpr (MODE == 3) {
      d2 o[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) o[k] = (128 * k + 2 * lane < SPAN + 1) ? *reinterpret_cast<const d2 *>(p + 128 * k + 2 * lane) : d2{0, 0};
#pragma unroll
      for (int k = 0; k < 2; ++k) if (128 * k + 2 * lane < SPAN + 1) *reinterpret_cast<d2 *>(p + 128 * k + 2 * lane) = o[k] + d2{v, v};
    } else if constexpr (MODE == 4) {
#pragma unroll
      for (int k = 0; k < 2; ++k) if (128 * k + 2 * lane < SPAN + 1) *reinterpret_cast<d2 *>(p + 128 * k + 2 * lane) = d2{v, v};
    } else {
      // software pipeline over the rows: the values of row a + 1 are requested before the stores of row a
      d2 o[2];
      if (a == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) o[k] = (128 * k + 2 * lane < SPAN + 1) ? *reinterpret_cast<const d2 *>(p + 128 * k + 2 * lane) : d2{0, 0};
      } else { o[0] = nxt; o[1] = d2{0, 0}; if (128 + 2 * lane < SPAN + 1) o[1] = *reinterpret_cast<const d2 *>(p + 128 + 2 * lane); }
      if (a + 1 < 27) {
        const unsigned h2 = hash32(cell * 27u + a + 1);
        const unsigned row2 = LOCAL ? (hash32(cell) % (n_rows - 256) + 8u * (a + 1) + (h2 & 7u)) : h2 % n_rows;
        const double *q = buf + size_t(row2) * ROW + (hash32(h2) % 98u) * 9u;
        nxt = *reinterpret_cast<const d2 *>(q + 2 * lane);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) if (128 * k + 2 * lane < SPAN + 1) *reinterpret_cast<d2 *>(p + 128 * k + 2 * lane) = o[k] + d2{v, v};
    }
  }
}

template <class F> float timeit(F f, int reps = 2) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

template <int MODE, bool LOCAL> void run(double *buf, unsigned n_rows, unsigned n_cells, const char *name) {
  const float ms = timeit([&] { hipLaunchKernelGGL((k_scatter<MODE, LOCAL>), dim3((n_cells + 1) / 2), dim3(128), 0, 0, buf, n_rows, n_cells); });
  const double bytes = double(n_cells) * 27 * SPAN * 8;
  printf("%-44s %-7s rows: %8.2f ms for %u cells = %6.2f TB/s of values, a 128^3 mesh (2.1 M cells) in %6.1f ms\n", name, LOCAL ? "nearby" : "random", ms, n_cells,
         bytes / ms / 1e9, ms * 2097152.0 / n_cells);
}

int main() {
  const unsigned n_rows = 7000000; // x 9000 bytes = 63 GB
  double *buf; CK(hipMalloc(&buf, size_t(n_rows) * ROW * 8)); CK(hipMemset(buf, 0, size_t(n_rows) * ROW * 8));
  const unsigned n_cells = 1u << 20;
  run<0, false>(buf, n_rows, n_cells, "f64 atomics, 8 bytes per lane");
  run<1, false>(buf, n_rows, n_cells, "plain stores, 8 bytes per lane");
  run<4, false>(buf, n_rows, n_cells, "plain stores, 16 bytes per lane");
  run<2, false>(buf, n_rows, n_cells, "load + add + store, 8 bytes per lane");
  run<3, false>(buf, n_rows, n_cells, "load + add + store, 16 bytes per lane");
  run<5, false>(buf, n_rows, n_cells, "... next row's loads before the stores");
  run<0, true>(buf, n_rows, n_cells, "f64 atomics, 8 bytes per lane");
  run<1, true>(buf, n_rows, n_cells, "plain stores, 8 bytes per lane");
  run<4, true>(buf, n_rows, n_cells, "plain stores, 16 bytes per lane");
  run<2, true>(buf, n_rows, n_cells, "load + add + store, 8 bytes per lane");
  run<3, true>(buf, n_rows, n_cells, "load + add + store, 16 bytes per lane");
  run<5, true>(buf, n_rows, n_cells, "... next row's loads before the stores");
  return 0;
}
