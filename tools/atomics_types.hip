// atomics_types.hip -- atomic-add throughput on gfx950 by DATA TYPE (f64 / f32 / u64 / u32), access pattern, footprint
// and number of active CUs: is the 24 G segments/s of tools/atomics_scope.hip a property of the f64 atomic ALU of the L2 or
// of the atomic path as such?  (Design input for the assembly scatter: a fixed-point accumulation would use u64 adds.)
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/atomics_types.hip -o tools/atomics_types
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// stride_bytes between consecutive lanes; every element is sizeof(T) wide
template <class T>
__global__ void k_atomic(T *buf, size_t mask_bytes, int per_thread, size_t stride_bytes) {
  size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (int i = 0; i < per_thread; ++i) {
    size_t off = (t * stride_bytes + (size_t)i * 7919 * 512) & mask_bytes & ~(size_t)(sizeof(T) - 1);
    __hip_atomic_fetch_add(reinterpret_cast<T *>(reinterpret_cast<char *>(buf) + off), T(1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// the scatter's shape: a wave instruction covers 512 contiguous bytes (64 lanes x 8 bytes) starting at a multiple of 72 bytes
template <class T>
__global__ void k_atomic_rows(T *buf, size_t mask_bytes, int per_thread) {
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  for (int i = 0; i < per_thread; ++i) {
    size_t off = ((wave * 131 + (size_t)i * 7919) * 72 * 57 + lane * 8) & mask_bytes & ~(size_t)7;
    __hip_atomic_fetch_add(reinterpret_cast<T *>(reinterpret_cast<char *>(buf) + off), T(1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// the same access shape as plain load + add + store (no two waves touch the same bytes at the same time in this test), NB
// independent 512-byte rows in flight per wave
template <int NB>
__global__ void k_rmw_rows(double *buf, size_t mask_bytes, int per_thread) {
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  for (int i = 0; i < per_thread; i += NB) {
    double *p[NB], v[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      size_t off = ((wave * 131 + (size_t)(i + k) * 7919) * 72 * 57 + lane * 8) & mask_bytes & ~(size_t)7;
      p[k] = reinterpret_cast<double *>(reinterpret_cast<char *>(buf) + off);
      v[k] = __builtin_nontemporal_load(p[k]);
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) *p[k] = v[k] + 1.0;
  }
}
template <class F> float timeit(F f, int reps = 3) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
template <class T> void run(const char *name, void *buf, size_t bytes, size_t stride_bytes, int nb) {
  const int per = 64;
  float ms = timeit([&] { hipLaunchKernelGGL(k_atomic<T>, dim3(nb), dim3(256), 0, 0, (T *)buf, bytes - 1, per, stride_bytes); });
  printf("  %s %7.1f G/s", name, (double)nb * 256 * per / ms / 1e6);
}
int main() {
  void *buf; const size_t n = (size_t)1 << 32; CK(hipMalloc(&buf, n)); CK(hipMemset(buf, 0, n));
  for (int nb : {256 * 16, 64 * 16, 32 * 2})
    for (size_t bytes : {(size_t)1 << 20, (size_t)1 << 32})
      for (size_t stride : {(size_t)4, (size_t)8, (size_t)64, (size_t)72, (size_t)128, (size_t)264}) {
        printf("blocks %5d footprint %5zu MiB lane stride %3zu B:", nb, bytes >> 20, stride);
        run<double>("f64", buf, bytes, stride, nb);
        run<float>("f32", buf, bytes, stride, nb);
        run<unsigned long long>("u64", buf, bytes, stride, nb);
        run<unsigned int>("u32", buf, bytes, stride, nb);
        printf("\n");
      }
  for (size_t bytes : {(size_t)1 << 20, (size_t)1 << 32}) {
    const int per = 64, nb = 256 * 16;
    float m0 = timeit([&] { hipLaunchKernelGGL(k_atomic_rows<double>, dim3(nb), dim3(256), 0, 0, (double *)buf, bytes - 1, per); });
    float m1 = timeit([&] { hipLaunchKernelGGL(k_atomic_rows<unsigned long long>, dim3(nb), dim3(256), 0, 0, (unsigned long long *)buf, bytes - 1, per); });
    printf("512 contiguous bytes per wave instruction at 72-byte granularity, footprint %5zu MiB: f64 %.1f G/s  u64 %.1f G/s\n", bytes >> 20,
           (double)nb * 256 * per / m0 / 1e6, (double)nb * 256 * per / m1 / 1e6);
  }
  for (size_t bytes : {(size_t)1 << 20, (size_t)1 << 32}) {
    const int per = 64, nb = 256 * 16;
    float m1 = timeit([&] { hipLaunchKernelGGL(k_rmw_rows<1>, dim3(nb), dim3(256), 0, 0, (double *)buf, bytes - 1, per); });
    float m4 = timeit([&] { hipLaunchKernelGGL(k_rmw_rows<4>, dim3(nb), dim3(256), 0, 0, (double *)buf, bytes - 1, per); });
    float m8 = timeit([&] { hipLaunchKernelGGL(k_rmw_rows<8>, dim3(nb), dim3(256), 0, 0, (double *)buf, bytes - 1, per); });
    printf("plain load + add + store of such rows, footprint %5zu MiB: 1 row in flight %.1f G/s  4 rows %.1f G/s  8 rows %.1f G/s\n", bytes >> 20,
           (double)nb * 256 * per / m1 / 1e6, (double)nb * 256 * per / m4 / 1e6, (double)nb * 256 * per / m8 / 1e6);
  }
  return 0;
}
