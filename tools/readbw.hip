// readbw.hip -- read-only streaming bandwidth of the device (what a perfect SpMV of a stored matrix could reach):
// hipcc -O3 --offload-arch=gfx950 tools/readbw.hip -o tools/readbw
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
template <int U>
__global__ __launch_bounds__(256) void k_read(const v2d *__restrict__ a, size_t n, double *out) {
  v2d acc = {0, 0};
  size_t i = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x * U;
  for (; i + (U - 1) * blockDim.x < n; i += stride) {
    v2d v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = a[i + (size_t)u * blockDim.x];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc.x + acc.y == 1.2345e-300) out[0] = acc.x;
}
template <class F> float timeit(F f, int reps = 5) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
  const size_t n = (size_t)1 << 31; // 2^31 double2 = 32 GiB
  v2d *a; double *out; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&out, 64)); CK(hipMemset(a, 0, n * 16));
  for (int blocks : {256 * 8, 256 * 16, 256 * 32, 256 * 64}) {
    float m1 = timeit([&] { hipLaunchKernelGGL((k_read<1>), dim3(blocks), dim3(256), 0, 0, a, n, out); });
    float m4 = timeit([&] { hipLaunchKernelGGL((k_read<4>), dim3(blocks), dim3(256), 0, 0, a, n, out); });
    float m8 = timeit([&] { hipLaunchKernelGGL((k_read<8>), dim3(blocks), dim3(256), 0, 0, a, n, out); });
    printf("read 32 GiB, %6d blocks: 16 B/lane x1 %.2f TB/s, x4 %.2f TB/s, x8 %.2f TB/s\n", blocks, n * 16 / m1 / 1e9, n * 16 / m4 / 1e9, n * 16 / m8 / 1e9);
  }
  return 0;
}
