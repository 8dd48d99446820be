// atomics_scope.hip -- f64 atomic-add throughput on gfx950 by memory scope and access pattern (design input for the
// assembly scatter): build with hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics, run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int SCOPE> // 0 agent, 1 workgroup, 2 wavefront, 3 plain load+store (racy, rate only), 4 plain store
__global__ void k_atomic(double *buf, size_t mask, int per_thread, size_t stride) {
  size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (int i = 0; i < per_thread; ++i) {
    size_t idx = (t * stride + (size_t)i * 7919 * 64) & mask;
    if constexpr (SCOPE == 0) __hip_atomic_fetch_add(&buf[idx], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (SCOPE == 1) __hip_atomic_fetch_add(&buf[idx], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if constexpr (SCOPE == 2) __hip_atomic_fetch_add(&buf[idx], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    else if constexpr (SCOPE == 3) buf[idx] += 1.0;
    else buf[idx] = 1.0;
  }
}
template <class F> float timeit(F f, int reps = 3) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
  double *buf; const size_t n = (size_t)1 << 29; CK(hipMalloc(&buf, n * 8)); CK(hipMemset(buf, 0, n * 8));
  const char *names[] = {"agent", "workgroup", "wavefront", "plain rmw", "plain store"};
  for (size_t sz : {(size_t)1 << 17, (size_t)1 << 29}) for (size_t stride : {(size_t)1, (size_t)3, (size_t)9, (size_t)33}) {
    const int per = 64, nb = 256 * 16;
    float ms[5];
    ms[0] = timeit([&] { hipLaunchKernelGGL(k_atomic<0>, dim3(nb), dim3(256), 0, 0, buf, sz - 1, per, stride); });
    ms[1] = timeit([&] { hipLaunchKernelGGL(k_atomic<1>, dim3(nb), dim3(256), 0, 0, buf, sz - 1, per, stride); });
    ms[2] = timeit([&] { hipLaunchKernelGGL(k_atomic<2>, dim3(nb), dim3(256), 0, 0, buf, sz - 1, per, stride); });
    ms[3] = timeit([&] { hipLaunchKernelGGL(k_atomic<3>, dim3(nb), dim3(256), 0, 0, buf, sz - 1, per, stride); });
    ms[4] = timeit([&] { hipLaunchKernelGGL(k_atomic<4>, dim3(nb), dim3(256), 0, 0, buf, sz - 1, per, stride); });
    printf("footprint %5zu MiB stride %2zu:", sz * 8 >> 20, stride);
    for (int k = 0; k < 5; ++k) printf("  %s %.1f G/s", names[k], (double)nb * 256 * per / ms[k] / 1e6);
    printf("\n");
  }
  return 0;
}
