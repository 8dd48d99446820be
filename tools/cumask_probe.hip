// cumask_probe.hip -- does hipExtStreamCreateWithCUMask restrict a kernel to the masked CUs on this device / driver?  A compute-bound
// kernel (v_fma_f64 chains, 2048 workgroups) on streams with 256 / 128 / 64 / 32 CUs enabled: time must scale with 256 / enabled.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k_fma(double *out, int iters) {
  double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, b = 1.0000001, c = 0.5;
  for (int i = 0; i < iters; ++i) { a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c); }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
int main() {
  double *out; CK(hipMalloc(&out, 2048 * 256 * 8));
  for (uint32_t pat : {0xffffffffu, 0x55555555u, 0x11111111u, 0x01010101u}) {
    uint32_t mask[8]; for (auto &w : mask) w = pat;
    hipStream_t s; const hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    if (e != hipSuccess) { printf("pattern %08x: %s\n", pat, hipGetErrorString(e)); continue; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_fma, dim3(2048), dim3(256), 0, s, out, 1000); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); hipLaunchKernelGGL(k_fma, dim3(2048), dim3(256), 0, s, out, 40000); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("CU mask pattern %08x (%d of 32 bits per word): %.3f ms\n", pat, __builtin_popcount(pat), ms);
    CK(hipStreamDestroy(s));
  }
  return 0;
}
