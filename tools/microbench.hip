// microbench.hip -- gfx950 rates the design depends on: HBM copy, f64 FMA, f64 MFMA (16x16x4, 4x4x4), f64 atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_copy(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_fma(double *out, int iters) {
  double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 1.0000001, c = 0.5;
  for (int i = 0; i < iters; ++i) {
    a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c);
    a4 = fma(a4, b, c); a5 = fma(a5, b, c); a6 = fma(a6, b, c); a7 = fma(a7, b, c);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void k_mfma16(double *out, int iters) {
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void k_mfma4(double *out, int iters) {
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
}
// MFMA layout probe: D = A*B with A[i][k] = i + 100 k, B[k][j] = (k==0) * j + (k==1) ... use identity-like
__global__ void k_layout(double *out) {
  const int l = threadIdx.x;
  // A = I-like on k: A[i][k] = (i % 4 == k) ? 1 : 0 scaled by (i+1); B[k][j] = 10*k + j  => D[i][j] = (i+1) * (10*(i%4) + j)
  const int i = l & 15, k = l >> 4;
  double a = ((i & 3) == k) ? double(i + 1) : 0.0;
  double b = 10.0 * k + (l & 15);
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
__global__ void k_atomic(double *buf, size_t mask, int per_thread, size_t stride) {
  size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (int i = 0; i < per_thread; ++i) {
    size_t idx = (t * stride + (size_t)i * 7919 * 64) & mask;
    unsafeAtomicAdd(&buf[idx], 1.0);
  }
}
template <class F> float timeit(F f, int reps = 5) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz mem %.1f GB\n", p.name, p.multiProcessorCount, p.clockRate, p.totalGlobalMem / 1e9);
  size_t n = (size_t)1 << 28; // 2^28 double2 = 4 GiB
  double2 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMemset(a, 1, n * 16));
  float ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(256 * 32), dim3(256), 0, 0, a, b, n); });
  printf("copy 4GiB->4GiB: %.3f ms  %.2f TB/s (read+write)\n", ms, 2.0 * n * 16 / ms / 1e9);
  ms = timeit([&] { CK(hipMemsetAsync(b, 0, n * 16, 0)); });
  printf("memset 4GiB: %.3f ms  %.2f TB/s\n", ms, n * 16 / ms / 1e9);
  double *out; CK(hipMalloc(&out, 256 * 1024 * 8 * 8));
  const int blocks = 256 * 8, iters = 20000;
  ms = timeit([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters); });
  printf("v_fma_f64: %.2f TFLOP/s\n", 2.0 * 8 * iters * blocks * 256 / ms / 1e9);
  ms = timeit([&] { hipLaunchKernelGGL(k_mfma16, dim3(blocks), dim3(256), 0, 0, out, iters); });
  printf("mfma_f64_16x16x4 (4 acc/wave, 8 waves/SIMD): %.2f TFLOP/s\n", 2.0 * 16 * 16 * 4 * 4.0 * iters * blocks * 4 / ms / 1e9);
  ms = timeit([&] { hipLaunchKernelGGL(k_mfma16, dim3(256), dim3(256), 0, 0, out, iters); });
  printf("mfma_f64_16x16x4 (4 acc/wave, 1 wave/SIMD): %.2f TFLOP/s\n", 2.0 * 16 * 16 * 4 * 4.0 * iters * 256 * 4 / ms / 1e9);
  ms = timeit([&] { hipLaunchKernelGGL(k_mfma4, dim3(blocks), dim3(256), 0, 0, out, iters); });
  printf("mfma_f64_4x4x4_4b: %.2f TFLOP/s\n", 2.0 * 4 * 4 * 4 * 4 * 4.0 * iters * blocks * 4 / ms / 1e9);
  // layout probe
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, out); CK(hipDeviceSynchronize());
  std::vector<double> h(256); CK(hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost));
  int ok_guide = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    int col = l & 15, row = (l >> 4) + 4 * r; // guide: col = lane&15, row = (lane>>4) + 4*reg
    double expect = (row + 1) * (10.0 * (row & 3) + col);
    if (h[l * 4 + r] != expect) ok_guide = 0;
  }
  printf("f64 mfma C/D layout col=lane&15,row=(lane>>4)+4*reg with A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]: %s\n", ok_guide ? "CONFIRMED" : "MISMATCH");
  if (!ok_guide) { for (int l = 0; l < 64; l += 7) printf(" lane %d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
  // atomics
  double *buf = (double *)b;
  for (size_t sz : {(size_t)1 << 17, (size_t)1 << 24, (size_t)1 << 29}) { // 1 MiB, 128 MiB, 4 GiB of doubles
    for (size_t stride : {(size_t)1, (size_t)9}) {
      const int per = 64; const int nb = 256 * 16;
      ms = timeit([&] { hipLaunchKernelGGL(k_atomic, dim3(nb), dim3(256), 0, 0, buf, sz - 1, per, stride); }, 3);
      printf("atomicAdd f64 footprint %zu MiB stride %zu: %.2f Gatom/s\n", sz * 8 >> 20, stride, (double)nb * 256 * per / ms / 1e6);
    }
  }
  return 0;
}
