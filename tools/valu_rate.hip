// valu_rate.hip -- issue rate of the vector ALU on gfx950 per instruction kind and waves per SIMD: cycles per wave instruction on
// one SIMD (2.4 GHz).  Decides what "packing" (v_pk_fma_f32) and single precision buy in an issue-bound kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
// KIND 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_fma_f64, 3 v_add_u32, 4 v_mul_f32, 5 v_pk_mul_f32, 6 v_lshl_add_u32, 7 v_cndmask_b32
template <int KIND>
__global__ void k_rate(double *out, int iters) {
  float g[16]; f2 p[16]; double d[16]; unsigned u[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { g[k] = k + threadIdx.x; p[k] = f2{float(k), float(threadIdx.x)}; d[k] = k + threadIdx.x; u[k] = k * 977u + threadIdx.x; }
  const float m = 1.0000001f, a = 0.5f; const f2 pm = {m, m}, pa = {a, a}; const double dm = 1.0000001, da = 0.5;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(g[k]) : "v"(m), "v"(a));
      else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pm), "v"(pa));
      else if (KIND == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(dm), "v"(da));
      else if (KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 5) & 15]));
      else if (KIND == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(g[k]) : "v"(m));
      else if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pm));
      else if (KIND == 6) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[k]) : "v"(u[(k + 5) & 15]));
      else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 5) & 15]));
    }
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += g[k] + p[k].x + p[k].y + d[k] + u[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(double *out, const char *name) {
  const int iters = 20000;
  printf("%-16s", name);
  for (int wps : {1, 2, 4, 8}) { // waves per SIMD: blocks of 256 threads = one wave per SIMD of a CU, wps blocks per CU
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rate<KIND>), dim3(256 * wps), dim3(256), 0, 0, out, 100); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_rate<KIND>), dim3(256 * wps), dim3(256), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // cycles per wave instruction on one SIMD: time * clock / (instructions issued on that SIMD)
    printf("  %d w/SIMD: %5.2f cyc/instr", wps, ms * 1e-3 * 2.4e9 / (double(iters) * 16 * wps));
  }
  printf("\n");
}
int main() {
  double *out; CK(hipMalloc(&out, size_t(256) * 8 * 256 * 8));
  run<0>(out, "v_fma_f32"); run<1>(out, "v_pk_fma_f32"); run<4>(out, "v_mul_f32"); run<5>(out, "v_pk_mul_f32"); run<2>(out, "v_fma_f64");
  run<3>(out, "v_add_u32"); run<6>(out, "v_lshl_add_u32"); run<7>(out, "v_cndmask_b32");
  return 0;
}
