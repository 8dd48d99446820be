// overlap_mfma.hip -- does v_mfma_f64_16x16x4 on gfx950 run beside other VALU work, inside a wave and across the waves of a SIMD?
// The assembly kernel issues ~6000 VALU instructions and 322 MFMAs (64 cycles each) per wave; whether its floor is the sum or
// the maximum of the two decides what is worth removing.
//   1. one wave per SIMD: 4 independent MFMAs per trip, NV independent VALU instructions of a kind after every MFMA
//   2. two waves per SIMD: waves 0-3 of a block run the MFMA loop, waves 4-7 a VALU loop (same SIMDs: wave w and w + 4)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

// KIND 0: v_fma_f64, 1: v_fma_f32, 2: v_add_u32 (+ xor so that it is not folded), 3: ds_read_b64 (LDS)
template <int NV, int KIND>
__global__ __launch_bounds__(256) void k_mix(double *out, int iters) {
  __shared__ double lds[512];
  lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 256] = 1.0;
  __syncthreads();
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  double f[16]; float g[16]; unsigned u[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { f[k] = k + threadIdx.x; g[k] = float(k) + threadIdx.x; u[k] = k * 977u + threadIdx.x; }
  const double m = 1.0000001, ad = 0.5;
  for (int i = 0; i < iters; ++i) {
#define VAL(base)                                                                                                   \
    _Pragma("unroll") for (int k = 0; k < NV; ++k) {                                                                   \
      const int j = (base + k) & 15;                                                                                   \
      if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[j]) : "v"(m), "v"(ad));                          \
      else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(g[j]) : "v"(float(m)), "v"(float(ad)));       \
      else if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 15]));                    \
      else { double t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((unsigned)((threadIdx.x + j) & 255) * 8u)); f[j] = t; } \
    }
    asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b)); VAL(0)
    asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b)); VAL(4)
    asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b)); VAL(8)
    asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b)); VAL(12)
    if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)");
#undef VAL
  }
  double s = c0[0] + c1[1] + c2[2] + c3[3];
#pragma unroll
  for (int k = 0; k < 16; ++k) s += f[k] + g[k] + u[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MODE bit 0: the first four waves of the block run MFMAs; bit 1: the last four waves run v_fma_f64 (KIND 0) / v_add_u32 (KIND 2)
template <int KIND>
__global__ __launch_bounds__(512) void k_two(double *out, int iters, int mode, int nv) {
  const int wave = threadIdx.x >> 6;
  double s = 0;
  if (wave < 4) {
    if (mode & 1) {
      d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
      double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
      for (int i = 0; i < iters; ++i) {
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
      }
      s = c0[0] + c1[1] + c2[2] + c3[3];
    }
  } else if (mode & 2) {
    double f[8]; unsigned u[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { f[k] = k + threadIdx.x; u[k] = k * 977u + threadIdx.x; }
    const double m = 1.0000001, ad = 0.5;
    for (int i = 0; i < iters * nv; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[k]) : "v"(m), "v"(ad));
        else asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) s += f[k] + u[k];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F> float timeit(F f, int reps = 3) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}
template <int NV, int KIND> void mix(double *out, const char *kind) {
  const int iters = 20000;
  float ms = timeit([&] { hipLaunchKernelGGL((k_mix<NV, KIND>), dim3(256), dim3(256), 0, 0, out, iters); });
  // cycles per trip of one wave at 2.4 GHz: 4 MFMAs = 256 cycles when nothing else costs
  printf("1 wave/SIMD, 4 MFMA + 4x%2d %-10s per trip: %7.3f ms = %6.1f cycles per trip\n", NV, kind, ms, ms * 1e-3 * 2.4e9 / iters);
}
int main() {
  double *out; CK(hipMalloc(&out, 256 * 512 * 8));
  mix<0, 0>(out, "-");
  mix<2, 0>(out, "v_fma_f64"); mix<4, 0>(out, "v_fma_f64"); mix<8, 0>(out, "v_fma_f64"); mix<12, 0>(out, "v_fma_f64"); mix<16, 0>(out, "v_fma_f64");
  mix<4, 1>(out, "v_fma_f32"); mix<8, 1>(out, "v_fma_f32"); mix<12, 1>(out, "v_fma_f32"); mix<16, 1>(out, "v_fma_f32");
  mix<4, 2>(out, "v_add_u32"); mix<8, 2>(out, "v_add_u32"); mix<12, 2>(out, "v_add_u32"); mix<16, 2>(out, "v_add_u32");
  mix<2, 3>(out, "ds_read_b64"); mix<4, 3>(out, "ds_read_b64"); mix<8, 3>(out, "ds_read_b64");
  const int iters = 20000;
  for (int nv : {4, 8}) {
    for (int mode = 1; mode <= 3; ++mode) {
      float ms = timeit([&] { hipLaunchKernelGGL((k_two<0>), dim3(256), dim3(512), 0, 0, out, iters, mode, nv); });
      printf("2 waves/SIMD f64  nv %d mode %d (1 MFMA wave, 2 VALU wave [%d v_fma_f64 per trip], 3 both): %7.3f ms\n", nv, mode, 8 * nv, ms);
    }
    for (int mode = 2; mode <= 3; ++mode) {
      float ms = timeit([&] { hipLaunchKernelGGL((k_two<2>), dim3(256), dim3(512), 0, 0, out, iters, mode, nv); });
      printf("2 waves/SIMD u32  nv %d mode %d (2 VALU wave [%d v_add_u32 per trip], 3 both): %7.3f ms\n", nv, mode, 8 * nv, ms);
    }
  }
  return 0;
}
