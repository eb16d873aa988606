// How many independent VALU instructions hide in the shadow of one fp32 MFMA?  One wave per SIMD (256 threads per CU),
// a loop of 4 independent v_mfma_f32_32x32x2_f32 with NV v_fma_f32 after each; prints cycles per MFMA by NV.
// Second experiment: TWO waves per SIMD, one issuing only MFMAs, the other only VALU: do they overlap?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV>
__global__ __launch_bounds__(256) void same_wave(float* out, long long* cyc, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#define VAL(k) if (NV > k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v##k) : "v"(y));
#define VALS VAL(0) VAL(1) VAL(2) VAL(3) VAL(4) VAL(5) VAL(6) VAL(7) \
    if (NV > 8) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(y)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v1) : "v"(y)); \
                  asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v2) : "v"(y)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v3) : "v"(y)); } \
    if (NV > 12) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v4) : "v"(y)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v5) : "v"(y)); \
                   asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v6) : "v"(y)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v7) : "v"(y)); }
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y)); VALS
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y)); VALS
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y)); VALS
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(x), "v"(y)); VALS
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 512 threads: waves 0-3 MFMA only, waves 4-7 VALU only (NVW valu per "slot"); both loop `iters` times
template <int MODE>   // 0: both, 1: MFMA waves only work, 2: VALU waves only work
__global__ __launch_bounds__(512) void two_waves(float* out, long long* cyc, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3;
  const bool mf = threadIdx.x < 256;
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (mf) {
    if (MODE != 2)
      for (int i = 0; i < iters; ++i) {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y));
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(x), "v"(y));
      }
  } else {
    if (MODE != 1)
      for (int i = 0; i < iters; ++i) {     // 32 VALU per iteration (= 8 per MFMA of the other wave)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(y)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v1) : "v"(y));
          asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v2) : "v"(y)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v3) : "v"(y));
        }
      }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 255) == 0) cyc[blockIdx.x * 2 + (mf ? 0 : 1)] = t1 - t0;
  float s = v0 + v1 + v2 + v3;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 512 * 8);
  long long h[512];
  const int iters = 4000;
#define RUN(NV) { same_wave<NV><<<256, 256>>>(out, cyc, iters); same_wave<NV><<<256, 256>>>(out, cyc, iters); hipDeviceSynchronize(); \
    hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost); printf("same wave: %2d VALU per MFMA: %.1f cycles per MFMA\n", NV, (double)h[7] / (4.0 * iters)); }
  RUN(0) RUN(2) RUN(4) RUN(6) RUN(8) RUN(12) RUN(16)
#define RUN2(MODE, what) { two_waves<MODE><<<256, 512>>>(out, cyc, iters); two_waves<MODE><<<256, 512>>>(out, cyc, iters); hipDeviceSynchronize(); \
    hipMemcpy(h, cyc, 512 * 8, hipMemcpyDeviceToHost); printf("two waves per SIMD, %s: MFMA wave %.1f cycles per MFMA, VALU wave %.1f cycles per 8 VALU\n", what, (double)h[14] / (4.0 * iters), (double)h[15] / (4.0 * iters)); }
  RUN2(1, "MFMA wave alone") RUN2(2, "VALU wave alone") RUN2(0, "both")
  return 0;
}
