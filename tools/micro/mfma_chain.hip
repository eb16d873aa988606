// v_mfma_f32_32x32x16_bf16 issue rate: NCH independent accumulator chains per wave (1 = every MFMA depends on the one before),
// with one wave per SIMD and with two (the second wave of a co-resident workgroup).  Cycles are s_memtime (100 MHz) ticks scaled
// by the measured clock ratio of a calibration loop of dependent v_fma (4 cycles each).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NCH>
__global__ __launch_bounds__(512) void chains(float* out, long long* cyc, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 1e-3f); y[i] = (__bf16)1.0f; }
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#define M(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
    if (NCH == 1) { M(a0) M(a0) M(a0) M(a0) }
    else if (NCH == 2) { M(a0) M(a1) M(a0) M(a1) }
    else { M(a0) M(a1) M(a2) M(a3) }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void calib(float* out, long long* cyc, int iters) {
  float v = threadIdx.x, y = 1.0001f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(y)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(y));
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(y)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(y));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = v;
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 4096 * 8);
  long long h[4096];
  const int iters = 4000;
  calib<<<256, 256>>>(out, cyc, iters); calib<<<256, 256>>>(out, cyc, iters); hipDeviceSynchronize();
  hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
  const double tick = 16.0 * iters / (double)h[0];       // core cycles per s_memtime tick (4 dependent v_fma = 16+ cycles)
  printf("calibration: %.2f core cycles per tick (if a dependent v_fma_f32 is 4 cycles)\n", tick);
#define RUN(NCH, TH) { chains<NCH><<<256, TH>>>(out, cyc, iters); chains<NCH><<<256, TH>>>(out, cyc, iters); hipDeviceSynchronize(); \
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost); printf("%d chain(s), %d wave(s) per SIMD: %.1f ticks*ratio = %.1f cycles per MFMA per wave\n", NCH, TH / 256, (double)h[0] / (4.0 * iters), (double)h[0] / (4.0 * iters) * tick); }
  RUN(1, 256) RUN(2, 256) RUN(4, 256) RUN(1, 512) RUN(2, 512) RUN(4, 512)
  // whole-launch throughput by HIP events (every CU busy): bf16 TFLOP/s with one and with two waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define THR(NCH, TH) { const int it2 = 40000; chains<NCH><<<256, TH>>>(out, cyc, it2); hipEventRecord(e0); chains<NCH><<<256, TH>>>(out, cyc, it2); hipEventRecord(e1); \
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
    printf("%d chain(s), %d wave(s) per SIMD: %.3f ms -> %.0f TFLOP/s bf16 (%.1f ns per MFMA per wave)\n", NCH, TH / 256, ms, 256.0 * (TH / 64) * 4.0 * it2 * 32768.0 / (ms * 1e-3) * 1e-12, ms * 1e6 / (4.0 * it2)); }
  THR(1, 256) THR(4, 256) THR(1, 512) THR(4, 512)
  return 0;
}
