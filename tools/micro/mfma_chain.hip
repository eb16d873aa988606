// v_mfma_f32_32x32x16_bf16 throughput by HIP events, every CU busy: NCH accumulator chains per wave (1 = every MFMA depends on
// the one before) with one and with two waves per SIMD.  Measured on MI355X: one wave per SIMD reaches 2170 - 2205 TFLOP/s with
// dependent or independent chains alike (15.2 ns = 32 cycles at 2.1 GHz per MFMA), two waves share the pipe (27.9 ns per MFMA
// per wave, 2405 TFLOP/s together).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NCH>
__global__ __launch_bounds__(512) void chains(float* out, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 1e-3f); y[i] = (__bf16)1.0f; }
  for (int i = 0; i < iters; ++i) {
#define M(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
    if (NCH == 1) { M(a0) M(a0) M(a0) M(a0) }
    else if (NCH == 2) { M(a0) M(a1) M(a0) M(a1) }
    else { M(a0) M(a1) M(a2) M(a3) }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define THR(NCH, TH) { const int it2 = 40000; chains<NCH><<<256, TH>>>(out, it2); hipEventRecord(e0); chains<NCH><<<256, TH>>>(out, it2); hipEventRecord(e1); \
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
    printf("%d chain(s), %d wave(s) per SIMD: %.3f ms -> %.0f TFLOP/s bf16 (%.1f ns per MFMA per wave)\n", NCH, TH / 256, ms, 256.0 * (TH / 64) * 4.0 * it2 * 32768.0 / (ms * 1e-3) * 1e-12, ms * 1e6 / (4.0 * it2)); }
  THR(1, 256) THR(2, 256) THR(4, 256) THR(1, 512) THR(2, 512) THR(4, 512)
  return 0;
}
