// Follow-up of launch_floor.hip: what makes a launch in a chain of REAL kernels cost 4-5 us when a chain of empty ones
// costs 1.6?  Chains of (writer, empty), (writer, reader), writers of different sizes / store kinds, 512-thread blocks.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
__global__ void k_empty(float* p, int n) { if (n < 0) p[0] = 1.f; }
template <int WT> __global__ void k_write(float* p, int n) {          // grid x 256 threads, each writes n float4
  float4* q = reinterpret_cast<float4*>(p) + (size_t)blockIdx.x * blockDim.x * n + threadIdx.x;
  for (int i = 0; i < n; ++i) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = {1.f, 2.f, 3.f, (float)i};
    if (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(q + (size_t)i * blockDim.x), "v"(v) : "memory");
    else q[(size_t)i * blockDim.x] = make_float4(v[0], v[1], v[2], v[3]);
  }
}
__global__ void k_read(const float* p, float* o, int n) {
  const float4* q = reinterpret_cast<const float4*>(p) + (size_t)blockIdx.x * blockDim.x * n + threadIdx.x;
  float s = 0.f;
  for (int i = 0; i < n; ++i) { const float4 v = q[(size_t)i * blockDim.x]; s += v.x + v.w; }
  if (s == 12345.f) o[0] = s;
}
template <class F> double timeit(int per_chain, F&& body) {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  const int reps = 100;
  for (int i = 0; i < reps; ++i) body(s);
  hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  std::vector<float> ms;
  for (int r = 0; r < 20; ++r) { hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipStreamSynchronize(s); float m; hipEventElapsedTime(&m, e0, e1); ms.push_back(m); }
  std::sort(ms.begin(), ms.end());
  hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(s);
  return ms[ms.size() / 2] * 1000.0 / (reps * per_chain);
}
int main() {
  float *buf, *o; hipMalloc(&buf, 64 << 20); hipMalloc(&o, 1024); hipMemset(buf, 0, 64 << 20);
  printf("us per launch (hipGraph chains, median)\n");
  printf("empty, empty                         : %.2f\n", timeit(2, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(32), dim3(256), 0, s, o, 1); hipLaunchKernelGGL(k_empty, dim3(32), dim3(256), 0, s, o, 1); }));
  for (int kb : {16, 128, 1024, 8192}) {
    const int grid = 32, n = kb * 1024 / (grid * 256 * 16) > 0 ? kb * 1024 / (grid * 256 * 16) : 1;
    printf("write %5d KB plain, empty           : %.2f\n", grid * 256 * 16 * n / 1024, timeit(2, [&](hipStream_t s) { hipLaunchKernelGGL(k_write<0>, dim3(grid), dim3(256), 0, s, buf, n); hipLaunchKernelGGL(k_empty, dim3(32), dim3(256), 0, s, o, 1); }));
    printf("write %5d KB sc1,   empty           : %.2f\n", grid * 256 * 16 * n / 1024, timeit(2, [&](hipStream_t s) { hipLaunchKernelGGL(k_write<1>, dim3(grid), dim3(256), 0, s, buf, n); hipLaunchKernelGGL(k_empty, dim3(32), dim3(256), 0, s, o, 1); }));
    printf("write %5d KB plain, read it         : %.2f\n", grid * 256 * 16 * n / 1024, timeit(2, [&](hipStream_t s) { hipLaunchKernelGGL(k_write<0>, dim3(grid), dim3(256), 0, s, buf, n); hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, s, buf, o, n); }));
    printf("write %5d KB sc1,   read it         : %.2f\n", grid * 256 * 16 * n / 1024, timeit(2, [&](hipStream_t s) { hipLaunchKernelGGL(k_write<1>, dim3(grid), dim3(256), 0, s, buf, n); hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, s, buf, o, n); }));
  }
  printf("read 8 MB (cold), empty              : %.2f\n", timeit(2, [&](hipStream_t s) { hipLaunchKernelGGL(k_read, dim3(32), dim3(256), 0, s, buf, o, 64); hipLaunchKernelGGL(k_empty, dim3(32), dim3(256), 0, s, o, 1); }));
  return 0;
}
