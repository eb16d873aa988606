// What does one dependent launch cost inside a hipGraph on this stack, as a function of the kernel-argument size,
// the workgroup size and the static LDS?  (The skinny schedule is ten ~1 us kernels: its floor is this number x 10.)
// build: hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip ; run: ./launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
template <int N> struct Args { int v[N]; float* p; };
template <int N, int LDSF> __global__ void k_empty(const Args<N> a) {
  __shared__ float lds[LDSF > 0 ? LDSF : 1];
  if (a.v[0] < 0) { lds[threadIdx.x % (LDSF > 0 ? LDSF : 1)] = 1.f; a.p[0] = lds[0]; }
}
template <int N, int LDSF> double run(int grid, int block, int chain, float* buf) {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  Args<N> a; for (int i = 0; i < N; ++i) a.v[i] = i + 1; a.p = buf;
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < chain; ++i) hipLaunchKernelGGL((k_empty<N, LDSF>), dim3(grid), dim3(block), 0, s, a);
  hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  std::vector<float> ms;
  for (int r = 0; r < 30; ++r) { hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipStreamSynchronize(s); float m; hipEventElapsedTime(&m, e0, e1); ms.push_back(m); }
  std::sort(ms.begin(), ms.end());
  hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(s);
  return ms[ms.size() / 2] * 1000.0 / chain;
}
int main() {
  float* buf; hipMalloc(&buf, 1024);
  const int chain = 400;
  printf("us per launch in a %d-kernel hipGraph chain (median of 30 replays)\n", chain);
  printf("args   16 B, 256 thr x 32 WG, no LDS : %.2f\n", run<2, 0>(32, 256, chain, buf));
  printf("args   16 B, 512 thr x 32 WG, no LDS : %.2f\n", run<2, 0>(32, 512, chain, buf));
  printf("args   16 B, 512 thr x 32 WG, 32 KB  : %.2f\n", run<2, 8192>(32, 512, chain, buf));
  printf("args  272 B, 512 thr x 32 WG, 32 KB  : %.2f\n", run<66, 8192>(32, 512, chain, buf));
  printf("args 1.1 KB, 512 thr x 32 WG, 32 KB  : %.2f\n", run<280, 8192>(32, 512, chain, buf));
  printf("args 3.1 KB, 512 thr x 32 WG, 32 KB  : %.2f\n", run<780, 8192>(32, 512, chain, buf));
  printf("args 1.1 KB, 512 thr x 256 WG, 32 KB : %.2f\n", run<280, 8192>(256, 512, chain, buf));
  printf("args 1.1 KB, 256 thr x 64 WG, no LDS : %.2f\n", run<280, 0>(64, 256, chain, buf));
  printf("args   16 B, 256 thr x 1 WG, no LDS  : %.2f\n", run<2, 0>(1, 256, chain, buf));
  return 0;
}
