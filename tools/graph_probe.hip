// tools/graph_probe.hip -- what does the HOST pay per step when the step's launches (list build on one stream, two half-frame
// kernels on two others behind an event) are replayed from a captured hipGraph instead of being enqueued call by call?
//   hipcc --offload-arch=gfx950 -O3 -o tools/graph_probe tools/graph_probe.hip && tools/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Big { float v[96]; };                                   // a kernel argument the size of TraceParams
__global__ void k(Big b, float* out, int spin) {
  float a = b.v[threadIdx.x & 63];
  for (int i = 0; i < spin; ++i) a = __builtin_fmaf(a, 1.0000001f, 1e-7f);
  if (a == 123.456f) out[0] = a;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  float* d; CK(hipMalloc(&d, 4));
  hipStream_t sl, sa, sb; CK(hipStreamCreateWithFlags(&sl, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const int P = 40;                                            // steps per graph
  std::vector<hipEvent_t> ready(P), joinb(P);
  for (auto& e : ready) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : joinb) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t fork, jl, jb; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&jl, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&jb, hipEventDisableTiming));
  Big b{}; const int spin = 20000;                             // ~ 50 us kernels on 2 x 512 waves
  auto step = [&](int i) {
    hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, sl, b, d, spin / 4);
    hipEventRecord(ready[i], sl);
    hipStreamWaitEvent(sa, ready[i], 0); hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, sa, b, d, spin);
    hipStreamWaitEvent(sb, ready[i], 0); hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, sb, b, d, spin);
  };
  // direct
  for (int i = 0; i < P; ++i) step(i);
  CK(hipDeviceSynchronize());
  double t0 = now();
  for (int r = 0; r < 5; ++r) for (int i = 0; i < P; ++i) step(i);
  double t1 = now();
  CK(hipDeviceSynchronize());
  double t2 = now();
  printf("direct : host %.2f us per step, wall %.2f us per step\n", (t1 - t0) / (5 * P) * 1e6, (t2 - t0) / (5 * P) * 1e6);
  // captured: origin sa; sl and sb join through events
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal));
  CK(hipEventRecord(fork, sa)); CK(hipStreamWaitEvent(sl, fork, 0)); CK(hipStreamWaitEvent(sb, fork, 0));
  for (int i = 0; i < P; ++i) step(i);
  CK(hipEventRecord(jl, sl)); CK(hipEventRecord(jb, sb)); CK(hipStreamWaitEvent(sa, jl, 0)); CK(hipStreamWaitEvent(sa, jb, 0));
  CK(hipStreamEndCapture(sa, &g));
  size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, sa)); CK(hipStreamSynchronize(sa));
  t0 = now();
  for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, sa));
  t1 = now();
  CK(hipStreamSynchronize(sa));
  t2 = now();
  printf("graph  : %zu nodes; host %.2f us per step, wall %.2f us per step\n", nn, (t1 - t0) / (5 * P) * 1e6, (t2 - t0) / (5 * P) * 1e6);
  return 0;
}
