// valu_peak.hip -- calibrates the attainable fp32 VALU issue rate on this MI355X, the
// honest denominator for the trace kernel's VALU roofline (the path is VALU-bound).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_peak.hip -o tools/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int CHAINS>
__global__ __launch_bounds__(256) void fma_kernel(float* out, int iters, unsigned long long* clk) {
  float a[CHAINS];
  const float x = 1.0000001f + threadIdx.x * 1e-9f, y = 1e-7f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) a[c] = c + threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i += 16) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) a[c] = __builtin_fmaf(a[c], x, y);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += a[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int CHAINS>
double run(int blocks, int iters, float* d_out, unsigned long long* d_clk, double* ghz) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(fma_kernel<CHAINS>, dim3(blocks), dim3(256), 0, 0, d_out, iters, d_clk);
  hipEventRecord(e0);
  hipLaunchKernelGGL(fma_kernel<CHAINS>, dim3(blocks), dim3(256), 0, 0, d_out, iters, d_clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long clk[2];
  hipMemcpy(clk, d_clk, sizeof clk, hipMemcpyDeviceToHost);
  *ghz = (double)clk[0] / (double)clk[1] * 0.1;     // s_memrealtime ticks at 100 MHz
  return (double)blocks * 256.0 * iters * CHAINS / (ms * 1e-3);   // lane-FMA per second
}

int main() {
  float* d_out; unsigned long long* d_clk;
  hipMalloc(&d_out, 256 * 64 * 256 * sizeof(float)); hipMalloc(&d_clk, 16);
  const int iters = 20000;
  printf("%-28s %14s %10s %8s\n", "case", "Tlane-FMA/s", "TFLOP/s", "GHz");
  for (int wpb : {1, 2, 4, 8}) {       // blocks of 4 waves per CU -> waves per SIMD
    double ghz, r;
    const int blocks = 256 * wpb;
    r = run<8>(blocks, iters, d_out, d_clk, &ghz);
    printf("fma x8 chains, %d waves/SIMD   %14.3f %10.2f %8.3f\n", wpb, r / 1e12, 2 * r / 1e12, ghz);
    r = run<4>(blocks, iters, d_out, d_clk, &ghz);
    printf("fma x4 chains, %d waves/SIMD   %14.3f %10.2f %8.3f\n", wpb, r / 1e12, 2 * r / 1e12, ghz);
  }
  return 0;
}
