// tools/pk_probe.hip -- does v_pk_fma_f32 buy anything on gfx950 when a SIMD holds few waves?  The same number of fp32 fmas as
// plain v_fma_f32 (8 independent chains per lane) and as v_pk_fma_f32 (4 independent chains of 2), at 1 .. 8 resident waves per
// SIMD (grid = 256 CUs x 4 SIMDs x n waves, one wave per workgroup... of 64 threads).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o tools/pk_probe tools/pk_probe.hip && tools/pk_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int DEP>   // DEP independent chains (8 plain / 4 packed = same flops)
__global__ __launch_bounds__(64) void plain(float* out, int iters) {
  float a[DEP];
#pragma unroll
  for (int c = 0; c < DEP; ++c) a[c] = threadIdx.x * 1e-6f + c;
  const float x = 1.0000001f, y = 1e-7f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < DEP; ++c) a[c] = __builtin_fmaf(a[c], x, y);
  }
  float s = 0; for (int c = 0; c < DEP; ++c) s += a[c];
  if (s == 12345.678f) out[0] = s;
}
template <int DEP>
__global__ __launch_bounds__(64) void packed(float* out, int iters) {
  f2 a[DEP];
#pragma unroll
  for (int c = 0; c < DEP; ++c) a[c] = f2{threadIdx.x * 1e-6f + c, threadIdx.x * 2e-6f + c};
  const f2 x = {1.0000001f, 1.0000002f}, y = {1e-7f, 2e-7f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < DEP; ++c) a[c] = __builtin_elementwise_fma(a[c], x, y);
  }
  float s = 0; for (int c = 0; c < DEP; ++c) s += a[c].x + a[c].y;
  if (s == 12345.678f) out[0] = s;
}

int main() {
  float* d; CK(hipMalloc(&d, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;                                   // 2000 x 8 x DEP fmas per lane
  for (int rep = 0; rep < 2; ++rep)
    for (int occ = 1; occ <= 8; ++occ) {
      const int waves = 256 * 4 * occ;
      float ms[4];
      for (int v = 0; v < 4; ++v) {
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 5; ++r) {
          if (v == 0) hipLaunchKernelGGL(plain<8>, dim3(waves), dim3(64), 0, 0, d, iters);          // 8 chains
          if (v == 1) hipLaunchKernelGGL(packed<4>, dim3(waves), dim3(64), 0, 0, d, iters);         // 4 packed chains: same fmas
          if (v == 2) hipLaunchKernelGGL(plain<2>, dim3(waves), dim3(64), 0, 0, d, iters * 4);      // 2 chains (little ILP), same fmas
          if (v == 3) hipLaunchKernelGGL(packed<1>, dim3(waves), dim3(64), 0, 0, d, iters * 4);     // 1 packed chain, same fmas
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[v], e0, e1)); ms[v] /= 5;
      }
      const double fma = double(waves) * 64 * iters * 64;   // lane-fmas per launch
      if (rep == 1) printf("%d waves/SIMD: plain x8 %7.3f ms (%5.1f Tfma/s)  packed x4 %7.3f ms (%5.1f)  plain x2 %7.3f ms (%5.1f)  packed x1 %7.3f ms (%5.1f)\n", occ,
                           ms[0], fma / ms[0] / 1e9, ms[1], fma / ms[1] / 1e9, ms[2], fma / ms[2] / 1e9, ms[3], fma / ms[3] / 1e9);
    }
  return 0;
}
