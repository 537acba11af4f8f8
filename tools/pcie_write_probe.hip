// tools/pcie_write_probe.hip -- device stores into pinned host memory: does the run length per store instruction matter?
// An 1920x1080 BGRA8 image written (a) tile-wise as the trace kernel does (a wave = 8x8 pixels: eight 32-byte runs per store),
// (b) two 128-byte runs per store (a wave = 2 rows of a 32-pixel block), (c) one 256-byte run per store (a wave = 64 pixels of a row).
//   hipcc --offload-arch=gfx950 -O3 -o tools/pcie_write_probe tools/pcie_write_probe.hip && tools/pcie_write_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void w(uint32_t* out, uint32_t W, uint32_t H) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t px, py;
  if (MODE == 0) { px = blockIdx.x * 32u + wave * 8u + (lane & 7u); py = blockIdx.y * 8u + (lane >> 3); }
  else if (MODE == 1) { px = blockIdx.x * 32u + (lane & 31u); py = blockIdx.y * 8u + wave * 2u + (lane >> 5); }
  else { const uint32_t i = (blockIdx.y * gridDim.x + blockIdx.x) * 256u + threadIdx.x; px = i % W; py = i / W; }
  if (px < W && py < H) out[py * W + px] = px ^ (py << 16);
}
int main() {
  const uint32_t W = 1920, H = 1080;
  uint32_t *host, *dev; CK(hipHostMalloc(&host, W * H * 4, hipHostMallocDefault)); CK(hipMalloc(&dev, W * H * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep)
    for (int target = 0; target < 2; ++target)
      for (int mode = 0; mode < 3; ++mode) {
        uint32_t* out = target ? host : dev;
        const dim3 grid = mode == 2 ? dim3(W * H / 256) : dim3(W / 32, H / 8);
        const int n = 20;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < n; ++i) {
          if (mode == 0) hipLaunchKernelGGL(w<0>, grid, dim3(256), 0, 0, out, W, H);
          if (mode == 1) hipLaunchKernelGGL(w<1>, grid, dim3(256), 0, 0, out, W, H);
          if (mode == 2) hipLaunchKernelGGL(w<2>, grid, dim3(256), 0, 0, out, W, H);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        static const char* nm[3] = {"8 x 32-byte runs per store (8x8 tile)", "2 x 128-byte runs per store", "1 x 256-byte run per store"};
        if (rep == 1) printf("%-6s %-40s : %7.1f us per image = %5.1f GB/s\n", target ? "host" : "device", nm[mode], ms / n * 1e3, W * H * 4.0 / (ms / n * 1e-3) / 1e9);
      }
  return 0;
}
