// tools/dispatch_floor.hip -- what a launch of the C3 trace grid costs before / without computing anything, and a synthetic
// model of the mixed frame (attribute the empty wave slots before building).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/dispatch_floor tools/dispatch_floor.hip
//   tools/dispatch_floor [kinds.bin]        (kinds.bin: one byte per 8x8 tile of the 1920x1080 frame in trace-grid order,
//                                            1 = the tile generates rays; tools/dump_kinds.py writes it from the product's lists)
//
// Kernels share the trace kernel's geometry: one lane = one pixel, a wave = an 8x8 tile, a 32x8 block of four tiles is
// traced by 4 / WGW workgroups of WGW waves; 72 VGPRs, 1280 bytes of LDS per wave; per-pixel state = six u32 planes
// (read + written), float4 accumulator, count, BGRA8 word (written): the 24 + 48 bytes per pixel of a clearing C3 launch.
//   mode 0  empty: the dispatch floor of the grid
//   mode 1  state stream: loads and stores only
//   mode 2  + the 48 discarded XORWOW draws of a certain-winner tile (the all-certain frame)
//   mode 3  mixed: tiles marked in kinds.bin spin `heavy` VALU instructions instead (a ray-generating tile), the others as mode 2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct P {
  uint32_t* rng; float4* render; uint32_t* counts; uint32_t* image; const uint8_t* kinds;
  uint32_t W, rows, npix, heavy, by0;
  const uint32_t* order;      // mode 3 experiments: tile visited by wave slot i (null: the grid's own tile), see main()
  uint32_t n_slots, slot0;    // entries of `order` this launch covers: [slot0, slot0 + n_slots)
  uint32_t tile_major;        // RNG planes and counts stored tile by tile (64 consecutive words per tile and plane)
  uint32_t prio;              // 1: every wave starts at s_setprio 3, a heavy wave computes at priority 0 and stores at 3 again
  uint32_t heavy_only;        // mode 3: waves of tiles that are not marked leave at once (their pixels belong to light_blocks_kernel)
};

template <int MODE, int WGW>
__global__ __launch_bounds__(64 * WGW, 5) void k(const P p) {
  extern __shared__ float4 s_mem[];
  constexpr uint32_t WGS = 4u / WGW;
  const uint32_t lane = threadIdx.x & 63u, wl = threadIdx.x >> 6;
  uint32_t wave = (blockIdx.x % WGS) * WGW + wl, bx = blockIdx.x / WGS, by = blockIdx.y + p.by0, gxb = gridDim.x / WGS;
  asm volatile("v_mov_b32 v71, 0" ::: "v71");                          // the trace kernel's register footprint
  if (MODE == 3 && p.prio) __builtin_amdgcn_s_setprio(3);
  if (MODE == 0) { if (p.W == 0u) s_mem[threadIdx.x] = make_float4(0, 0, 0, 0); return; }
  if (p.order != nullptr) {                                            // 1-D grid over wave slots
    const uint32_t slot = blockIdx.x * WGW + wl;
    if (slot >= p.n_slots) return;
    const uint32_t t = __builtin_amdgcn_readfirstlane(p.order[p.slot0 + slot]);
    gxb = (p.W + 31u) / 32u;
    wave = t & 3u; bx = (t >> 2) % gxb; by = (t >> 2) / gxb;
  }
  const uint32_t px = bx * 32u + wave * 8u + (lane & 7u), ly = by * 8u + (lane >> 3);
  const bool inside = px < p.W && ly < p.rows;
  const size_t pix = inside ? static_cast<size_t>(px) + static_cast<size_t>(ly) * p.W : 0;
  if (MODE == 3 && p.heavy_only && __builtin_amdgcn_readfirstlane(p.kinds[(static_cast<size_t>(by) * gxb + bx) * 4u + wave]) == 0) return;
  const size_t spix = p.tile_major ? ((static_cast<size_t>(by) * gxb + bx) * 4u + wave) * 64u + lane : pix;   // state index
  uint32_t d = p.rng[spix], v0 = p.rng[p.npix + spix], v1 = p.rng[2 * (size_t)p.npix + spix], v2 = p.rng[3 * (size_t)p.npix + spix],
           v3 = p.rng[4 * (size_t)p.npix + spix], v4 = p.rng[5 * (size_t)p.npix + spix];
  float ax = 0.25f, ay = 0.5f, az = 0.75f;
  bool heavy = false;
  if (MODE == 3) heavy = __builtin_amdgcn_readfirstlane(p.kinds[(static_cast<size_t>(by) * gxb + bx) * 4u + wave]) != 0;
  if (MODE == 3 && p.heavy_only && !heavy) return;
  if (MODE >= 2 && !heavy) {
    auto f = [](uint32_t x, uint32_t v) -> uint32_t { const uint32_t t = x ^ (x >> 2); return (v ^ (v << 4)) ^ (t ^ (t << 1)); };
    for (uint32_t i = 0; i + 5u <= 48u; i += 5u) { v0 = f(v0, v4); v1 = f(v1, v0); v2 = f(v2, v1); v3 = f(v3, v2); v4 = f(v4, v3); }
    for (uint32_t i = 45; i < 48u; ++i) { const uint32_t nv = f(v0, v4); v0 = v1; v1 = v2; v2 = v3; v3 = v4; v4 = nv; }
    d += 362437u * 48u;
  }
  if (MODE == 3 && heavy) {
    if (p.prio) __builtin_amdgcn_s_setprio(0);
    float a[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = static_cast<float>(v0 >> (c + 8)) * 1e-9f;
    const float x = 1.0000001f, y = 1e-7f;
    for (uint32_t i = 0; i < p.heavy; i += 64u) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] = __builtin_fmaf(a[c], x, y);
      }
    }
    ax = a[0] + a[1] + a[2]; ay = a[3] + a[4] + a[5]; az = a[6] + a[7];
    v4 ^= 1u;
    if (p.prio) __builtin_amdgcn_s_setprio(3);
  }
  if (inside) {
    p.counts[spix] = 16u;
    p.render[pix] = make_float4(ax, ay, az, 0.0f);
    p.image[pix] = __builtin_bit_cast(uint32_t, ax) ^ d;
    p.rng[spix] = d; p.rng[p.npix + spix] = v0; p.rng[2 * (size_t)p.npix + spix] = v1; p.rng[3 * (size_t)p.npix + spix] = v2;
    p.rng[4 * (size_t)p.npix + spix] = v3; p.rng[5 * (size_t)p.npix + spix] = v4;
  }
}

// The light tiles of a 32x8 block by ONE wave: lane l owns the pixels (x0 + l % 32, y0 + 2 j + l / 32), j = 0..3 -- every load and
// store instruction covers two whole 128-byte lines of a plane -- and all 24 state loads are in flight before the first draw:
// four times the bytes in flight per wave slot of the tile-per-wave kernel.  Lanes whose tile is marked heavy are masked.
template <int WGW>
__global__ __launch_bounds__(64 * WGW, 5) void light_blocks_kernel(const P p) {
  const uint32_t lane = threadIdx.x & 63u, wl = threadIdx.x >> 6;
  const uint32_t gxb = (p.W + 31u) / 32u, gyb = (p.rows + 7u) / 8u;
  const uint32_t blk = blockIdx.x * WGW + wl;
  if (blk >= gxb * gyb) return;
  const uint32_t bx = blk % gxb, by = blk / gxb;
  const uint32_t kw = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const uint32_t*>(p.kinds + static_cast<size_t>(blk) * 4u));   // 4 tiles' kinds
  if ((kw & 0x01010101u) == 0x01010101u) return;                       // no light tile in this block
  const uint32_t px = bx * 32u + (lane & 31u);
  const bool mine = ((kw >> (8u * ((lane & 31u) >> 3))) & 1u) == 0u && px < p.W;
  uint32_t st[4][6];
  size_t pix[4];
  bool in[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t ly = by * 8u + 2u * j + (lane >> 5);
    in[j] = mine && ly < p.rows;
    pix[j] = in[j] ? static_cast<size_t>(px) + static_cast<size_t>(ly) * p.W : 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) st[j][k] = p.rng[k * static_cast<size_t>(p.npix) + pix[j]];
  }
  auto f = [](uint32_t x, uint32_t v) -> uint32_t { const uint32_t t = x ^ (x >> 2); return (v ^ (v << 4)) ^ (t ^ (t << 1)); };
  for (uint32_t i = 0; i + 5u <= 48u; i += 5u) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { uint32_t* s = st[j]; s[1] = f(s[1], s[5]); s[2] = f(s[2], s[1]); s[3] = f(s[3], s[2]); s[4] = f(s[4], s[3]); s[5] = f(s[5], s[4]); }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t* s = st[j];
    for (uint32_t i = 45; i < 48u; ++i) { const uint32_t nv = f(s[1], s[5]); s[1] = s[2]; s[2] = s[3]; s[3] = s[4]; s[4] = s[5]; s[5] = nv; }
    s[0] += 362437u * 48u;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (in[j]) {
      p.counts[pix[j]] = 16u;
      p.render[pix[j]] = make_float4(0.25f, 0.5f, 0.75f, 0.0f);
      p.image[pix[j]] = 0x3e800000u ^ st[j][0];
#pragma unroll
      for (int k = 0; k < 6; ++k) p.rng[k * static_cast<size_t>(p.npix) + pix[j]] = st[j][k];
    }
  }
}

// Persistent heavy waves: the grid is sized to the device (n workgroups per CU, kept there by an LDS pad), every wave pulls the
// next marked tile from a compacted list through an atomic counter until the list is empty; the last wave to leave resets the
// counters for the next launch.  ctr[0] = next entry, ctr[1] = waves that have left; ctr[2 + x] = next entry of XCD x's share (per_xcd).
struct HP { const uint32_t* list; uint32_t n; uint32_t* ctr; uint32_t n_waves; uint32_t per_xcd; uint32_t chunk; };
template <int WGW>
__global__ __launch_bounds__(64 * WGW, 5) void heavy_persistent_kernel(const P p, const HP h) {
  extern __shared__ float4 s_mem[];
  const uint32_t lane = threadIdx.x & 63u;
  asm volatile("v_mov_b32 v71, 0" ::: "v71");
  if (p.W == 0u) s_mem[threadIdx.x] = make_float4(0, 0, 0, 0);
  const uint32_t gxb = (p.W + 31u) / 32u;
  uint32_t xcc = 0;
  if (h.per_xcd) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 7u; }
  const uint32_t share = (h.n + 7u) / 8u;
  if (h.chunk == 0u) {                                                  // static: entry w, w + n_waves, ... (no counter at all)
    const uint32_t w = blockIdx.x * WGW + (threadIdx.x >> 6);
    for (uint32_t i = w; i < h.n; i += h.n_waves) {
      const uint32_t t = __builtin_amdgcn_readfirstlane(h.list[i]);
      const uint32_t wave = t & 3u, bx = (t >> 2) % gxb, by = (t >> 2) / gxb;
      const uint32_t px = bx * 32u + wave * 8u + (lane & 7u), ly = by * 8u + (lane >> 3);
      const bool inside = px < p.W && ly < p.rows;
      const size_t pix = inside ? static_cast<size_t>(px) + static_cast<size_t>(ly) * p.W : 0;
      uint32_t d = p.rng[pix], v0 = p.rng[p.npix + pix], v1 = p.rng[2 * (size_t)p.npix + pix], v2 = p.rng[3 * (size_t)p.npix + pix],
               v3 = p.rng[4 * (size_t)p.npix + pix], v4 = p.rng[5 * (size_t)p.npix + pix];
      float a[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) a[c] = static_cast<float>(v0 >> (c + 8)) * 1e-9f;
      const float x = 1.0000001f, y = 1e-7f;
      for (uint32_t k = 0; k < p.heavy; k += 64u) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
          for (int c = 0; c < 8; ++c) a[c] = __builtin_fmaf(a[c], x, y);
        }
      }
      v4 ^= 1u;
      if (inside) {
        p.counts[pix] = 16u;
        p.render[pix] = make_float4(a[0] + a[1] + a[2], a[3] + a[4] + a[5], a[6] + a[7], 0.0f);
        p.image[pix] = __builtin_bit_cast(uint32_t, a[0]) ^ d;
        p.rng[pix] = d; p.rng[p.npix + pix] = v0; p.rng[2 * (size_t)p.npix + pix] = v1; p.rng[3 * (size_t)p.npix + pix] = v2;
        p.rng[4 * (size_t)p.npix + pix] = v3; p.rng[5 * (size_t)p.npix + pix] = v4;
      }
    }
    return;
  }
  for (uint32_t round = 0; round < 9u; ++round) {                      // per_xcd: own share first, then the others' (work stealing)
    const uint32_t part = h.per_xcd ? (xcc + round) & 7u : 0u;
    const uint32_t lo = h.per_xcd ? part * share : 0u, hi = h.per_xcd ? (lo + share < h.n ? lo + share : h.n) : h.n;
    if (h.per_xcd == 0u && round > 0u) break;
    if (h.per_xcd && round >= 8u) break;
    for (;;) {
      uint32_t i0 = 0;
      if (lane == 0u) i0 = atomicAdd(h.ctr + (h.per_xcd ? 2u + part : 0u), h.chunk);
      i0 = __builtin_amdgcn_readfirstlane(i0) + lo;
      if (i0 >= hi) break;
      for (uint32_t i = i0; i < i0 + h.chunk && i < hi; ++i) {
        const uint32_t t = __builtin_amdgcn_readfirstlane(h.list[i]);
        const uint32_t wave = t & 3u, bx = (t >> 2) % gxb, by = (t >> 2) / gxb;
        const uint32_t px = bx * 32u + wave * 8u + (lane & 7u), ly = by * 8u + (lane >> 3);
        const bool inside = px < p.W && ly < p.rows;
        const size_t pix = inside ? static_cast<size_t>(px) + static_cast<size_t>(ly) * p.W : 0;
        uint32_t d = p.rng[pix], v0 = p.rng[p.npix + pix], v1 = p.rng[2 * (size_t)p.npix + pix], v2 = p.rng[3 * (size_t)p.npix + pix],
                 v3 = p.rng[4 * (size_t)p.npix + pix], v4 = p.rng[5 * (size_t)p.npix + pix];
        float a[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] = static_cast<float>(v0 >> (c + 8)) * 1e-9f;
        const float x = 1.0000001f, y = 1e-7f;
        for (uint32_t k = 0; k < p.heavy; k += 64u) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < 8; ++c) a[c] = __builtin_fmaf(a[c], x, y);
          }
        }
        v4 ^= 1u;
        if (inside) {
          p.counts[pix] = 16u;
          p.render[pix] = make_float4(a[0] + a[1] + a[2], a[3] + a[4] + a[5], a[6] + a[7], 0.0f);
          p.image[pix] = __builtin_bit_cast(uint32_t, a[0]) ^ d;
          p.rng[pix] = d; p.rng[p.npix + pix] = v0; p.rng[2 * (size_t)p.npix + pix] = v1; p.rng[3 * (size_t)p.npix + pix] = v2;
          p.rng[4 * (size_t)p.npix + pix] = v3; p.rng[5 * (size_t)p.npix + pix] = v4;
        }
      }
    }
  }
  if (lane == 0u) {
    const uint32_t left = atomicAdd(h.ctr + 1, 1u);
    if (left + 1u == h.n_waves) { for (int k = 0; k < 10; ++k) if (k != 1) atomicExch(h.ctr + k, 0u); __threadfence(); atomicExch(h.ctr + 1, 0u); }
  }
}

static size_t g_lds_pad = 0;       // extra dynamic LDS per workgroup: caps the workgroups a CU holds (160 KiB per CU)
template <int MODE>
static void launch(int wgw, const P& p, dim3 grid, hipStream_t st) {
  const size_t lds = 1280u * wgw + g_lds_pad;
  if (p.order != nullptr) grid = dim3((p.n_slots + wgw - 1) / wgw); else grid.x *= 4 / wgw;
  if (wgw == 4) hipLaunchKernelGGL((k<MODE, 4>), grid, dim3(256), lds, st, p);
  else if (wgw == 2) hipLaunchKernelGGL((k<MODE, 2>), grid, dim3(128), lds, st, p);
  else hipLaunchKernelGGL((k<MODE, 1>), grid, dim3(64), lds, st, p);
}
static void launch_mode(int mode, int wgw, const P& p, dim3 grid, hipStream_t st) {
  switch (mode) { case 0: launch<0>(wgw, p, grid, st); break; case 1: launch<1>(wgw, p, grid, st); break;
                  case 2: launch<2>(wgw, p, grid, st); break; default: launch<3>(wgw, p, grid, st); break; }
}

int main(int argc, char** argv) {
  const uint32_t W = 1920, H = 1080, gx = 60, gy = 135;
  const size_t npix = static_cast<size_t>(W) * H;
  P p{};
  CK(hipMalloc(&p.rng, npix * 24)); CK(hipMalloc(&p.render, npix * 16)); CK(hipMalloc(&p.counts, npix * 4)); CK(hipMalloc(&p.image, npix * 4));
  CK(hipMemset(p.rng, 0x5a, npix * 24));
  std::vector<uint8_t> kinds(static_cast<size_t>(gx) * gy * 4u, 0);
  double share = 0.0;
  if (argc > 1) {
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(kinds.data(), 1, kinds.size(), f) != kinds.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    fclose(f);
  } else {                                                              // 30 % of the tiles, in clusters of 4 x 2 blocks
    uint64_t s = 12345;
    for (uint32_t by = 0; by < gy; ++by) for (uint32_t bx = 0; bx < gx; ++bx) {
      uint64_t h = ((by / 2) * 977u + (bx / 4)) * 0x9E3779B97F4A7C15ull + s; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
      for (uint32_t w = 0; w < 4; ++w) kinds[(by * gx + bx) * 4u + w] = (h % 100u) < 30u;
    }
  }
  for (uint8_t v : kinds) share += v != 0;
  share /= kinds.size();
  uint8_t* dk; CK(hipMalloc(&dk, kinds.size())); CK(hipMemcpy(dk, kinds.data(), kinds.size(), hipMemcpyHostToDevice));
  p.kinds = dk; p.W = W; p.rows = H; p.npix = static_cast<uint32_t>(npix);
  p.heavy = argc > 2 ? atoi(argv[2]) : 2816;
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1, eb; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
  printf("tiles marked ray-generating: %.3f; heavy = %u VALU instructions per wave\n", share, p.heavy);
  for (int rep = 0; rep < 2; ++rep)                                     // (the first pass brings the clocks up; the second is printed)
  for (int mode = 0; mode <= 3; ++mode)
    for (int wgw = 4; wgw >= 1; wgw >>= 1)
      for (int split = 0; split <= 1; ++split) {
        const int n = 60;
        auto step = [&] {
          if (!split) { launch_mode(mode, wgw, p, dim3(gx, gy), sa); return; }
          P a = p, b = p; b.by0 = 68;                                   // upper 68 block rows on sa, lower 67 on sb
          launch_mode(mode, wgw, a, dim3(gx, 68), sa);
          launch_mode(mode, wgw, b, dim3(gx, gy - 68), sb);
        };
        for (int i = 0; i < 10; ++i) step();
        CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
        CK(hipEventRecord(e0, sa));
        for (int i = 0; i < n; ++i) step();
        CK(hipEventRecord(eb, sb)); CK(hipStreamWaitEvent(sa, eb, 0));
        CK(hipEventRecord(e1, sa));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 1) printf("mode %d  %d-wave workgroups  %s : %7.2f us per launch\n", mode, wgw, split ? "two half-frame kernels on two streams" : "one kernel", ms / n * 1e3);
      }
  const uint32_t nt0 = static_cast<uint32_t>(kinds.size());
  // ---- the frame as two kernels by KIND of tile: light_blocks_kernel (one wave per block, its light tiles) + the tile-per-wave
  //      kernel for the marked tiles only (the other waves leave at once)
  for (int lw = 1; lw <= 4; lw <<= 1)
    for (int variant = 0; variant < 4; ++variant) {
      // 0: light then heavy on one stream; 1: heavy then light on one stream; 2: light on sa, heavy on sb; 3: light on sa, heavy in two halves on sa / sb
      auto light = [&](hipStream_t st) {
        const uint32_t blocks = gx * gy;
        if (lw == 1) hipLaunchKernelGGL(light_blocks_kernel<1>, dim3(blocks), dim3(64), 0, st, p);
        else if (lw == 2) hipLaunchKernelGGL(light_blocks_kernel<2>, dim3((blocks + 1) / 2), dim3(128), 0, st, p);
        else hipLaunchKernelGGL(light_blocks_kernel<4>, dim3((blocks + 3) / 4), dim3(256), 0, st, p);
      };
      P h = p; h.heavy_only = 1;
      auto step = [&] {
        if (variant == 0) { light(sa); launch_mode(3, 4, h, dim3(gx, gy), sa); }
        else if (variant == 1) { launch_mode(3, 4, h, dim3(gx, gy), sa); light(sa); }
        else if (variant == 2) { light(sa); launch_mode(3, 4, h, dim3(gx, gy), sb); }
        else { P b = h; b.by0 = 68; launch_mode(3, 4, h, dim3(gx, 68), sb); light(sa); launch_mode(3, 4, b, dim3(gx, gy - 68), sa); }
      };
      const int n = 60;
      for (int i = 0; i < 10; ++i) step();
      CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
      CK(hipEventRecord(e0, sa));
      for (int i = 0; i < n; ++i) step();
      CK(hipEventRecord(eb, sb)); CK(hipStreamWaitEvent(sa, eb, 0));
      CK(hipEventRecord(e1, sa));
      CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      static const char* vn[4] = {"light, heavy on one stream", "heavy, light on one stream", "light | heavy on two streams", "heavy upper | light, heavy lower"};
      printf("by kind: light blocks by one wave each (%d-wave wg) + heavy tiles; %-34s : %7.2f us\n", lw, vn[variant], ms / n * 1e3);
    }
  for (size_t pad : {size_t(0), size_t(18) << 10, size_t(25) << 10, size_t(31) << 10, size_t(39) << 10, size_t(52) << 10, size_t(79) << 10}) {
    // the heavy kernel capped in workgroups per CU by an LDS pad: alone, and beside the light kernel
    for (int variant = 0; variant < 3; ++variant) {
      P h = p; h.heavy_only = 1;
      auto step = [&] {
        g_lds_pad = pad;
        if (variant == 0) launch_mode(3, 4, h, dim3(gx, gy), sa);
        else if (variant == 1) { launch_mode(3, 4, h, dim3(gx, gy), sb); g_lds_pad = 0; hipLaunchKernelGGL(light_blocks_kernel<4>, dim3((gx * gy + 3) / 4), dim3(256), 0, sa, p); }
        else { launch_mode(3, 4, h, dim3(gx, gy), sa); g_lds_pad = 0; hipLaunchKernelGGL(light_blocks_kernel<4>, dim3((gx * gy + 3) / 4), dim3(256), 0, sa, p); }
        g_lds_pad = 0;
      };
      const int n = 60;
      for (int i = 0; i < 10; ++i) step();
      CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
      CK(hipEventRecord(e0, sa));
      for (int i = 0; i < n; ++i) step();
      CK(hipEventRecord(eb, sb)); CK(hipStreamWaitEvent(sa, eb, 0));
      CK(hipEventRecord(e1, sa));
      CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      static const char* vn[3] = {"heavy alone", "heavy on sb | light blocks on sa", "heavy, light blocks on one stream"};
      printf("heavy kernel with %2zu KiB LDS pad (<= %zu workgroups per CU): %-34s : %7.2f us\n", pad >> 10, (size_t(160) << 10) / (pad + 5120), vn[variant], ms / n * 1e3);
    }
  }
  {
    // persistent heavy waves: n workgroups of 4 waves per CU (LDS pad keeps them there), alone and beside the light-block kernel
    std::vector<uint32_t> hl;
    for (uint32_t t = 0; t < nt0; ++t) if (kinds[t]) hl.push_back(t);
    uint32_t* dl; CK(hipMalloc(&dl, hl.size() * 4 + 4)); CK(hipMemcpy(dl, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
    uint32_t* dc; CK(hipMalloc(&dc, 64)); CK(hipMemset(dc, 0, 64));
    hipStream_t shi; int plo = 0, phi = 0; CK(hipDeviceGetStreamPriorityRange(&plo, &phi)); CK(hipStreamCreateWithPriority(&shi, hipStreamNonBlocking, phi));
    for (int per_cu = 1; per_cu <= 5; ++per_cu)
      for (int per_xcd = 0; per_xcd <= 0; ++per_xcd)
        for (uint32_t chunk : {0u})
          for (int variant = 0; variant < 3; ++variant) {
            const size_t pad = per_cu == 1 ? (size_t(90) << 10) : per_cu == 2 ? (size_t(60) << 10) : per_cu == 3 ? (size_t(42) << 10) : per_cu == 4 ? (size_t(33) << 10) : (size_t(27) << 10);
            HP h{dl, static_cast<uint32_t>(hl.size()), dc, static_cast<uint32_t>(256 * per_cu * 4), static_cast<uint32_t>(per_xcd), chunk};
            auto step = [&] {
              hipStream_t hs = variant == 2 ? shi : variant == 1 ? sb : sa;
              hipLaunchKernelGGL(heavy_persistent_kernel<4>, dim3(256 * per_cu), dim3(256), pad, hs, p, h);
              if (variant >= 1) hipLaunchKernelGGL(light_blocks_kernel<4>, dim3((gx * gy + 3) / 4), dim3(256), 0, sa, p);
            };
            const int n = 60;
            for (int i = 0; i < 10; ++i) step();
            CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb)); CK(hipStreamSynchronize(shi));
            CK(hipEventRecord(e0, sa));
            for (int i = 0; i < n; ++i) step();
            CK(hipEventRecord(eb, variant == 2 ? shi : sb)); CK(hipStreamWaitEvent(sa, eb, 0));
            CK(hipEventRecord(e1, sa));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            static const char* vn[3] = {"alone", "| light blocks (two streams)", "(high-priority stream) | light blocks"};
            printf("persistent heavy waves, %d x 4 per CU, %s counter, %u tiles per grab, %-40s : %7.2f us\n", per_cu, per_xcd ? "per-XCD" : "one    ", chunk, vn[variant], ms / n * 1e3);
          }
    uint32_t chk[16]; CK(hipMemcpy(chk, dc, 64, hipMemcpyDeviceToHost));
    printf("counters after the runs (all 0 expected): %u %u %u %u\n", chk[0], chk[1], chk[2], chk[9]);
  }
  {
    // (iii) what sits between two kernels of one stream: a chain of all-certain frames (mode 2) on ONE stream, bare; with a
    // satisfied-later cross-stream wait in front of every kernel (a small kernel + event record on a third stream per step, as the
    // product's list builder does); with an event record behind every kernel; with both.
    hipStream_t sl; CK(hipStreamCreateWithFlags(&sl, hipStreamNonBlocking));
    const int NE = 8;
    hipEvent_t ready[NE], rec[NE];
    for (int i = 0; i < NE; ++i) { CK(hipEventCreateWithFlags(&ready[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&rec[i], hipEventDisableTiming)); }
    uint32_t* sig = nullptr;                                            // stream memory operations instead of events (variants 5, 6)
    int can_wait = 0;
    (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0);
    if (can_wait && hipExtMallocWithFlags(reinterpret_cast<void**>(&sig), 64, hipMallocSignalMemory) != hipSuccess) { sig = nullptr; (void)hipGetLastError(); }
    if (sig) CK(hipMemset(sig, 0, 64));
    uint32_t sig_value = 0;
    for (int variant = 0; variant < 7; ++variant) {
      const int n = 100;
      if (variant >= 5 && !sig) { printf("one stream: hipStreamWaitValue32 not available on this device\n"); break; }
      auto step = [&](int i) {
        if (variant >= 5) {                                             // the builder publishes a counter, the trace stream waits for it
          launch_mode(0, 4, p, dim3(gx, variant == 6 ? 34 : 4), sl);
          ++sig_value;
          CK(hipStreamWriteValue32(sl, sig, sig_value, 0));
          CK(hipStreamWaitValue32(sa, sig, sig_value, hipStreamWaitValueGte, 0xFFFFFFFFu));
          launch_mode(2, 4, p, dim3(gx, gy), sa);
          return;
        }
        if (variant == 1 || variant == 3 || variant == 4) {
          launch_mode(0, 4, p, dim3(gx, variant == 4 ? 34 : 4), sl);                    // the "builder" (variant 4: a quarter of the grid)
          CK(hipEventRecord(ready[i % NE], sl));
          CK(hipStreamWaitEvent(sa, ready[i % NE], 0));
        }
        launch_mode(2, 4, p, dim3(gx, gy), sa);
        if (variant == 2 || variant == 3) CK(hipEventRecord(rec[i % NE], sa));
      };
      for (int i = 0; i < 10; ++i) step(i);
      CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sl));
      CK(hipEventRecord(e0, sa));
      for (int i = 0; i < n; ++i) step(i);
      CK(hipEventRecord(e1, sa));
      CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      static const char* vn[7] = {"bare", "cross-stream wait in front of every kernel", "event record behind every kernel", "wait + record", "wait (on a longer builder)",
                                  "hipStreamWaitValue32 in front of every kernel", "hipStreamWaitValue32 (on a longer builder)"};
      printf("one stream, all-certain frames back to back, %-46s : %7.2f us per step\n", vn[variant], ms / n * 1e3);
    }
  }
  {                                                                     // each kind alone
    for (int which = 0; which < 3; ++which) {
      P h = p; h.heavy_only = 1;
      std::vector<uint8_t> all0(kinds.size(), 0);
      uint8_t* dk0; CK(hipMalloc(&dk0, kinds.size())); CK(hipMemset(dk0, 0, kinds.size()));
      P l0 = p; l0.kinds = dk0;                                          // every tile light
      auto step = [&] {
        if (which == 0) launch_mode(3, 4, h, dim3(gx, gy), sa);
        else if (which == 1) hipLaunchKernelGGL(light_blocks_kernel<1>, dim3(gx * gy), dim3(64), 0, sa, p);
        else hipLaunchKernelGGL(light_blocks_kernel<1>, dim3(gx * gy), dim3(64), 0, sa, l0);
      };
      const int n = 60;
      for (int i = 0; i < 10; ++i) step();
      CK(hipStreamSynchronize(sa));
      CK(hipEventRecord(e0, sa));
      for (int i = 0; i < n; ++i) step();
      CK(hipEventRecord(e1, sa));
      CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      static const char* wn[3] = {"heavy tiles only (tile-per-wave kernel, light waves leave)", "light tiles only (light_blocks_kernel)", "ALL tiles light (light_blocks_kernel)"};
      printf("alone: %-60s : %7.2f us\n", wn[which], ms / n * 1e3);
      CK(hipFree(dk0));
    }
  }
  // ---- mode 3 with the tiles visited in a given order (1-D grid over wave slots), linear / tile-major state ----------
  const uint32_t nt = static_cast<uint32_t>(kinds.size());
  std::vector<std::vector<uint32_t>> orders;
  std::vector<const char*> names;
  auto hash = [](uint32_t i, uint32_t n, uint32_t mul) { return static_cast<uint32_t>((static_cast<uint64_t>(i) * mul + 17u) % n); };
  { std::vector<uint32_t> o(nt); for (uint32_t i = 0; i < nt; ++i) o[i] = i; orders.push_back(o); names.push_back("raster (table)"); }
  { std::vector<uint32_t> o(nt); for (uint32_t i = 0; i < nt; ++i) o[i] = hash(i / 4u, nt / 4u, 4099u) * 4u + (i & 3u); orders.push_back(o); names.push_back("blocks hashed"); }
  { std::vector<uint32_t> o(nt); for (uint32_t i = 0; i < nt; ++i) o[i] = hash(i, nt, 16411u); orders.push_back(o); names.push_back("tiles hashed"); }
  std::vector<uint32_t> hv, lt;
  for (uint32_t i = 0; i < nt; ++i) { const uint32_t t = hash(i, nt, 16411u); (kinds[t] ? hv : lt).push_back(t); }
  { std::vector<uint32_t> o = hv; o.insert(o.end(), lt.begin(), lt.end()); orders.push_back(o); names.push_back("by kind: heavy first, then light"); }
  auto interleave = [&](double upto) {                                  // groups of 4 same-kind tiles, heavy groups spread over the first `upto` of the launch
    std::vector<uint32_t> o;
    const size_t gh = (hv.size() + 3) / 4, gl = (lt.size() + 3) / 4, total = gh + gl;
    const size_t span = static_cast<size_t>(total * upto);
    size_t ih = 0, il = 0;
    for (size_t g = 0; g < total; ++g) {
      const bool want_h = ih < gh && (il >= gl || (g < span ? (ih * span <= g * gh) : false) || (total - g) <= (gh - ih));
      std::vector<uint32_t>& src = want_h ? hv : lt;
      size_t& idx = want_h ? ih : il;
      for (size_t k = 0; k < 4; ++k) { const size_t e = idx * 4 + k; o.push_back(e < src.size() ? src[e] : 0xFFFFFFFFu); }
      ++idx;
    }
    return o;
  };
  orders.push_back(interleave(1.0)); names.push_back("same-kind groups of 4, heavy spread over the whole launch");
  orders.push_back(interleave(0.75)); names.push_back("same-kind groups of 4, heavy spread over the first 75 %");
  orders.push_back(interleave(0.5)); names.push_back("same-kind groups of 4, heavy spread over the first 50 %");
  for (size_t oi = 0; oi < orders.size(); ++oi) {
    std::vector<uint32_t>& o = orders[oi];
    for (uint32_t& t : o) if (t == 0xFFFFFFFFu) t = 0;                 // (padding of a partial group: tile 0 again, idempotent here)
    uint32_t* dord; CK(hipMalloc(&dord, o.size() * 4)); CK(hipMemcpy(dord, o.data(), o.size() * 4, hipMemcpyHostToDevice));
    for (int prio = 0; prio <= 1; ++prio)
    for (int tm = 0; tm <= 1; ++tm)
      for (int wgw = 4; wgw >= 1; wgw >>= 1)
        for (int split = 0; split <= 1; ++split) {
          P q = p; q.order = dord; q.tile_major = tm; q.prio = prio;
          const uint32_t total = static_cast<uint32_t>(o.size()), half = (total / 2u) & ~3u;
          auto step = [&] {
            if (!split) { q.slot0 = 0; q.n_slots = total; launch<3>(wgw, q, dim3(1), sa); return; }
            P a = q, b = q; a.slot0 = 0; a.n_slots = half; b.slot0 = half; b.n_slots = total - half;
            launch<3>(wgw, a, dim3(1), sa); launch<3>(wgw, b, dim3(1), sb);
          };
          const int n = 60;
          for (int i = 0; i < 10; ++i) step();
          CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
          CK(hipEventRecord(e0, sa));
          for (int i = 0; i < n; ++i) step();
          CK(hipEventRecord(eb, sb)); CK(hipStreamWaitEvent(sa, eb, 0));
          CK(hipEventRecord(e1, sa));
          CK(hipEventSynchronize(e1));
          float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
          printf("order %-62s %s %s  %d-wave wg  %s : %7.2f us\n", names[oi], prio ? "setprio" : "       ", tm ? "tile-major" : "linear    ", wgw, split ? "2 kernels" : "1 kernel ", ms / n * 1e3);
        }
    CK(hipFree(dord));
  }
  return 0;
}
