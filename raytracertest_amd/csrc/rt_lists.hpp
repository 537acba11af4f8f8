// rt_lists.hpp -- the kernels that BUILD candidate lists ahead of the trace kernels: macro_bin_kernel (dense scenes: the
// macro tiles' lists), tile_lists_kernel / region_lists_kernel (small scenes: every tile's list + certain-winner verdict).
// Included through rt_trace.hpp.
#pragma once
#include "rt_classify.hpp"

namespace rtk {

// ------------------------------------------------------------------------------------
// Macro level of the triangle classification (scenes larger than the per-wave list).
// One block per macro tile of macro_w x macro_h pixels (whole trace blocks): the focal points of
// ALL its pixels, computed exactly as the trace kernel computes them, give the macro tile's ray
// family; every triangle the family certainly misses is dropped, the survivors' indices are
// written in ascending order.  The trace kernel's blocks then pre-cull their macro tile's list
// instead of the whole scene (C4: ~40 steps of 256 triangles per block -> 1-2).  Same
// conservative test as the block and wave levels, so the result stays bit-identical to the
// full scan.  Runs once per launch (the camera may have changed): N x macro tiles tests.
// ------------------------------------------------------------------------------------
// Focal bounds of the pixel rectangle [x0, x0 + rw) x [y0, y0 + rh) of the band (clipped to it), by all 64 * WAVES
// threads of the block: every pixel's focal point exactly as the trace kernel computes it.  s_box: WAVES x 8 floats of LDS.
template <bool FMA, int WAVES>
__device__ __forceinline__ FocalBounds rect_focal_bounds(const TraceParams& p, uint32_t x0, uint32_t y0, uint32_t rw, uint32_t rh, float (*s_box)[8]) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t x1 = (x0 + rw < p.W) ? x0 + rw : p.W;
  const uint32_t y1 = (y0 + rh < p.rows) ? y0 + rh : p.rows;
  const uint32_t w = x1 - x0, count_px = w * (y1 - y0);
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  bool finite = true;
  for (uint32_t i = threadIdx.x; i < count_px; i += 64u * WAVES) {
    const uint32_t px = x0 + i % w, ly = y0 + i / w;
    V3 po, pd;
    pinhole<FMA>(p, px, p.row0 + ly, po, pd);
    const V3 f = focal_point<FMA>(p, pd);
    const float fl[3] = {f.x, f.y, f.z};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      finite = finite && (__builtin_fabsf(fl[c]) <= FLT_MAX);
      lo[c] = fminf(lo[c], fl[c]);
      hi[c] = fmaxf(hi[c], fl[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) { lo[c] = uniform(wave_min(lo[c])); hi[c] = uniform(wave_max(hi[c])); }
  const bool wave_ok = __builtin_amdgcn_ballot_w64(!finite) == 0ull;
  if (lane == 0u) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { s_box[wave][c] = lo[c]; s_box[wave][3 + c] = hi[c]; }
    s_box[wave][6] = wave_ok ? 1.0f : 0.0f;
  }
  __syncthreads();
  FocalBounds bb;
  bb.ok = true; bb.any = count_px != 0u;
#pragma unroll
  for (int c = 0; c < 3; ++c) { bb.lo[c] = FLT_MAX; bb.hi[c] = -FLT_MAX; }
  for (uint32_t v = 0; v < static_cast<uint32_t>(WAVES); ++v) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      bb.lo[c] = fminf(bb.lo[c], s_box[v][c]);
      bb.hi[c] = fmaxf(bb.hi[c], s_box[v][3 + c]);
    }
    bb.ok = bb.ok && (s_box[v][6] != 0.0f);
  }
  return bb;
}

// One pass of a bin kernel (256 threads): the triangles tri_of(0 .. n) against `fam`, the survivors in ascending order to
// out[1..], their number (0xFFFFFFFF: more than cap) to out[0].  s_cnt: 2 x 4 words of LDS.
template <class Src>
__device__ __forceinline__ void bin_pass(const TraceParams& p, const TileFamily& fam, Src tri_of, uint32_t n, uint32_t* out, uint32_t cap,
                                         uint32_t (*s_cnt)[4]) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t total = 0, step = 0;
  bool overflow = false;
  for (uint32_t base = 0; base < n; base += 256u, ++step) {
    const uint32_t e = base + threadIdx.x;
    const bool valid = e < n;
    const uint32_t tri = tri_of(valid ? e : (n - 1u));
    const float4 A0 = p.tri_a[2u * tri], A1 = p.tri_a[2u * tri + 1u];
    const float bz = p.tri_b[tri];
    bool keep = valid;
    if (fam.usable)
      keep = valid && !tile_misses_triangle(fam, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z});
    const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
    uint32_t* const slot = s_cnt[step & 1u];                     // double-buffered: one barrier per step
    if (lane == 0u) slot[wave] = static_cast<uint32_t>(__builtin_popcountll(m));
    __syncthreads();
    const uint32_t c0 = slot[0], c1 = slot[1], c2 = slot[2], c3 = slot[3];
    const uint32_t before = (wave > 0u ? c0 : 0u) + (wave > 1u ? c1 : 0u) + (wave > 2u ? c2 : 0u);
    const uint32_t step_total = c0 + c1 + c2 + c3;
    if (total + step_total > cap) { overflow = true; break; }     // block-uniform
    const uint32_t pos = total + before + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                                __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
    if (keep) out[1u + pos] = tri;                                // ascending order across waves and steps
    total += step_total;
  }
  if (threadIdx.x == 0u) out[0] = overflow ? 0xFFFFFFFFu : total;
}

// ---- the level above the macro tiles (dense scenes; TraceParams::super_lists) ------------------------------------------
// A macro tile's list costs one test per triangle of the scene: 1 020 macro tiles x 10 000 triangles at C4, 72 M
// instructions and -- 40 dependent steps per block -- 0.16 ms at the head of every half-launch's build chain.  Three short,
// wide kernels instead:
//   macro_bounds_kernel   every macro tile's focal box (all its pixels, as before) to p.macro_bounds;
//   super_bin_kernel      a super tile = super_f x super_f macro tiles, its box the union of theirs; block (super tile, chunk)
//                         tests the chunk's kSuperChunk triangles against the super tile's family and writes its own short
//                         list: a super tile's list is the concatenation of its chunks' lists (ascending: so are the chunks);
//   macro_bin_kernel      reads its box, walks its super tile's chunk lists.
// The super tile's family contains the macro tile's, and what tile_misses_triangle drops for a family no ray of the family can
// hit: whatever the super level drops, the macro level could only have kept in vain.

__device__ __forceinline__ FocalBounds load_bounds(const float* b) {
  FocalBounds r;
#pragma unroll
  for (int c = 0; c < 3; ++c) { r.lo[c] = uniform(b[c]); r.hi[c] = uniform(b[3 + c]); }
  r.ok = uniform(b[6]) != 0.0f; r.any = uniform(b[7]) != 0.0f;
  return r;
}

template <bool FMA>
__global__ __launch_bounds__(256) void macro_bounds_kernel(TraceParams p) {
  __shared__ float s_box[4][8];
  const FocalBounds bb = rect_focal_bounds<FMA, 4>(p, blockIdx.x * p.macro_w, blockIdx.y * p.macro_h, p.macro_w, p.macro_h, s_box);
  if (threadIdx.x == 0u) {
    float* const o = p.macro_bounds + (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 8u;
#pragma unroll
    for (int c = 0; c < 3; ++c) { o[c] = bb.lo[c]; o[3 + c] = bb.hi[c]; }
    o[6] = bb.ok ? 1.0f : 0.0f; o[7] = bb.any ? 1.0f : 0.0f;
  }
}

template <bool FMA>
__global__ __launch_bounds__(256) void super_bin_kernel(TraceParams p) {     // grid = (super tiles, chunks)
  __shared__ uint32_t s_cnt[2][4];
  const uint32_t sx = blockIdx.x % p.super_nx, sy = blockIdx.x / p.super_nx;
  const uint32_t macro_ny = (p.rows + p.macro_h - 1u) / p.macro_h;
  FocalBounds bb;
  bb.ok = true; bb.any = false;
#pragma unroll
  for (int c = 0; c < 3; ++c) { bb.lo[c] = FLT_MAX; bb.hi[c] = -FLT_MAX; }
  for (uint32_t j = 0; j < p.super_f; ++j) {
    for (uint32_t i = 0; i < p.super_f; ++i) {
      const uint32_t mx = sx * p.super_f + i, my = sy * p.super_f + j;
      if (mx >= p.macro_nx || my >= macro_ny) continue;                  // (block-uniform)
      const FocalBounds mb = load_bounds(p.macro_bounds + (static_cast<size_t>(my) * p.macro_nx + mx) * 8u);
#pragma unroll
      for (int c = 0; c < 3; ++c) { bb.lo[c] = fminf(bb.lo[c], mb.lo[c]); bb.hi[c] = fmaxf(bb.hi[c], mb.hi[c]); }
      bb.ok = bb.ok && mb.ok; bb.any = bb.any || mb.any;
    }
  }
  const TileFamily fam = make_family(p, bb);
  const uint32_t first = blockIdx.y * kSuperChunk;
  const uint32_t n = p.n_tris - first < kSuperChunk ? p.n_tris - first : kSuperChunk;
  uint32_t* const out = p.super_lists + (static_cast<size_t>(blockIdx.x) * gridDim.y + blockIdx.y) * (kSuperChunk + 1u);
  bin_pass(p, fam, [&](uint32_t e) { return first + e; }, n, out, kSuperChunk, s_cnt);
}

template <bool FMA>
__global__ __launch_bounds__(256) void macro_bin_kernel(TraceParams p) {
  __shared__ float s_box[4][8];
  __shared__ uint32_t s_cnt[2][4];
  __shared__ uint32_t s_first[kSuperMaxChunks + 1u];                    // super level: where each chunk's list starts in the concatenation
  FocalBounds bb;
  if (p.macro_bounds != nullptr) bb = load_bounds(p.macro_bounds + (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 8u);
  else bb = rect_focal_bounds<FMA, 4>(p, blockIdx.x * p.macro_w, blockIdx.y * p.macro_h, p.macro_w, p.macro_h, s_box);
  const TileFamily fam = make_family(p, bb);
  uint32_t* const out = p.macro_lists + (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * (p.macro_cap + 1u);
  if (p.super_lists == nullptr) {
    bin_pass(p, fam, [](uint32_t e) { return e; }, p.n_tris, out, p.macro_cap, s_cnt);
    return;
  }
  const uint32_t S = p.super_chunks;
  const uint32_t* const sl = p.super_lists + static_cast<size_t>((blockIdx.y / p.super_f) * p.super_nx + blockIdx.x / p.super_f) * S * (kSuperChunk + 1u);
  if (threadIdx.x == 0u) {
    uint32_t acc = 0;
    for (uint32_t c = 0; c < S; ++c) { s_first[c] = acc; acc += sl[static_cast<size_t>(c) * (kSuperChunk + 1u)]; }
    s_first[S] = acc;
  }
  __syncthreads();
  const uint32_t n = s_first[S];
  bin_pass(p, fam, [&](uint32_t e) {
    uint32_t c = 0;
    for (uint32_t k = 1; k < S; ++k) c += (e >= s_first[k]) ? 1u : 0u;   // (the chunks' starts ascend)
    return sl[static_cast<size_t>(c) * (kSuperChunk + 1u) + 1u + (e - s_first[c])];
  }, n, out, p.macro_cap, s_cnt);
}

// ------------------------------------------------------------------------------------
// Small scenes (no more triangles than the per-wave list holds): the tiles' candidate lists and certain-winner
// verdicts are built by a kernel of their own, ahead of the trace launch that needs them.
//
// A tile's list depends on the camera, the scene and the frame, not on the samples: it is a camera-dependent
// acceleration structure, and building it needs neither RNG states nor rays -- only the tile's ray family, i.e. the focal
// points of its four corner pixels (full tiles; every in-image pixel otherwise), and one pass of tile_misses_triangle with
// lane = triangle.  Inside the trace kernel that pass ran once per wave with half of its lanes idle (C3: 32 triangles),
// behind a 64-pixel pinhole pass, and held the trace kernel's registers while it did: 675 of a C3 tile's 2 565
// instructions, 58 % of what a certain-winner tile costs.  Here G = 32 or 64 lanes own one tile (two tiles per wave for
// scenes of up to 32 triangles), the trace kernel's small-scene instantiations contain no classification code at all and
// a tile with a certain winner generates no pinhole ray either.  The lists are the same ones the wave would have built --
// same focal_bounds arithmetic (corner path or the range over the in-image pixels), same make_family, same
// tile_misses_triangle<.., SURE> -- so nothing a trace computes changes (rt_dbg_classify is that same code, checked
// verdict by verdict against the reference's arithmetic: tests/test_gpu_classification.py).
//
// The certain-winner verdict now spans classification steps (scenes of 65 ... 256 triangles): per tile the running
// winner A = the kept, certainly-hit triangle with the largest lower bound of q (first in scan order on ties) and the
// two largest upper bounds of q over the kept triangles, so that R = the largest upper bound over the kept triangles
// other than A is known at the end; the rule itself is unchanged (A alone, or R < Q - 1e-4 (|R| + |Q|)).
//
// Per tile slot (grid order of the trace launch, 4 per 32x8 block): word 0 = count | winner << 10 | certain << 31,
// then the kept triangle indices, ascending.  grid = ceil(slots / (4 * (64 / G))) blocks of 256 threads.
// ------------------------------------------------------------------------------------
// Max / min over the G lanes of a group, in every lane.  The list builder is latency-bound (one dependent chain per wave,
// few waves per SIMD), unlike the VALU-issue-bound trace kernel: inside a row of 16 lanes the butterfly runs on DPP
// (quad_perm [1,0,3,2], [2,3,0,1], row_ror:4, row_ror:8 -- VALU latency, no LDS round trip), only the steps across rows go
// through ds_bpermute.  Every lane of the wave is active here.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
#define RT_ROW_REDUCE(OP, v)                 \
  v = OP(v, dpp_f<0xB1>(v));                 \
  v = OP(v, dpp_f<0x4E>(v));                 \
  v = OP(v, dpp_f<0x124>(v));                \
  v = OP(v, dpp_f<0x128>(v));
template <int G>
__device__ __forceinline__ float group_max(float v) {
  RT_ROW_REDUCE(fmaxf, v)
#pragma unroll
  for (int off = G / 2; off >= 16; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
template <int G>
__device__ __forceinline__ float group_min(float v) {
  RT_ROW_REDUCE(fminf, v)
#pragma unroll
  for (int off = G / 2; off >= 16; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}
#undef RT_ROW_REDUCE

// The focal bounds of the 8x8 tile at (x0, y0) of the band for the G lanes that own it (gl = lane within the group, gbase =
// its first lane): full tiles take the four corner pixels' focal points -- computed exactly as the rays use them -- widened
// by what a focal point of the tile can lie off the corners' bilinear interpolant (p.tile_curv, host) and by the roundings
// of the evaluations (see focal_bounds, whose corner path this is); partial tiles (image edge) and cameras the host does not
// vouch for (p.tile_curv <= 0) take the range over their in-image pixels.  Group-uniform result.
template <bool FMA, int G, class SL>
__device__ __forceinline__ FocalBounds group_focal_bounds(const TraceParams& p, uint32_t x0, uint32_t y0, bool in_image, uint32_t gl, uint32_t gbase) {
  constexpr uint32_t T = 64u / G;
  const bool full = in_image && x0 + 8u <= p.W && y0 + 8u <= p.rows;
  FocalBounds b;
  const bool corners = p.tile_curv > 0.0f && full;                  // group-uniform
  const unsigned long long need_range = __builtin_amdgcn_ballot_w64(in_image && !corners);
  {
    // corner path: lanes 0..3 of the group take the pixels (x0, y0), (x0 + 7, y0), (x0, y0 + 7), (x0 + 7, y0 + 7)
    const uint32_t cx = x0 + ((gl & 1u) ? 7u : 0u), cy = y0 + ((gl & 2u) ? 7u : 0u);
    V3 po, pd;
    pinhole<FMA>(p, cx < p.W ? cx : 0u, p.row0 + (cy < p.rows ? cy : 0u), po, pd);
    const V3 f = focal_point<FMA>(p, pd);
    const float fl[3] = {f.x, f.y, f.z};
    const float dev = p.tile_curv + RT_SLK(4e-6f) * p.tile_round;
    bool finite = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float c0 = __shfl(fl[i], static_cast<int>(gbase), 64), c1 = __shfl(fl[i], static_cast<int>(gbase + 1u), 64);
      const float c2 = __shfl(fl[i], static_cast<int>(gbase + 2u), 64), c3 = __shfl(fl[i], static_cast<int>(gbase + 3u), 64);
      b.lo[i] = fminf(fminf(c0, c1), fminf(c2, c3)) - dev;
      b.hi[i] = fmaxf(fmaxf(c0, c1), fmaxf(c2, c3)) + dev;
      // (the trace wave asks every in-image lane for a finite focal point; of a full tile's 64 monotone-bounded points
      // the corners' range +- dev is finite iff they are: lo/hi are checked instead, make_family drops to "keep all")
      finite = finite && (__builtin_fabsf(b.lo[i]) <= FLT_MAX) && (__builtin_fabsf(b.hi[i]) <= FLT_MAX);
    }
    b.ok = finite; b.any = true;
  }
  if (need_range != 0ull) {                                         // wave-uniform: some group of this wave has a partial tile
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    bool finite = true, any = false;
#pragma unroll
    for (uint32_t i = 0; i < T; ++i) {                              // the 64 pixels of the tile, G at a time
      const uint32_t pi = gl + i * G;
      const uint32_t px = x0 + (pi & 7u), py = y0 + (pi >> 3);
      const bool inside = in_image && px < p.W && py < p.rows;
      V3 po, pd;
      pinhole<FMA>(p, inside ? px : 0u, p.row0 + (inside ? py : 0u), po, pd);
      const V3 f = focal_point<FMA>(p, pd);
      const float fl[3] = {f.x, f.y, f.z};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (inside) { lo[c] = fminf(lo[c], fl[c]); hi[c] = fmaxf(hi[c], fl[c]); finite = finite && (__builtin_fabsf(fl[c]) <= FLT_MAX); }
      }
      any = any || inside;
    }
    const unsigned long long gmask = G == 64 ? ~0ull : (0xFFFFFFFFull << gbase);
    const bool g_ok = (__builtin_amdgcn_ballot_w64(!finite) & gmask) == 0ull;
    const bool g_any = (__builtin_amdgcn_ballot_w64(any) & gmask) != 0ull;
#pragma unroll
    for (int c = 0; c < 3; ++c) { lo[c] = group_min<G>(lo[c]); hi[c] = group_max<G>(hi[c]); }
    if (!corners) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { b.lo[c] = lo[c]; b.hi[c] = hi[c]; }
      b.ok = g_ok; b.any = g_any;
    }
  }
  return b;
}

template <bool FMA, int G, class SL = SlackProduct>
__global__ __launch_bounds__(256, 5) void tile_lists_kernel(const TraceParams p) {
  static_assert(G == 32 || G == 64, "lanes per tile");
  constexpr uint32_t T = 64u / G;                                   // tiles per wave
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t gl = lane & (G - 1u), gbase = lane & ~(G - 1u);    // lane within its group, first lane of the group
  const uint32_t gx = (p.W + 31u) / 32u, gy = (p.rows + 7u) / 8u;
  const uint32_t slots = gx * gy * 4u;
  const uint32_t slot = (blockIdx.x * 4u + wave) * T + lane / G;
  const bool live = slot < slots;                                   // (group-uniform)
  const uint32_t sl = live ? slot : 0u;
  const uint32_t x0 = ((sl / 4u) % gx) * 32u + (sl % 4u) * 8u, y0 = ((sl / 4u) / gx) * 8u;
  const bool in_image = live && x0 < p.W;                           // a slot right of the image has no pixel: empty list

  const FocalBounds b = group_focal_bounds<FMA, G, SL>(p, x0, y0, in_image, gl, gbase);
  const TileFamily fam = make_family<SL>(p, b);

  // ---- classification, lane = triangle, G triangles per step ---------------------------------------------------------
  uint32_t* const saved = p.tile_lists + static_cast<size_t>(sl) * (1u + p.bin_list);
  const uint32_t n = p.n_tris;
  uint32_t count = 0;
  const float NEG = -__builtin_inff();
  bool haveA = false;
  float Q = NEG, M1 = NEG, M2 = NEG;
  uint32_t A = 0, I1 = 0xFFFFFFFFu;
  for (uint32_t base = 0; base < n; base += G) {                    // (wave-uniform trip count)
    const uint32_t tri = base + gl;
    const bool valid = in_image && tri < n;
    const uint32_t ti = tri < n ? tri : n - 1u;
    const float4 A0 = p.tri_a[2u * ti], A1 = p.tri_a[2u * ti + 1u];
    const float bz = p.tri_b[ti];
    bool keep = valid, sure = false;
    float q[2] = {0.0f, 0.0f};
    if (fam.usable) {                                               // (per lane: group-uniform)
      const bool miss = tile_misses_triangle<false, true, SL>(fam, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z}, nullptr, &sure, q);
      keep = valid && !miss;
    } else {
      sure = false;
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
    const unsigned long long gm = G == 64 ? m : ((m >> gbase) & 0xFFFFFFFFull);
    const uint32_t pos = count + static_cast<uint32_t>(__builtin_popcountll(gm & ((1ull << gl) - 1ull)));
    if (keep) saved[1u + pos] = tri;                                // ascending order
    count += static_cast<uint32_t>(__builtin_popcountll(gm));
    // running winner: the certainly-hit candidate with the largest lower bound of q, first in scan order on ties
    const bool cand = keep && sure && fam.usable;
    const float Qs = group_max<G>(cand ? q[0] : NEG);
    const unsigned long long bm = __builtin_amdgcn_ballot_w64(cand && q[0] == Qs);
    const unsigned long long gbm = G == 64 ? bm : ((bm >> gbase) & 0xFFFFFFFFull);
    if (gbm != 0ull && (!haveA || Qs > Q)) { haveA = true; Q = Qs; A = base + static_cast<uint32_t>(__builtin_ctzll(gbm)); }
    // the two largest upper bounds of q over the kept triangles (a NaN bound is no bound)
    const float qh = keep ? ((q[1] == q[1]) ? q[1] : __builtin_inff()) : NEG;
    if (n <= G) {                                                    // one step (wave-uniform): A is final, R directly
      const uint32_t la = gbm != 0ull ? static_cast<uint32_t>(__builtin_ctzll(gbm)) : 0xFFFFFFFFu;
      M1 = group_max<G>((keep && gl != la) ? qh : NEG);
      I1 = 0xFFFFFFFEu;                                              // "not A": R = M1 below
      break;
    }
    const float m1s = group_max<G>(qh);
    const unsigned long long tm = __builtin_amdgcn_ballot_w64(keep && qh == m1s);
    const unsigned long long gtm = G == 64 ? tm : ((tm >> gbase) & 0xFFFFFFFFull);
    const uint32_t l1 = gtm != 0ull ? static_cast<uint32_t>(__builtin_ctzll(gtm)) : 0xFFFFFFFFu;
    const float m2s = group_max<G>((keep && gl != l1) ? qh : NEG);
    if (gtm != 0ull) {
      if (m1s > M1) { M2 = fmaxf(M1, m2s); M1 = m1s; I1 = base + l1; }
      else { M2 = fmaxf(M2, m1s); }
    }
  }
  bool sure_one = false;
  if (haveA) {
    const float R = (I1 == A) ? M2 : M1;                            // the largest upper bound among the OTHER kept triangles
    sure_one = count <= 1u || (R < Q - 1e-4f * (__builtin_fabsf(R) + __builtin_fabsf(Q)));
  }
  if (live && gl == 0u) saved[0] = count | (A << 10) | (sure_one ? 0x80000000u : 0u);
}

// ------------------------------------------------------------------------------------
// The same lists, built in two levels (the default whenever the host vouches for the corner bound, p.tile_curv > 0).
//
// tile_lists_kernel above spends one full classification (~260 instructions on G lanes) per tile although a C3 tile keeps
// 1.2 of the 32 triangles: 11 M wave-instructions per 1080p frame, latency-bound, a fifth of what the trace itself costs.
// Here one wave owns a REGION of 4 x 2 tiles (32 x 16 pixels, two stacked trace blocks):
//   1. one pinhole pass gives the focal points of all 32 tile-corner pixels (lane = tile * 4 + corner; a tile clipped by the
//      image edge takes the corners of its in-image rectangle: cx, cy are monotone in the pixel index and the curvature term of
//      a smaller rectangle is smaller, so the same allowance p.tile_curv bounds it); quad-wide DPP min/max give every tile its
//      focal box, a row reduction their union = the region's box;
//   2. level 1, lane = triangle: the whole scene against the REGION's family (a superset of every tile's family, so whatever
//      it drops no ray of any of the tiles can hit); survivors, ascending, to a per-wave list in LDS (C3: ~3 of 32);
//   3. level 2, lane = tile * 8 + candidate: 8 region candidates per pass against each of the 8 tiles' own families, with the
//      certain-winner bounds; per-tile compaction (8-lane groups), the running winner / top-two bookkeeping of
//      tile_lists_kernel, the list and its header word to the tile's slot.
// ~130 instead of ~350 instructions per tile, a quarter of the waves.  A tile's list is a subset of what the one-level
// build keeps (both are conservative: the image cannot tell them apart); rt_dbg_classify exports both levels' verdicts
// (level 0: the tile, all triangles; level 3: the region) and tests/test_gpu_classification.py checks the lists the product
// really stored against the reference's per-ray arithmetic.
// grid = ceil(regions / 4) blocks of 256 threads, regions = ceil(W / 32) * ceil(ceil(rows / 8) / 2).
// ------------------------------------------------------------------------------------
template <class SL>
__device__ __forceinline__ float tile_dev(const TraceParams& p) { return p.tile_curv + RT_SLK(4e-6f) * p.tile_round; }

// lane -> (tile of the region, corner): the corner pixel of the tile's in-image rectangle, band-local; valid = tile in the band
__device__ __forceinline__ void region_corner_pixel(const TraceParams& p, uint32_t rx, uint32_t ry, uint32_t tile, uint32_t corner,
                                                    uint32_t& px, uint32_t& py, bool& valid) {
  const uint32_t x0 = rx * 32u + (tile & 3u) * 8u, y0 = (ry * 2u + (tile >> 2)) * 8u;
  valid = x0 < p.W && y0 < p.rows;
  const uint32_t x1 = (x0 + 7u < p.W) ? x0 + 7u : p.W - 1u, y1 = (y0 + 7u < p.rows) ? y0 + 7u : p.rows - 1u;
  px = valid ? ((corner & 1u) ? x1 : x0) : 0u;
  py = valid ? ((corner & 2u) ? y1 : y0) : 0u;
}

// Focal boxes of the 8 tiles of region (rx, ry) -- in lanes 4 t .. 4 t + 3 of both half-waves -- and their union (every lane).
// PACK: the two half-waves own different regions (rx, ry differ between them): every quantity is taken per half-wave.
template <bool FMA, class SL, bool PACK = false>
__device__ __forceinline__ void region_focal_bounds(const TraceParams& p, uint32_t rx, uint32_t ry, uint32_t lane,
                                                    FocalBounds& tile_b, FocalBounds& region_b) {
  const uint32_t l32 = lane & 31u;
  uint32_t px, py;
  bool valid;
  region_corner_pixel(p, rx, ry, l32 >> 2, l32 & 3u, px, py, valid);
  V3 po, pd;
  pinhole<FMA>(p, px, p.row0 + py, po, pd);
  const V3 f = focal_point<FMA>(p, pd);
  const float fl[3] = {f.x, f.y, f.z};
  const float dev = tile_dev<SL>(p);
  bool fin = true;
  float rlo[3], rhi[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float lo = fl[i], hi = fl[i];
    lo = fminf(lo, dpp_f<0xB1>(lo)); lo = fminf(lo, dpp_f<0x4E>(lo));          // the tile's four corners: quad-wide
    hi = fmaxf(hi, dpp_f<0xB1>(hi)); hi = fmaxf(hi, dpp_f<0x4E>(hi));
    tile_b.lo[i] = lo - dev;
    tile_b.hi[i] = hi + dev;
    fin = fin && (__builtin_fabsf(tile_b.lo[i]) <= FLT_MAX) && (__builtin_fabsf(tile_b.hi[i]) <= FLT_MAX) && (fl[i] == fl[i]);
    rlo[i] = valid ? tile_b.lo[i] : FLT_MAX;
    rhi[i] = valid ? tile_b.hi[i] : -FLT_MAX;
  }
  // (a NaN corner makes lo/hi of its quad NaN-free through fmin/fmax: the corners themselves are asked, quad-wide)
  const unsigned long long badm = __builtin_amdgcn_ballot_w64(!fin);
  const uint32_t quad_bad = (static_cast<uint32_t>(badm >> (lane & (PACK ? 60u : 28u))) & 0xFu);   // (!PACK: lanes 0..31 mirror 32..63)
  tile_b.ok = quad_bad == 0u;
  tile_b.any = valid;
  const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
  const uint32_t hs = PACK ? (lane & 32u) : 0u;                                                   // this half-wave's bits
  region_b.any = ((vm >> hs) & 0xFFFFFFFFull) != 0ull;
  region_b.ok = (((badm & vm) >> hs) & 0xFFFFFFFFull) == 0ull;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    region_b.lo[i] = group_min<32>(rlo[i]);
    region_b.hi[i] = group_max<32>(rhi[i]);
  }
}


// G-lane groups (G = 8 or 4): max in every lane of the group (quad_perm x 2, row_half_mirror)
template <int G>
__device__ __forceinline__ float max_group(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  if constexpr (G == 8) v = fmaxf(v, dpp_f<0x141>(v));
  return v;
}

// Level 2 of the region builder: G2 candidates per pass against the family of the tile this lane's G2-lane group owns, with the
// certain-winner bounds; per-tile compaction, the running winner / top-two bookkeeping of tile_lists_kernel, the list and its
// header word to the tile's slot.  mine: the tile's focal box; cand / cnt: the candidate list of the tile's REGION (the same in
// every lane of a group; cnt_max >= cnt wave-uniform: the trip count); tslot / slot_live / lower: where the tile's list goes.
template <class SL, int G2>
__device__ __forceinline__ void region_level2(const TraceParams& p, uint32_t lane, const FocalBounds& mine, const uint32_t* cand, uint32_t cnt,
                                              uint32_t cnt_max, uint32_t tslot, bool slot_live, bool lower) {
  constexpr uint32_t GM = (1u << G2) - 1u;
  const uint32_t j = lane & (G2 - 1u), gsh = lane & ~static_cast<uint32_t>(G2 - 1);              // slot within the group, its first lane
  const bool tile_valid = mine.any;                                  // the tile has pixels in the band
  const TileFamily fam = make_family<SL>(p, mine);
  uint32_t* const saved = p.tile_lists + static_cast<size_t>(tile_valid ? tslot : 0u) * (1u + p.bin_list);
  const float NEG = -__builtin_inff();
  uint32_t count = 0;
  bool haveA = false;
  float Q = NEG, M1 = NEG, M2 = NEG;
  uint32_t A = 0, I1 = 0xFFFFFFFFu;
  const bool one_pass = cnt_max <= static_cast<uint32_t>(G2);        // every candidate of the region has a lane: pairs can be compared
  bool others_ok = true;                                             // one_pass: every other kept triangle is certainly nearer than A
  for (uint32_t c0 = 0; c0 < cnt_max; c0 += G2) {                    // (wave-uniform trip count)
    const bool has = tile_valid && c0 + j < cnt;
    const uint32_t tri = (c0 + j < cnt) ? cand[c0 + j] : 0u;         // (a lane without a candidate takes triangle 0: n_tris >= 1 here)
    const float4 A0 = p.tri_a[2u * tri], A1 = p.tri_a[2u * tri + 1u];
    const float bz = p.tri_b[tri];
    bool keep = has, sure = false;
    float q[2] = {0.0f, 0.0f};
    float pr[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (fam.usable) {
      const bool miss = tile_misses_triangle<false, true, SL>(fam, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z}, nullptr, &sure, q,
                                                              nullptr, pr);
      keep = has && !miss;
    }
    const uint32_t gm = static_cast<uint32_t>(__builtin_amdgcn_ballot_w64(keep) >> gsh) & GM;
    const uint32_t pos = count + static_cast<uint32_t>(__builtin_popcount(gm & ((1u << j) - 1u)));
    if (keep) saved[1u + pos] = tri;                                 // ascending: candidates and passes ascend
    count += static_cast<uint32_t>(__builtin_popcount(gm));
    const bool cd = keep && sure && fam.usable;
    const float Qs = max_group<G2>(cd ? q[0] : NEG);
    const uint32_t gbm = static_cast<uint32_t>(__builtin_amdgcn_ballot_w64(cd && q[0] == Qs) >> gsh) & GM;
    if (gbm != 0u && (!haveA || Qs > Q)) { haveA = true; Q = Qs; A = cand[c0 + static_cast<uint32_t>(__builtin_ctz(gbm))]; }
    const float qh = keep ? ((q[1] == q[1]) ? q[1] : __builtin_inff()) : NEG;
    if (one_pass) {
      // A's rivals one by one: nearer by the q intervals, or -- where those overlap -- by the pairwise bound (pair_farther)
      const uint32_t la = gbm != 0u ? static_cast<uint32_t>(__builtin_ctz(gbm)) : 0xFFFFFFFFu;
      bool lane_ok = !keep || j == la || (gbm != 0u && qh < Qs - 1e-4f * (__builtin_fabsf(qh) + __builtin_fabsf(Qs)));
      const bool need = gbm != 0u && !lane_ok;
      if (__builtin_amdgcn_ballot_w64(need) != 0ull) {               // (wave-uniform: rare)
        float pa[9];
        const int src = static_cast<int>(gsh + (la & (G2 - 1u)));
#pragma unroll
        for (int i = 0; i < 9; ++i) pa[i] = __shfl(pr[i], src, 64);
        lane_ok = lane_ok || (need && pair_farther<SL>(fam, pa, pr));
      }
      others_ok = (static_cast<uint32_t>(__builtin_amdgcn_ballot_w64(!lane_ok) >> gsh) & GM) == 0u;
    }
    const float m1s = max_group<G2>(qh);
    const uint32_t gtm = static_cast<uint32_t>(__builtin_amdgcn_ballot_w64(keep && qh == m1s) >> gsh) & GM;
    const uint32_t l1 = gtm != 0u ? static_cast<uint32_t>(__builtin_ctz(gtm)) : 0xFFFFFFFFu;
    const float m2s = max_group<G2>((keep && j != l1) ? qh : NEG);
    if (gtm != 0u) {
      const uint32_t i1s = cand[c0 + l1];
      if (m1s > M1) { M2 = fmaxf(M1, m2s); M1 = m1s; I1 = i1s; }
      else { M2 = fmaxf(M2, m1s); }
    }
  }
  bool sure_one = false;
  if (haveA) {
    const float R = (I1 == A) ? M2 : M1;                             // the largest upper bound among the OTHER kept triangles
    sure_one = count <= 1u || (R < Q - 1e-4f * (__builtin_fabsf(R) + __builtin_fabsf(Q)));
    if (one_pass) sure_one = sure_one || others_ok;                  // (others_ok alone would do: the line above is what it generalises)
  }
  if (j == 0u) {
    if (tile_valid) saved[0] = count | (A << 10) | (sure_one ? 0x80000000u : 0u);
    else if (slot_live) p.tile_lists[static_cast<size_t>(tslot) * (1u + p.bin_list)] = 0u;
  }
  if (p.half_cost != nullptr) {                                      // tiles that will generate rays, per half of the band
    const bool rays = j == 0u && tile_valid && !sure_one;
    const uint32_t nu = static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(rays && !lower)));
    const uint32_t nl = static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(rays && lower)));
    if (lane == 0u) {
      if (nu != 0u) atomicAdd(p.half_cost, nu);
      if (nl != 0u) atomicAdd(p.half_cost + 1, nl);
    }
  }
}

template <bool FMA, class SL = SlackProduct>
__global__ __launch_bounds__(256, RT_LISTS_WAVES) void region_lists_kernel(const TraceParams p) {
  __shared__ uint32_t s_cand[4][256];                               // per wave: the region's candidates, ascending
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t gx = (p.W + 31u) / 32u, gy = (p.rows + 7u) / 8u, gry = (gy + 1u) / 2u;
  const uint32_t region = blockIdx.x * 4u + wave;
  if (region >= gx * gry) return;                                   // wave-uniform
  const uint32_t rx = region % gx, ry = region / gx;
  FocalBounds tb, rb;
  region_focal_bounds<FMA, SL>(p, rx, ry, lane, tb, rb);
  const uint32_t n = p.n_tris;
  uint32_t* const cand = s_cand[wave];

  // ---- level 1: the scene against the region's family, lane = triangle ---------------------------------------------
  uint32_t cnt = 0;
  {
    const TileFamily rf = make_family<SL>(p, rb);
    for (uint32_t base = 0; base < n; base += 64u) {
      const uint32_t tri = base + lane;
      const bool valid = tri < n;
      const uint32_t ti = valid ? tri : n - 1u;
      const float4 A0 = p.tri_a[2u * ti], A1 = p.tri_a[2u * ti + 1u];
      const float bz = p.tri_b[ti];
      bool keep = valid;
      if (rf.usable) keep = valid && !tile_misses_triangle<false, false, SL>(rf, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z});
      const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
      const uint32_t pos = cnt + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
      if (keep) cand[pos] = tri;
      cnt += static_cast<uint32_t>(__builtin_popcountll(m));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // this wave's ds_writes before its ds_reads
    __builtin_amdgcn_wave_barrier();
  }

  // ---- level 2: 8 candidates x the 8 tiles, lane = tile * 8 + candidate slot ---------------------------------------------
  const uint32_t t8 = lane >> 3;
  FocalBounds mine;                                                  // tile t8's box: from lane 4 * t8
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    mine.lo[i] = __shfl(tb.lo[i], static_cast<int>(4u * t8), 64);
    mine.hi[i] = __shfl(tb.hi[i], static_cast<int>(4u * t8), 64);
  }
  const unsigned long long okm = __builtin_amdgcn_ballot_w64(tb.ok), anym = __builtin_amdgcn_ballot_w64(tb.any);
  mine.ok = ((okm >> (4u * t8)) & 1ull) != 0ull;
  mine.any = ((anym >> (4u * t8)) & 1ull) != 0ull;
  const uint32_t brow = ry * 2u + (t8 >> 2);
  region_level2<SL, 8>(p, lane, mine, cand, cnt, cnt, (brow * gx + rx) * 4u + (t8 & 3u), brow < gy, brow >= p.cost_split_brow);
}

// ------------------------------------------------------------------------------------
// Scenes of up to 32 triangles (C3): the same two levels with TWO regions per wave.  Level 1 uses half of a wave's lanes there
// (lane = triangle) and level 2 three of its eight candidate slots per tile (a C3 region keeps ~3 of the 32 triangles):
//   1. lanes 0..31 own region 2 w, lanes 32..63 region 2 w + 1: one pinhole pass for both regions' 64 tile corners;
//   2. level 1, lane = region half * 32 + triangle, per-half compaction into the half's candidate list;
//   3. level 2 with FOUR slots per tile, lane = tile * 4 + slot over the 16 tiles of both regions, when neither region keeps more
//      than four candidates (every candidate has a lane: the pairwise certain-winner bound applies as with eight); else the
//      eight-slot pass of region_lists_kernel for one region after the other.
// Verdict calls, their inputs and the order of the survivors are those of region_lists_kernel: the same lists and header words.
// grid = ceil(regions / 8) blocks of 256 threads.
// ------------------------------------------------------------------------------------
template <bool FMA, class SL = SlackProduct>
// (116 VGPRs, four waves per SIMD: held to 96 for five it spills 84 bytes per lane and the step costs 57.9 instead of 57.3 us)
__global__ __launch_bounds__(256, 4) void region_pair_lists_kernel(const TraceParams p) {
  __shared__ uint32_t s_cand[4][2][32];                             // per wave and half: the region's candidates, ascending
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, half = lane >> 5;
  const uint32_t gx = (p.W + 31u) / 32u, gy = (p.rows + 7u) / 8u, gry = (gy + 1u) / 2u, regions = gx * gry;
  const uint32_t first = (blockIdx.x * 4u + wave) * 2u;
  if (first >= regions) return;                                     // wave-uniform
  const bool live = first + half < regions;                         // (the last wave may own one region only)
  const uint32_t region = live ? first + half : first;              // a half without a region shadows the other one and writes nothing
  const uint32_t rx = region % gx, ry = region / gx;
  FocalBounds tb, rb;
  region_focal_bounds<FMA, SL, true>(p, rx, ry, lane, tb, rb);
  const uint32_t n = p.n_tris;                                       // <= 32 (host)
  uint32_t* const cand = s_cand[wave][half];

  // ---- level 1, lane = half * 32 + triangle
  uint32_t cnt;
  {
    const TileFamily rf = make_family<SL>(p, rb);
    const uint32_t tri = lane & 31u;
    const bool valid = tri < n;
    const uint32_t ti = valid ? tri : n - 1u;
    const float4 A0 = p.tri_a[2u * ti], A1 = p.tri_a[2u * ti + 1u];
    const float bz = p.tri_b[ti];
    bool keep = valid;
    if (rf.usable) keep = valid && !tile_misses_triangle<false, false, SL>(rf, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z});
    const uint32_t hm = static_cast<uint32_t>(__builtin_amdgcn_ballot_w64(keep) >> (lane & 32u));   // this half's survivors
    if (keep) cand[__builtin_popcount(hm & ((1u << tri) - 1u))] = tri;
    cnt = static_cast<uint32_t>(__builtin_popcount(hm));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // this wave's ds_writes before its ds_reads
    __builtin_amdgcn_wave_barrier();
  }
  const uint32_t cnt_a = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cnt), 0));
  const uint32_t cnt_b = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cnt), 32));
  const unsigned long long okm = __builtin_amdgcn_ballot_w64(tb.ok), anym = __builtin_amdgcn_ballot_w64(tb.any && live);

  if (cnt_a <= 4u && cnt_b <= 4u) {
    // ---- level 2, four slots per tile: lane = tile * 4 + slot, tiles 0..7 of region A, 8..15 of region B (lanes 4 t .. 4 t + 3
    //      hold tile t's box already)
    const uint32_t t8 = (lane >> 2) & 7u;
    FocalBounds mine = tb;
    mine.ok = ((okm >> (lane & 60u)) & 1ull) != 0ull;
    mine.any = ((anym >> (lane & 60u)) & 1ull) != 0ull;
    const uint32_t brow = ry * 2u + (t8 >> 2);
    region_level2<SL, 4>(p, lane, mine, cand, cnt, cnt_a > cnt_b ? cnt_a : cnt_b, (brow * gx + rx) * 4u + (t8 & 3u), live && brow < gy,
                         brow >= p.cost_split_brow);
  } else {
    // ---- level 2 as region_lists_kernel runs it, one region after the other
#pragma nounroll
    for (uint32_t h = 0; h < 2u; ++h) {
      if (first + h >= regions) break;                               // wave-uniform
      const uint32_t t8 = lane >> 3, src = h * 32u + 4u * t8;
      FocalBounds mine;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        mine.lo[i] = __shfl(tb.lo[i], static_cast<int>(src), 64);
        mine.hi[i] = __shfl(tb.hi[i], static_cast<int>(src), 64);
      }
      mine.ok = ((okm >> src) & 1ull) != 0ull;
      mine.any = ((anym >> src) & 1ull) != 0ull;
      const uint32_t hrx = (first + h) % gx, hry = (first + h) / gx, brow = hry * 2u + (t8 >> 2);
      const uint32_t hc = h == 0u ? cnt_a : cnt_b;
      region_level2<SL, 8>(p, lane, mine, s_cand[wave][h], hc, hc, (brow * gx + hrx) * 4u + (t8 & 3u), brow < gy, brow >= p.cost_split_brow);
    }
  }
}


}  // namespace rtk
