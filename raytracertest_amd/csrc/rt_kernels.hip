// rt_kernels.hip -- hand-written gfx950 kernels of the trace path and their launchers.
//
//   (rng_init_kernel   <- random::InitRandomStates: rt_rng_init.hip)
//   prep_triangles     <- per-triangle invariants hoisted out of rt::Radiance
//   trace_kernel       <- rt::TraceKernel + Radiance + HitTriangle + ThinLensCamera::GetRay
//                         (rt_trace.hpp; RayTracer/Kernels.cuh:29-147, ThinLensCamera.cuh:30-52,111-130)
//   convert_kernel     <- rt::ConverterKernel              (RayTracer/Kernels.cuh:149-169)
//   dbg_* kernels      <- single-function harnesses used by the parity tests
#include <stdlib.h>

#include "rt_dense.hpp"

namespace rtk {

// ------------------------------------------------------------------------------------
// Per-triangle invariants.  e1 = v1 - v0 and e2 = v2 - v0 are the same fp32 subtractions
// HitTriangle performs per ray (Kernels.cuh:37-38); colour = abs(normalize(cross(e1,e2)))
// is the shade of a hit (Kernels.cuh:97-99), a function of the triangle only.
// Record layout (36 bytes per triangle, what the trace kernel keeps in LDS):
//   tri_a[2i]   = (e2.x, e2.y, e2.z, e1.x)      stage A reads tri_a[2i], tri_a[2i+1]
//   tri_a[2i+1] = (e1.y, e1.z, v0.x, v0.y)      (two ds_read_b128, wave-uniform)
//   tri_b[i]    = v0.z                          stage B adds one ds_read_b32
// ------------------------------------------------------------------------------------
// `edges`: the input is already in the v0, e0 = v1-v0, e1 = v2-v0 layout of
// Documentation/gpu.meshes.txt:16-17 (no subtraction here; .w = packed vertex normals, kept
// by the caller, unused by the reference's flat shading).
template <bool FMA>
__global__ __launch_bounds__(256) void prep_triangles_kernel(const float4* __restrict__ verts, uint32_t n, bool edges,
                                                              float4* __restrict__ tri_a,
                                                              float* __restrict__ tri_b,
                                                              float4* __restrict__ color,
                                                              float4* __restrict__ normals) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const float4 a = verts[3 * i + 0], b = verts[3 * i + 1], c = verts[3 * i + 2];
  const V3 v0 = {a.x, a.y, a.z};
  const V3 e1 = edges ? V3{b.x, b.y, b.z} : rtd::sub({b.x, b.y, b.z}, v0);
  const V3 e2 = edges ? V3{c.x, c.y, c.z} : rtd::sub({c.x, c.y, c.z}, v0);
  tri_a[2 * i + 0] = make_float4(e2.x, e2.y, e2.z, e1.x);
  tri_a[2 * i + 1] = make_float4(e1.y, e1.z, v0.x, v0.y);
  tri_b[i] = v0.z;
  const V3 nn = Math<FMA>::normalize(Math<FMA>::cross(e1, e2));
  color[i] = make_float4(rtd::absf(nn.x), rtd::absf(nn.y), rtd::absf(nn.z), 0.0f);
  if (normals != nullptr) {            // edge-format rows carry a packed vertex normal in .w
    const V3 n0 = rtd::unpack_normal(a.w), n1 = rtd::unpack_normal(b.w), n2 = rtd::unpack_normal(c.w);
    normals[3 * i + 0] = make_float4(n0.x, n0.y, n0.z, 0.0f);
    normals[3 * i + 1] = make_float4(n1.x, n1.y, n1.z, 0.0f);
    normals[3 * i + 2] = make_float4(n2.x, n2.y, n2.z, 0.0f);
  }
}

// ------------------------------------------------------------------------------------
// rt::ConverterKernel, Kernels.cuh:149-169: BGRA8 = pack(255 * sum / count)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void convert_kernel(const float4* __restrict__ render,
                                                       const uint32_t* __restrict__ counts,
                                                       uint32_t* __restrict__ image, uint32_t npix) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= npix) return;
  const float4 s = render[i];
  const float cnt = static_cast<float>(counts[i]);                  // :164
  image[i] = rtd::pack_color(255.0f * (s.x / cnt), 255.0f * (s.y / cnt), 255.0f * (s.z / cnt));
}

// ------------------------------------------------------------------------------------
// Debug harness kernels (parity tests of single functions on the device)
// ------------------------------------------------------------------------------------
template <bool FMA>
__global__ void dbg_hit_triangle_kernel(uint32_t n, const float* __restrict__ rays,
                                        const float* __restrict__ tris, int eps_mode,
                                        int* __restrict__ hit, float* __restrict__ tuv,
                                        float* __restrict__ normal, float* __restrict__ point) {
  using M = Math<FMA>;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = rays + 6 * static_cast<size_t>(i);
  const float* q = tris + 9 * static_cast<size_t>(i);
  const V3 o = {r[0], r[1], r[2]};
  const V3 d = M::normalize({r[3], r[4], r[5]});                    // rt::Ray( o, d, true ), Ray.cuh:12-17
  const V3 a = {q[0], q[1], q[2]}, b = {q[3], q[4], q[5]}, c = {q[6], q[7], q[8]};
  const V3 e1 = rtd::sub(b, a), e2 = rtd::sub(c, a);
  float t = 0.0f, u = 0.0f, v = 0.0f;
  int stage;
  const bool h = hit_triangle_exact<FMA>(o, d, a, e1, e2, eps_mode ? FLT_EPSILON : 0.0000000001f, t, u, v, stage);
  hit[i] = h ? 1 : 0;
  tuv[3 * i + 0] = t; tuv[3 * i + 1] = u; tuv[3 * i + 2] = v;
  const V3 nn = M::normalize(M::cross(e1, e2));
  normal[3 * i + 0] = nn.x; normal[3 * i + 1] = nn.y; normal[3 * i + 2] = nn.z;
  point[3 * i + 0] = M::madd1(d.x, t, o.x);                         // Ray::point
  point[3 * i + 1] = M::madd1(d.y, t, o.y);
  point[3 * i + 2] = M::madd1(d.z, t, o.z);
}

__global__ void dbg_sincos_kernel(uint32_t n, const float* __restrict__ x, float* __restrict__ s,
                                  float* __restrict__ c) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float sn, cs;
  rtd::sincos_spec(x[i], sn, cs);
  s[i] = sn; c[i] = cs;
}

// n states (array of {d,v0..v4}); m uniforms each -> out[n][m]; states advanced in place
__global__ void dbg_uniform_kernel(uint32_t n, uint32_t m, uint32_t* __restrict__ states,
                                   float* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t* s = states + 6 * static_cast<size_t>(i);
  Rng r = {s[0], s[1], s[2], s[3], s[4], s[5]};
  for (uint32_t j = 0; j < m; ++j) out[static_cast<size_t>(i) * m + j] = rtd::rng_uniform(r);
  s[0] = r.d; s[1] = r.v0; s[2] = r.v1; s[3] = r.v2; s[4] = r.v3; s[5] = r.v4;
}

// thin-lens rays for n (px, py) pairs with their RNG states -> rays[n][6]
template <bool FMA>
__global__ void dbg_get_ray_kernel(const TraceParams p, uint32_t n, const uint32_t* __restrict__ pixels,
                                   uint32_t* __restrict__ states, float* __restrict__ rays) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t* s = states + 6 * static_cast<size_t>(i);
  Rng r = {s[0], s[1], s[2], s[3], s[4], s[5]};
  V3 po, pd, o, d;
  pinhole<FMA>(p, pixels[2 * i], pixels[2 * i + 1], po, pd);
  get_ray<FMA>(p, focal_point<FMA>(p, pd), r, o, d);
  float* out = rays + 6 * static_cast<size_t>(i);
  out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = d.x; out[4] = d.y; out[5] = d.z;
  s[0] = r.d; s[1] = r.v0; s[2] = r.v1; s[3] = r.v2; s[4] = r.v3; s[5] = r.v4;
}

// The focal box every wave tile of a trace launch would classify with (focal_bounds: four corner pixels + curvature for
// full tiles, all in-image lanes otherwise) next to the focal points of its pixels, for the test that the box holds them
// all.  Same grid as the trace kernel.  boxes[tile][8] = lo[3], hi[3], corner path taken, usable; focal[pixel of the band][3].
template <bool FMA>
__global__ __launch_bounds__(256) void dbg_focal_boxes_kernel(const TraceParams p, float* __restrict__ boxes, float* __restrict__ focal_out) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t px = blockIdx.x * 32u + wave * 8u + (lane & 7u);
  const uint32_t ly = blockIdx.y * 8u + (lane >> 3);
  const bool inside = px < p.W && ly < p.rows;
  const uint32_t cxp = inside ? px : 0u, cyp = inside ? ly : 0u;
  V3 po, pd;
  pinhole<FMA>(p, cxp, p.row0 + cyp, po, pd);
  const V3 f = focal_point<FMA>(p, pd);
  const FocalBounds b = focal_bounds(p, f, inside);
  const bool corner_path = p.tile_curv > 0.0f && __builtin_amdgcn_ballot_w64(inside) == ~0ull;
  if (inside) {
    float* o = focal_out + 3u * (static_cast<size_t>(ly) * p.W + px);
    o[0] = f.x; o[1] = f.y; o[2] = f.z;
  }
  if (lane == 0u) {
    float* o = boxes + 8u * ((static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 4u + wave);
#pragma unroll
    for (int i = 0; i < 3; ++i) { o[i] = b.lo[i]; o[3 + i] = b.hi[i]; }
    o[6] = corner_path ? 1.0f : 0.0f;
    o[7] = (b.ok && b.any) ? 1.0f : 0.0f;
  }
}

// ------------------------------------------------------------------------------------
// The conservative classification, verdict by verdict (harness of tests/test_gpu_classification.py).
// One block per REGION of the band -- level 0: the 8x8 wave tile at (x0, y0), bounded exactly as a trace wave bounds
// it (focal_bounds: corner path for full tiles); level 1: the 32x8 block at (x0, y0), the union of its four wave tiles
// (block_focal_union, the trace kernel's block-level pre-cull); level 2: the macro tile of p.macro_w x p.macro_h pixels
// at (x0, y0) (rect_focal_bounds, macro_bin_kernel); level 4: the super tile of super_f x super_f macro tiles (super_bin_kernel) -- and for EVERY triangle of the scene what
// tile_misses_triangle decides and the interval ends it decides from, with every rounding allowance scaled by SL::scale.
//   out[region] = 16 header floats: focal lo[3], hi[3], lmin, lmax, usable (+ 2: the focal-bound paths agree, + 4: p.tile_curv > 0, the two-level list builder is in use), A, orad[3], fc[3]
//               + n_tris x stride floats: flags (1 keep | 2 certainly hit), det_lo, det_hi, U_lo, U_hi, V_lo, V_hi,
//                 q_lo, q_hi, Nt_lo, Nt_hi, 0 (stride 12; the small-scene instantiation <false, SURE>), or, FORMS:
//                 flags (1 keep), the same six ends, 5 x 0, then the 18 numbers of the per-sample forms -- each form scaled by
//                 its power of two, the gradients [9..17] the fp16 values the trace kernel stores (stride 32; the large-scene
//                 instantiation).
// ------------------------------------------------------------------------------------
template <bool FMA, bool FORMS, class SL>
__global__ __launch_bounds__(256) void dbg_classify_kernel(const TraceParams p, uint32_t level, const uint32_t* __restrict__ regions,
                                                            float* __restrict__ out) {
  __shared__ float s_box[4][8];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t x0 = regions[2u * blockIdx.x], y0 = regions[2u * blockIdx.x + 1u];
  constexpr uint32_t stride = FORMS ? 32u : 12u;
  float* const o = out + static_cast<size_t>(blockIdx.x) * (16u + static_cast<size_t>(p.n_tris) * stride);
  FocalBounds bb;
  bool paths_agree = true;           // level 0: focal_bounds (trace waves of large scenes) == group_focal_bounds (list builder)
  if (level == 2u) {
    bb = rect_focal_bounds<FMA, 4>(p, x0, y0, p.macro_w, p.macro_h, s_box);
  } else if (level == 4u) {                          // the super tile of super_bin_kernel: every pixel of super_f x super_f macro tiles
    bb = rect_focal_bounds<FMA, 4>(p, x0, y0, p.macro_w * p.super_f, p.macro_h * p.super_f, s_box);
  } else if (level == 3u) {                          // the 32x16 region of the two-level list builder: the union of its tiles' boxes
    FocalBounds tb;
    region_focal_bounds<FMA, SL>(p, x0 / 32u, y0 / 16u, lane, tb, bb);
  } else {
    const uint32_t px = x0 + wave * 8u + (lane & 7u), ly = y0 + (lane >> 3);
    const bool inside = px < p.W && ly < p.rows;
    const uint32_t cxp = inside ? px : 0u, cyp = inside ? ly : 0u;
    V3 po, pd;
    pinhole<FMA>(p, cxp, p.row0 + cyp, po, pd);
    const FocalBounds wb = focal_bounds<SL>(p, focal_point<FMA>(p, pd), inside);
    const FocalBounds ub = block_focal_union(wb, &s_box[0][0], wave, lane);
    if (level == 1u) {
      bb = ub;
    } else {
      // the wave tile at (x0, y0): the bounds tile_lists_kernel builds the tile's list from (small scenes) -- identical to
      // what the trace wave of a large scene computes for its tile (focal_bounds, wave 0 above: s_box[0]); both are exported
      // paths of the same arithmetic, the harness takes the list builder's
      const bool full = x0 + 8u <= p.W && y0 + 8u <= p.rows;
      if (p.tile_curv > 0.0f) {                      // what region_lists_kernel gives this tile (clamped corners for edge tiles)
        FocalBounds tb, rb;
        region_focal_bounds<FMA, SL>(p, x0 / 32u, y0 / 16u, lane, tb, rb);
        const int src = static_cast<int>(4u * ((((y0 / 8u) & 1u) << 2) | ((x0 / 8u) & 3u)));
#pragma unroll
        for (int i = 0; i < 3; ++i) { bb.lo[i] = __shfl(tb.lo[i], src, 64); bb.hi[i] = __shfl(tb.hi[i], src, 64); }
        bb.ok = ((__builtin_amdgcn_ballot_w64(tb.ok) >> src) & 1ull) != 0ull;
        bb.any = ((__builtin_amdgcn_ballot_w64(tb.any) >> src) & 1ull) != 0ull;
      } else {
        bb = group_focal_bounds<FMA, 64, SL>(p, x0, y0, x0 < p.W, lane, 0u);
      }
      // full tiles: every path is the same four corner pixels (or the same 64): bit-identical boxes
      paths_agree = !full || (s_box[0][0] == bb.lo[0] && s_box[0][1] == bb.lo[1] && s_box[0][2] == bb.lo[2] && s_box[0][3] == bb.hi[0] &&
                              s_box[0][4] == bb.hi[1] && s_box[0][5] == bb.hi[2] && (s_box[0][6] != 0.0f) == bb.ok && (s_box[0][7] != 0.0f) == bb.any);
    }
  }
  const TileFamily fam = make_family<SL>(p, bb);
  if (threadIdx.x == 0u) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { o[i] = bb.lo[i]; o[3 + i] = bb.hi[i]; o[10 + i] = fam.orad[i]; o[13 + i] = fam.fc[i]; }
    o[6] = fam.lmin; o[7] = fam.lmax; o[9] = fam.A;
    o[8] = (fam.usable ? 1.0f : 0.0f) + (paths_agree ? 2.0f : 0.0f) + (p.tile_curv > 0.0f ? 4.0f : 0.0f);
  }
  for (uint32_t tri = threadIdx.x; tri < p.n_tris; tri += 256u) {
    const float4 A0 = p.tri_a[2u * tri], A1 = p.tri_a[2u * tri + 1u];
    const float bz = p.tri_b[tri];
    float* const r = o + 16u + static_cast<size_t>(tri) * stride;
    float ends[10] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (FORMS) {
      float forms[18];
#pragma unroll
      for (int i = 0; i < 18; ++i) forms[i] = 0.0f;
      bool keep = true;
      if (fam.usable) {     // as the trace wave does it: the S rules as a call of their own, then the forms call (whose ends are exported)
        float ends3[10] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        const bool miss3 = tile_misses_triangle<false, false, SL, true>(fam, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z}, nullptr, nullptr, nullptr, ends3);
        keep = !tile_misses_triangle<true, false, SL>(fam, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z}, forms, nullptr, nullptr, ends) && !miss3;
        keep = keep && lens_can_pass_forms(forms, fam.frad, fam.A);      // (wave_lists_kernel's last drop rule: the forms taken jointly)
        ends[8] = ends3[8]; ends[9] = ends3[9];
      }
      r[0] = keep ? 1.0f : 0.0f;
#pragma unroll
      for (int i = 0; i < 6; ++i) r[1 + i] = ends[i];
#pragma unroll
      for (int i = 9; i < 12; ++i) r[i] = 0.0f;
      r[7] = ends[8]; r[8] = ends[9];                                // S_lo, S_hi
#pragma unroll
      for (int i = 0; i < 9; ++i) r[12 + i] = forms[i];
#pragma unroll
      for (int i = 9; i < 18; ++i) r[12 + i] = forms[i];           // fp16-exact values, as the trace kernel stores them
      r[30] = 0.0f; r[31] = 0.0f;
    } else {
      bool keep = true, sure = false;
      float q[2] = {0.0f, 0.0f};
      if (fam.usable) keep = !tile_misses_triangle<false, true, SL>(fam, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z}, nullptr, &sure, q, ends);
      r[0] = (keep ? 1.0f : 0.0f) + ((fam.usable && sure) ? 2.0f : 0.0f);
#pragma unroll
      for (int i = 0; i < 6; ++i) r[1 + i] = ends[i];
      r[7] = q[0]; r[8] = q[1]; r[9] = ends[8]; r[10] = ends[9]; r[11] = 0.0f;   // q bounds; S_lo, S_hi
    }
  }
}

// fp32 VALU calibration: 8 independent fma chains per lane, 16x unrolled.  Measures the
// attainable lane-FMA rate of THIS device under load (the honest denominator of the trace
// kernel's VALU roofline) and the clock it holds (s_memtime ticks / 100 MHz realtime).
__global__ __launch_bounds__(256) void dbg_valu_peak_kernel(float* __restrict__ out, int iters,
                                                             unsigned long long* __restrict__ clk) {
  float a[8];
  const float x = 1.0000001f + threadIdx.x * 1e-9f, y = 1e-7f;
#pragma unroll
  for (int c = 0; c < 8; ++c) a[c] = static_cast<float>(c + threadIdx.x);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i += 16) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < 8; ++c) a[c] = __builtin_fmaf(a[c], x, y);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += a[c];
  out[blockIdx.x * 256u + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// ------------------------------------------------------------------------------------
// Host-side launchers (the only symbols other translation units see)
// ------------------------------------------------------------------------------------
static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

hipError_t launch_prep_triangles(bool fma, bool edges, const float4* verts, uint32_t n, float4* tri_a, float* tri_b,
                                 float4* color, float4* normals, hipStream_t st) {
  if (n == 0) return hipSuccess;
  if (fma) hipLaunchKernelGGL(prep_triangles_kernel<true>, dim3(cdiv(n, 256)), dim3(256), 0, st, verts, n, edges, tri_a, tri_b, color, normals);
  else hipLaunchKernelGGL(prep_triangles_kernel<false>, dim3(cdiv(n, 256)), dim3(256), 0, st, verts, n, edges, tri_a, tri_b, color, normals);
  return hipGetLastError();
}

// small-scene kernels (BIN && ONEPASS): a block of four tiles is 4 / RT_SMALL_WG_WAVES workgroups (trace_kernel)
constexpr uint32_t kSmallWgWaves = RT_SMALL_WG_WAVES;
static_assert(kSmallWgWaves == 1u || kSmallWgWaves == 2u || kSmallWgWaves == 4u, "waves per small-scene workgroup");

uint32_t trace_lds_bytes(const TraceParams& p, bool bin) {
  if (bin) {
    const bool large = p.n_tris > p.bin_list;
    const uint32_t per_candidate = (large && (p.flags & TRACE_PRETEST)) ? 104u : 40u;
    return (large ? 4u : kSmallWgWaves) * p.bin_list * per_candidate + (large ? p.block_list * 4u + 160u : 0u);
  }
  const uint32_t staged = p.n_tris < p.chunk ? p.n_tris : p.chunk;
  return staged * 36u;
}

template <bool FMA, bool FILTER, bool STATS, bool BIN, bool ONEPASS>
static void launch_trace_o(const TraceParams& p, int K, dim3 grid, size_t lds, hipStream_t st) {
  dim3 blk(256);
  if (BIN && ONEPASS) { grid.x *= 4u / kSmallWgWaves; blk.x = 64u * kSmallWgWaves; }
  switch (K) {
    case 1: hipLaunchKernelGGL((trace_kernel<FMA, 1, FILTER, STATS, BIN, ONEPASS>), grid, blk, lds, st, p); break;
    case 2: hipLaunchKernelGGL((trace_kernel<FMA, 2, FILTER, STATS, BIN, ONEPASS>), grid, blk, lds, st, p); break;
    default: hipLaunchKernelGGL((trace_kernel<FMA, 4, FILTER, STATS, BIN, ONEPASS>), grid, blk, lds, st, p); break;
  }
}

template <bool FMA, bool FILTER, bool STATS, bool BIN>
static void launch_trace_k(const TraceParams& p, int K, dim3 grid, size_t lds, hipStream_t st) {
  // one classification pass suffices when the whole scene fits the per-wave candidate list
  if (BIN && p.n_tris <= p.bin_list) launch_trace_o<FMA, FILTER, STATS, BIN, BIN>(p, K, grid, lds, st);
  else launch_trace_o<FMA, FILTER, STATS, BIN, false>(p, K, grid, lds, st);
}

template <bool FMA, bool FILTER, bool STATS>
static void launch_trace_b(const TraceParams& p, bool bin, int K, dim3 grid, size_t lds, hipStream_t st) {
  if (bin) launch_trace_k<FMA, FILTER, STATS, true>(p, K, grid, lds, st);
  else launch_trace_k<FMA, FILTER, STATS, false>(p, K, grid, lds, st);
}

template <bool FMA, bool STATS>
static void launch_trace_f(const TraceParams& p, bool filter, bool bin, int K, dim3 grid, size_t lds, hipStream_t st) {
  if (filter) launch_trace_b<FMA, true, STATS>(p, bin, K, grid, lds, st);
  else launch_trace_b<FMA, false, STATS>(p, bin, K, grid, lds, st);
}

// Large-scene kernels with the per-sample forms (TRACE_PRETEST): filtered + classified only; STATS and FUSE variants
template <bool FMA, bool STATS, bool FUSE>
static void launch_trace_pre(const TraceParams& p, int K, dim3 grid, size_t lds, hipStream_t st) {
#define RT_PRE(KK) hipLaunchKernelGGL((trace_kernel<FMA, KK, true, STATS, true, false, FUSE, true>), grid, dim3(256), lds, st, p)
  if (K == 1) RT_PRE(1); else if (K == 2) RT_PRE(2); else RT_PRE(4);
#undef RT_PRE
}

template <bool FMA>
static void launch_trace_fused(const TraceParams& p, int K, dim3 grid, size_t lds, hipStream_t st) {
  const bool onepass = p.n_tris <= p.bin_list;
  if (!onepass && (p.flags & TRACE_PRETEST)) { launch_trace_pre<FMA, false, true>(p, K, grid, lds, st); return; }
#define RT_FUSED(KK, OP) hipLaunchKernelGGL((trace_kernel<FMA, KK, true, false, true, OP, true>), grid, blk, lds, st, p)
  dim3 blk(256);
  if (onepass) { grid.x *= 4u / kSmallWgWaves; blk.x = 64u * kSmallWgWaves; }
  if (onepass) { if (K == 1) RT_FUSED(1, true); else if (K == 2) RT_FUSED(2, true); else RT_FUSED(4, true); }
  else { if (K == 1) RT_FUSED(1, false); else if (K == 2) RT_FUSED(2, false); else RT_FUSED(4, false); }
#undef RT_FUSED
}

bool trace_can_fuse(bool filter, bool bin) { return filter && bin; }

// dense scenes with lists in HBM (rt_dense.hpp): default, un-instrumented launches only
template <bool FMA, bool FUSE>
static void launch_dense(const TraceParams& p, int K, dim3 grid, size_t lds, hipStream_t st) {
#define RT_DENSE(KK) hipLaunchKernelGGL((trace_kernel<FMA, KK, true, false, true, false, FUSE, true, true>), grid, dim3(256), lds, st, p)
  if (K == 1) RT_DENSE(1); else if (K == 2) RT_DENSE(2); else RT_DENSE(4);
#undef RT_DENSE
}

hipError_t launch_wave_lists(const TraceParams& p, bool fma, hipStream_t st) {
  if (p.wave_lists == nullptr || p.rows == 0u || p.W == 0u) return hipSuccess;
  const dim3 grid(cdiv(cdiv(p.W, 32) * cdiv(p.rows, 8), 4));           // one wave per block of 32 x 8 pixels
  const size_t lds = static_cast<size_t>(p.block_list) * 4u * 4u;
  if (fma) hipLaunchKernelGGL(wave_lists_kernel<true>, grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL(wave_lists_kernel<false>, grid, dim3(256), lds, st, p);
  return hipGetLastError();
}

hipError_t launch_trace(const TraceParams& p, bool fma, bool filter, bool bin, int K, hipStream_t st) {
  // samples == 0 is a real launch, as in the reference (TraceKernel with sampleCount 0: counts += 0,
  // render += 0, RNG written back, Kernels.cuh:133-146): the fused clear / BGRA8 emit / list store of the
  // launch still have to happen
  if (p.rows == 0 || p.W == 0) return hipSuccess;
  dim3 grid(cdiv(p.W, 32), cdiv(p.rows, 8));
  if (p.row_il != 0u) {                                  // every second block row (small-scene kernels only)
    if (!(bin && p.n_tris <= p.bin_list) || p.stats != nullptr) return hipErrorInvalidValue;
    const uint32_t R = grid.y, G = p.row_il, full = R / (2u * G), rest = R % (2u * G);      // groups of G block rows, alternating
    grid.y = full * G + (p.row_phase == 0u ? (rest < G ? rest : G) : (rest > G ? rest - G : 0u));
    if (grid.y == 0u) return hipSuccess;
  }
  if (p.wave_lists != nullptr) {     // dense scenes, lists in HBM: no LDS, no classification in the trace kernel
    if (!((p.flags & TRACE_PRETEST) && filter && bin && p.n_tris > p.bin_list) || p.stats != nullptr) return hipErrorInvalidValue;
    const size_t lds_d = 4u * p.bin_list * 116u;                       // four waves' records + forms + colours; no block list
    if (p.iters > 1u) { if (fma) launch_dense<true, true>(p, K, grid, lds_d, st); else launch_dense<false, true>(p, K, grid, lds_d, st); }
    else { if (fma) launch_dense<true, false>(p, K, grid, lds_d, st); else launch_dense<false, false>(p, K, grid, lds_d, st); }
    return hipGetLastError();
  }
  const size_t lds = trace_lds_bytes(p, bin);
  if (p.iters > 1u) {                // fused iterations: default (filtered, classified, un-instrumented) kernels only
    if (!trace_can_fuse(filter, bin) || p.stats != nullptr) return hipErrorInvalidValue;
    if (fma) launch_trace_fused<true>(p, K, grid, lds, st);
    else launch_trace_fused<false>(p, K, grid, lds, st);
    return hipGetLastError();
  }
  if ((p.flags & TRACE_PRETEST) && filter && bin && p.n_tris > p.bin_list) {   // large-scene kernels with the per-sample forms
    if (p.stats != nullptr) { if (fma) launch_trace_pre<true, true, false>(p, K, grid, lds, st); else launch_trace_pre<false, true, false>(p, K, grid, lds, st); }
    else { if (fma) launch_trace_pre<true, false, false>(p, K, grid, lds, st); else launch_trace_pre<false, false, false>(p, K, grid, lds, st); }
    return hipGetLastError();
  }
  if (p.stats != nullptr) {          // instrumented build of the same kernel (not the timed path)
    if (fma) launch_trace_f<true, true>(p, filter, bin, K, grid, lds, st);
    else launch_trace_f<false, true>(p, filter, bin, K, grid, lds, st);
  } else {
    if (fma) launch_trace_f<true, false>(p, filter, bin, K, grid, lds, st);
    else launch_trace_f<false, false>(p, filter, bin, K, grid, lds, st);
  }
  return hipGetLastError();
}

// blocks of the default (fma, filtered, binned) trace kernel the occupancy API admits per CU
int trace_occupancy(int K, size_t lds) {
  int n = 0;
  hipError_t e;
  if (K == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<true, 1, true, false, true, true>, 64 * kSmallWgWaves, lds);
  else if (K == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<true, 2, true, false, true, true>, 64 * kSmallWgWaves, lds);
  else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<true, 4, true, false, true, false>, 256, lds);
  return e == hipSuccess ? n : -1;
}

hipError_t launch_convert(const float4* render, const uint32_t* counts, uint32_t* image, uint32_t npix,
                          hipStream_t st) {
  if (npix == 0) return hipSuccess;
  hipLaunchKernelGGL(convert_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, st, render, counts, image, npix);
  return hipGetLastError();
}

hipError_t launch_dbg_hit_triangle(bool fma, uint32_t n, const float* rays, const float* tris, int eps_mode,
                                   int* hit, float* tuv, float* normal, float* point, hipStream_t st) {
  if (n == 0) return hipSuccess;
  if (fma) hipLaunchKernelGGL(dbg_hit_triangle_kernel<true>, dim3(cdiv(n, 64)), dim3(64), 0, st, n, rays, tris, eps_mode, hit, tuv, normal, point);
  else hipLaunchKernelGGL(dbg_hit_triangle_kernel<false>, dim3(cdiv(n, 64)), dim3(64), 0, st, n, rays, tris, eps_mode, hit, tuv, normal, point);
  return hipGetLastError();
}

hipError_t launch_super_bin(const TraceParams& p, bool fma, hipStream_t st) {
  if (p.super_lists == nullptr || p.macro_bounds == nullptr || p.macro_lists == nullptr || p.n_tris == 0u || p.rows == 0u || p.W == 0u || p.super_f == 0u ||
      p.super_chunks != cdiv(p.n_tris, kSuperChunk) || p.super_chunks > kSuperMaxChunks) return hipErrorInvalidValue;
  const dim3 mgrid(cdiv(p.W, p.macro_w), cdiv(p.rows, p.macro_h));
  const dim3 sgrid(p.super_nx * cdiv(mgrid.y, p.super_f), p.super_chunks);
  if (fma) {
    hipLaunchKernelGGL(macro_bounds_kernel<true>, mgrid, dim3(256), 0, st, p);
    hipLaunchKernelGGL(super_bin_kernel<true>, sgrid, dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL(macro_bounds_kernel<false>, mgrid, dim3(256), 0, st, p);
    hipLaunchKernelGGL(super_bin_kernel<false>, sgrid, dim3(256), 0, st, p);
  }
  return hipGetLastError();
}

hipError_t launch_macro_bin(const TraceParams& p, bool fma, hipStream_t st) {
  if (p.macro_lists == nullptr || p.n_tris == 0u || p.rows == 0u || p.W == 0u) return hipSuccess;
  const dim3 grid(cdiv(p.W, p.macro_w), cdiv(p.rows, p.macro_h));
  if (fma) hipLaunchKernelGGL(macro_bin_kernel<true>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(macro_bin_kernel<false>, grid, dim3(256), 0, st, p);
  return hipGetLastError();
}

// What `samples` samples of a certain-winner tile add up to, per triangle: the additions of Kernels.cuh:137 in sample order
// (accu starts at 0, :133), the accumulation into a cleared RenderBuffer (:141-143) and the conversion of :164-168 -- the
// same operations the trace kernel would run per pixel, run once per triangle.
__global__ __launch_bounds__(256) void sure_table_kernel(const float4* __restrict__ colors, uint32_t n_tris, uint32_t samples,
                                                          float4* __restrict__ out) {
  const uint32_t tri = blockIdx.x * 256u + threadIdx.x;
  if (tri >= n_tris) return;
  const float4 col = colors[tri];
  float ax = 0.0f, ay = 0.0f, az = 0.0f;
  for (uint32_t s = 0; s < samples; ++s) { ax += col.x; ay += col.y; az += col.z; }
  float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  acc.x += ax; acc.y += ay; acc.z += az;
  const float c = static_cast<float>(samples);
  const uint32_t bgra = rtd::pack_color(255.0f * (acc.x / c), 255.0f * (acc.y / c), 255.0f * (acc.z / c));
  out[tri] = make_float4(ax, ay, az, __builtin_bit_cast(float, bgra));
}

// A stagger without a dependency: the lower half of the first split launch after idle has to start about half a kernel
// behind the upper half (HISTORY.md, phase regimes).  Waiting for the upper half's END costs the whole kernel with half the
// device idle; this wave just lets the time pass (s_sleep between reads of the 100 MHz counter; bounded by the trip count).
__global__ __launch_bounds__(64) void delay_kernel(uint32_t ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (uint32_t i = 0; i < 100000u; ++i) {
    if (__builtin_amdgcn_s_memrealtime() - t0 >= ticks) break;
    __builtin_amdgcn_s_sleep(8);
  }
}

hipError_t launch_delay(uint32_t us, hipStream_t st) {
  if (us == 0u) return hipSuccess;
  hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, st, (us > 5000u ? 5000u : us) * 100u);
  return hipGetLastError();
}

__global__ void publish_half_cost_kernel(uint32_t* __restrict__ half_cost, unsigned long long* __restrict__ host_word) {
  const unsigned long long u = half_cost[0], l = half_cost[1];
  half_cost[0] = 0u; half_cost[1] = 0u;
  *reinterpret_cast<volatile unsigned long long*>(host_word) = u | (l << 32);
}

hipError_t launch_publish_half_cost(uint32_t* half_cost, unsigned long long* host_word, hipStream_t st) {
  hipLaunchKernelGGL(publish_half_cost_kernel, dim3(1), dim3(1), 0, st, half_cost, host_word);
  return hipGetLastError();
}

hipError_t launch_sure_table(const float4* colors, uint32_t n_tris, uint32_t samples, float4* out, hipStream_t st) {
  if (n_tris == 0u) return hipSuccess;
  hipLaunchKernelGGL(sure_table_kernel, dim3(cdiv(n_tris, 256)), dim3(256), 0, st, colors, n_tris, samples, out);
  return hipGetLastError();
}

hipError_t launch_tile_lists(const TraceParams& p, bool fma, hipStream_t st) {
  if (p.tile_lists == nullptr || p.rows == 0u || p.W == 0u) return hipSuccess;
  if (p.tile_curv > 0.0f && p.n_tris <= 32u && p.n_tris != 0u) {   // two levels, two regions per wave
    const dim3 grid(cdiv(cdiv(p.W, 32) * cdiv(cdiv(p.rows, 8), 2), 8u));
    if (fma) hipLaunchKernelGGL((region_pair_lists_kernel<true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((region_pair_lists_kernel<false>), grid, dim3(256), 0, st, p);
    return hipGetLastError();
  }
  if (p.tile_curv > 0.0f && p.n_tris <= 256u) {             // two levels: region -> tiles (its LDS candidate list holds 256)
    const dim3 grid(cdiv(cdiv(p.W, 32) * cdiv(cdiv(p.rows, 8), 2), 4u));
    if (fma) hipLaunchKernelGGL((region_lists_kernel<true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((region_lists_kernel<false>), grid, dim3(256), 0, st, p);
    return hipGetLastError();
  }
  const uint32_t slots = cdiv(p.W, 32) * cdiv(p.rows, 8) * 4u;
  if (p.n_tris <= 32u) {                                             // two tiles per wave
    const dim3 grid(cdiv(slots, 8u));
    if (fma) hipLaunchKernelGGL((tile_lists_kernel<true, 32>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((tile_lists_kernel<false, 32>), grid, dim3(256), 0, st, p);
  } else {
    const dim3 grid(cdiv(slots, 4u));
    if (fma) hipLaunchKernelGGL((tile_lists_kernel<true, 64>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((tile_lists_kernel<false, 64>), grid, dim3(256), 0, st, p);
  }
  return hipGetLastError();
}

// every float in [2^-96, 2^96]: sqrt_midrange / rcp_midrange against the generic expansions
__global__ __launch_bounds__(256) void dbg_check_midrange_kernel(unsigned long long* __restrict__ out) {
  const uint32_t first = 0x0F800000u, last = 0x6F800000u;
  unsigned long long bad_sqrt = 0, bad_rcp = 0, seen = 0;
  uint32_t first_bad = 0;
  for (unsigned long long b = first + static_cast<unsigned long long>(blockIdx.x) * 256u + threadIdx.x; b <= last;
       b += static_cast<unsigned long long>(gridDim.x) * 256u) {
    const float x = __builtin_bit_cast(float, static_cast<uint32_t>(b));
    const float s0 = __builtin_sqrtf(x), s1 = rtd::sqrt_midrange(x);
    const float r0 = 1.0f / x, r1 = rtd::rcp_midrange(x);
    const bool bs = __builtin_bit_cast(uint32_t, s0) != __builtin_bit_cast(uint32_t, s1);
    const bool br = __builtin_bit_cast(uint32_t, r0) != __builtin_bit_cast(uint32_t, r1);
    bad_sqrt += bs; bad_rcp += br; seen += rtd::midrange(x) ? 1u : 0u;
    if ((bs || br) && first_bad == 0u) first_bad = static_cast<uint32_t>(b);
  }
  atomicAdd(out + 0, seen);
  atomicAdd(out + 1, bad_sqrt);
  atomicAdd(out + 2, bad_rcp);
  if (first_bad != 0u) atomicMax(out + 3, static_cast<unsigned long long>(first_bad));
}

hipError_t launch_dbg_check_midrange(unsigned long long* out, hipStream_t st) {
  hipLaunchKernelGGL(dbg_check_midrange_kernel, dim3(8192), dim3(256), 0, st, out);
  return hipGetLastError();
}

hipError_t launch_dbg_valu_peak(uint32_t blocks, int iters, float* out, unsigned long long* clk, hipStream_t st) {
  hipLaunchKernelGGL(dbg_valu_peak_kernel, dim3(blocks), dim3(256), 0, st, out, iters, clk);
  return hipGetLastError();
}

hipError_t launch_dbg_sincos(uint32_t n, const float* x, float* s, float* c, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(dbg_sincos_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, n, x, s, c);
  return hipGetLastError();
}

hipError_t launch_dbg_uniform(uint32_t n, uint32_t m, uint32_t* states, float* out, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(dbg_uniform_kernel, dim3(cdiv(n, 64)), dim3(64), 0, st, n, m, states, out);
  return hipGetLastError();
}

hipError_t launch_dbg_get_ray(bool fma, const TraceParams& p, uint32_t n, const uint32_t* pixels,
                              uint32_t* states, float* rays, hipStream_t st) {
  if (n == 0) return hipSuccess;
  if (fma) hipLaunchKernelGGL(dbg_get_ray_kernel<true>, dim3(cdiv(n, 64)), dim3(64), 0, st, p, n, pixels, states, rays);
  else hipLaunchKernelGGL(dbg_get_ray_kernel<false>, dim3(cdiv(n, 64)), dim3(64), 0, st, p, n, pixels, states, rays);
  return hipGetLastError();
}

// slack_milli: the scale of every rounding allowance in thousandths -- one of 1000 (the product's), 300, 100, 30, 10, 0
hipError_t launch_dbg_classify(bool fma, bool forms, uint32_t slack_milli, const TraceParams& p, uint32_t level, uint32_t n_regions,
                               const uint32_t* regions, float* out, hipStream_t st) {
  if (n_regions == 0u) return hipSuccess;
#define RT_CLS(F, FO, M) hipLaunchKernelGGL((dbg_classify_kernel<F, FO, SlackMilli<M>>), dim3(n_regions), dim3(256), 0, st, p, level, regions, out)
#define RT_CLS_M(F, FO)                                                                     \
  switch (slack_milli) {                                                                    \
    case 1000u: RT_CLS(F, FO, 1000); break; case 300u: RT_CLS(F, FO, 300); break;          \
    case 100u: RT_CLS(F, FO, 100); break;   case 30u: RT_CLS(F, FO, 30); break;            \
    case 10u: RT_CLS(F, FO, 10); break;     case 0u: RT_CLS(F, FO, 0); break;              \
    default: return hipErrorInvalidValue;                                                   \
  }
  if (fma) { if (forms) { RT_CLS_M(true, true) } else { RT_CLS_M(true, false) } }
  else { if (forms) { RT_CLS_M(false, true) } else { RT_CLS_M(false, false) } }
#undef RT_CLS_M
#undef RT_CLS
  return hipGetLastError();
}

hipError_t launch_dbg_focal_boxes(bool fma, const TraceParams& p, float* boxes, float* focal, hipStream_t st) {
  const dim3 grid(cdiv(p.W, 32), cdiv(p.rows, 8));
  if (fma) hipLaunchKernelGGL(dbg_focal_boxes_kernel<true>, grid, dim3(256), 0, st, p, boxes, focal);
  else hipLaunchKernelGGL(dbg_focal_boxes_kernel<false>, grid, dim3(256), 0, st, p, boxes, focal);
  return hipGetLastError();
}

}  // namespace rtk
