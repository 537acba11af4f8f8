// rt_trace.hpp -- the trace kernel (rt::TraceKernel + Radiance + HitTriangle +
// ThinLensCamera::GetRay; RayTracer/Kernels.cuh:29-147, ThinLensCamera.cuh:30-52,111-130).
// Included by rt_kernels.hip only.
//
// Execution shape (MI355X: 256 CUs x 4 SIMD, wave64, 160 KiB LDS/CU):
//   * one lane = one pixel; a wave covers an 8x8 pixel tile, a 256-thread block 32x8 pixels,
//     so every 128-byte line of the per-pixel buffers is read and written whole by one block;
//   * every lane keeps K samples of its pixel in registers and tests them against one
//     triangle at a time; the 36-byte triangle record is read from LDS with wave-uniform
//     (broadcast) ds_read_b128 x2 (+ ds_read_b32), amortised over 64*K rays;
//   * triangles are always scanned in ascending index order (first-scanned wins ties,
//     Kernels.cuh:84);
//   * __ballot-driven wave-uniform early-outs after the culling test, the u test and the v
//     test (FILTER); the IEEE division only runs for triangles some lane may really hit;
//   * BIN: before tracing, each wave classifies the whole triangle list against the ray
//     family of its tile (lane = triangle, interval bounds) and compacts the survivors with
//     __ballot + mbcnt into a per-wave LDS candidate list, so the exact tests run only on
//     triangles some ray of the tile could hit.  Without BIN the block stages the list into
//     LDS in chunks and every ray scans all of it.
#pragma once
#include <float.h>
#include <type_traits>

#include "rt_device_math.hpp"
#include "rt_kernels.hpp"

namespace rtk {

using rtd::Math;
using rtd::Rng;
using rtd::V3;

#define RT_EPS 0.0000000001f
// Every ROUNDING allowance of the conservative classification goes through these two macros (additive terms
// relative to a magnitude, and factors 1 + x).  RT_BIN_SLACK_SCALE = 1 in the product; the teeth test of the
// adversarial campaign builds the library with the allowances scaled down (tools/stress_boundaries.py must then
// FIND mismatches: profiles/r02_boundary_campaign.txt) -- the thresholds that come from proofs about the
// reference's own tests (-1e-6 det, 1.0002 det) and the fp16 quantisation bound of the forms are not scaled.
#ifndef RT_BIN_SLACK_SCALE
#define RT_BIN_SLACK_SCALE 1.0f
#endif
// The scale is a compile-time policy of the classification functions: the product instantiates them with SlackProduct
// (scale 1: `1.0f * x` folds away), the per-(tile, triangle) harness dbg_classify_kernel with the whole ladder
// 1, 0.3, 0.1, 0.03, 0.01, 0 in ONE library, so that the margin of every allowance is a measured number
// (tests/test_gpu_classification.py, CLASSIFICATION.md).
struct SlackProduct { static constexpr float scale = RT_BIN_SLACK_SCALE; };
template <int MILLI> struct SlackMilli { static constexpr float scale = static_cast<float>(MILLI) / 1000.0f; };
#define RT_SLK(x) (SL::scale * (x))
#define RT_SLKM(x) (1.0f + SL::scale * (x))
#ifndef RT_TRACE_MIN_WAVES
#define RT_TRACE_MIN_WAVES 4     // __launch_bounds__ 2nd argument: waves per SIMD the allocator must allow
                                 // (<= 128 VGPRs; measured C3 213 -> 193 us, C4 27.2 -> 24.3 ms vs the 136-VGPR build)
#endif
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES(K) RT_TRACE_MIN_WAVES
#endif
// candidate records per wave in LDS (40 bytes each): TraceParams::bin_list, a multiple of 64

// ------------------------------------------------------------------------------------
// Exact HitTriangle in the reference's operation order (Kernels.cuh:29-65) on a
// precomputed (v0, e1, e2).  Used by the unfiltered trace path and the dbg harness.
// `stage` reports the exit point: 0 culled at det, 1 rejected at u, 2 rejected at v, 3 hit.
// ------------------------------------------------------------------------------------
template <bool FMA>
__device__ __forceinline__ bool hit_triangle_exact(V3 o, V3 d, V3 v0, V3 e1, V3 e2, float eps,
                                                   float& t, float& u, float& v, int& stage) {
  using M = Math<FMA>;
  stage = 0;
  const V3 pv = M::cross(d, e2);                       // :39
  const float det = M::dot(e1, pv);                    // :40
  if (det < eps) return false;                         // :42
  stage = 1;
  const float inv = 1.0f / det;                        // :47
  const V3 tv = rtd::sub(o, v0);                       // :49
  u = M::dot(tv, pv) * inv;                            // :50
  if (u < 0.0f || u > 1.0f) return false;              // :51
  stage = 2;
  const V3 qv = M::cross(tv, e1);                      // :56
  v = M::dot(d, qv) * inv;                             // :57
  if (v < 0.0f || u + v > 1.0f) return false;          // :58
  stage = 3;
  t = M::dot(e2, qv) * inv;                            // :63
  return true;
}

// Build-defined ray-sphere (Documentation/ray.sphere.png; absent from the reference code)
template <bool FMA>
__device__ __forceinline__ bool hit_sphere(V3 o, V3 d, float4 sph, float& t) {
  using M = Math<FMA>;
  const V3 vv = rtd::sub(o, {sph.x, sph.y, sph.z});
  const float a = M::dot(d, d);
  const float b = 2.0f * M::dot(vv, d);
  const float dvv = M::dot(vv, vv);
  float cc, disc;
  if constexpr (FMA) {
    cc = __builtin_fmaf(-sph.w, sph.w, dvv);
    disc = __builtin_fmaf(b, b, -((4.0f * a) * cc));
  } else {
    cc = dvv - sph.w * sph.w;
    disc = b * b - (4.0f * a) * cc;
  }
  if (disc < 0.0f) return false;
  t = (-b - __builtin_sqrtf(disc)) / (2.0f * a);
  return true;
}

// ThinLensCamera::PinHoleRay, ThinLensCamera.cuh:111-130 (tan(fov/2) and aspect are
// launch constants computed once on the host with the same operations)
template <bool FMA>
__device__ __forceinline__ void pinhole(const TraceParams& p, uint32_t px, uint32_t py, V3& o, V3& d) {
  using M = Math<FMA>;
  const float nx = (static_cast<float>(px) + 0.5f) / static_cast<float>(p.W);     // :116
  const float ny = (static_cast<float>(py) + 0.5f) / static_cast<float>(p.H);     // :117
  const float cx = ((2.0f * nx - 1.0f) * p.half_height) * p.aspect;               // :118
  const float cy = (1.0f - 2.0f * ny) * p.half_height;                            // :119
  o = M::mat_mul_point(p.cam, 0.0f, 0.0f, 0.0f, 1.0f);                            // :124
  const V3 pw = M::mat_mul_point(p.cam, cx, cy, -1.0f, 1.0f);                     // :125
  d = M::normalize(rtd::sub(pw, o));                                              // :127-128
}

// focal point of a pixel, ThinLensCamera.cuh:44: Position() + mFocalLength * primary.direction()
template <bool FMA>
__device__ __forceinline__ V3 focal_point(const TraceParams& p, V3 pd) {
  using M = Math<FMA>;
  return {M::madd1(p.focal, pd.x, p.cam[9]), M::madd1(p.focal, pd.y, p.cam[10]),
          M::madd1(p.focal, pd.z, p.cam[11])};
}

// ThinLensCamera::GetRay, ThinLensCamera.cuh:30-52; `focal` is the pixel's focal point
// (sample-invariant, hoisted)
template <bool FMA>
__device__ __forceinline__ void get_ray(const TraceParams& p, V3 focal, Rng& rng, V3& o, V3& d) {
  using M = Math<FMA>;
  float dx, dy;
  rtd::uniform_on_disk(rng, dx, dy);                                              // :41
  const V3 pos = {p.cam[9], p.cam[10], p.cam[11]};                                // Position(), :54-57
  const V3 off = {dx * p.aperture, dy * p.aperture, 0.0f};
  o = rtd::add(pos, off);                                                         // :47
  d = M::normalize(rtd::sub(focal, o));                                           // :50
}

// ------------------------------------------------------------------------------------
// One triangle against the K rays of every lane.
//
// FILTER: three wave-uniform early-outs, decided with __ballot on CONSERVATIVE per-ray
// rejections -- a ray is only ever dropped when the reference's own test is certain to
// miss, and the triangle is skipped only when every ray of the wave is dropped; whenever
// any ray survives, stage D evaluates the reference's exact test (division included) for
// all lanes from the values already computed.  With u = fl(U*inv), v = fl(V*inv),
// inv = fl(1/det), det >= 1e-10 (not culled):
//   U > fl(det*1.0001)            => U/det > 1.00009            => u > 1      (miss, :51)
//   U < fl(det*-1e-6)             => U/det < -0.99e-6 (normal)  => u < 0      (miss, :51)
//   V < fl(det*-1e-6)             =>                               v < 0      (miss, :58)
//   U+V > fl(det*1.0001), with U,V >= -1e-6 det (not dropped above)
//                                 => u+v > 1.0001 - 4e-6 - roundoff > 1       (miss, :58)
// NaN/inf operands make every comparison false: the ray is kept and stage D decides.
// tests/test_gpu_parity.py::test_filter_off_equals_filter_on checks FILTER against the
// plain reference-order path bit for bit.
// ------------------------------------------------------------------------------------
template <bool FMA, int K, bool FILTER, bool STATS, class GetB>
__device__ __forceinline__ void test_triangle(const float4 A0, const float4 A1, GetB get_v0z, int tri_index,
                                              const V3 (&o)[K], const V3 (&d)[K], float (&best_t)[K],
                                              int (&best_i)[K], bool nearest, bool counted_lane, uint32_t valid_k,
                                              unsigned long long (&st_exit)[4],
                                              unsigned long long (&st_skip)[4]) {
  using M = Math<FMA>;
  const V3 e2 = {A0.x, A0.y, A0.z}, e1 = {A0.w, A1.x, A1.y};

  if constexpr (!FILTER) {
    const V3 v0 = {A1.z, A1.w, get_v0z()};
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float t = 0.0f, u = 0.0f, v = 0.0f;
      int stage;
      const bool h = hit_triangle_exact<FMA>(o[k], d[k], v0, e1, e2, RT_EPS, t, u, v, stage);
      if (h && (nearest ? (t > 0.0f && t < best_t[k]) : best_t[k] < t)) {   // :84 (or nearest-hit extension)
        best_t[k] = t;
        best_i[k] = tri_index;
      }
      if constexpr (STATS) {
        const bool counted = counted_lane && (static_cast<uint32_t>(k) < valid_k);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          st_exit[e] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(counted && stage == e));
      }
    }
  } else {
    // stage A: pv = cross(dir, e2), det = dot(e1, pv), culling (:39-45)
    V3 pv[K];
    float det[K];
    unsigned long long mk[K];
    unsigned long long live = 0ull;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      pv[k] = M::cross(d[k], e2);
      det[k] = M::dot(e1, pv[k]);
      mk[k] = __builtin_amdgcn_ballot_w64(!(det[k] < RT_EPS));
      live |= mk[k];
    }
    if (live == 0ull) { if constexpr (STATS) st_skip[0]++; return; }   // whole wave culled

    // stage B: U = dot(origin - v0, pv) (:49-50), conservative u rejection
    const V3 v0 = {A1.z, A1.w, get_v0z()};
    V3 tv[K];
    float U[K], thi[K], tlo[K];
    live = 0ull;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      tv[k] = rtd::sub(o[k], v0);
      U[k] = M::dot(tv[k], pv[k]);
      thi[k] = det[k] * 1.0001f;
      tlo[k] = det[k] * -1e-6f;
      mk[k] &= __builtin_amdgcn_ballot_w64(!(U[k] > thi[k])) & __builtin_amdgcn_ballot_w64(!(U[k] < tlo[k]));
      live |= mk[k];
    }
    if (live == 0ull) { if constexpr (STATS) st_skip[1]++; return; }

    // stage C: V = dot(dir, cross(tv, e1)) (:56-57), conservative v rejection
    V3 qv[K];
    float V[K];
    live = 0ull;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      qv[k] = M::cross(tv[k], e1);
      V[k] = M::dot(d[k], qv[k]);
      mk[k] &= __builtin_amdgcn_ballot_w64(!(V[k] < tlo[k])) &
               __builtin_amdgcn_ballot_w64(!((U[k] + V[k]) > thi[k]));
      live |= mk[k];
    }
    if (live == 0ull) { if constexpr (STATS) st_skip[2]++; return; }
    if constexpr (STATS) st_skip[3]++;

    // stage D: the reference's exact tests (:42-63, :84) wherever a ray may hit
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (mk[k] != 0ull) {
        const float inv = 1.0f / det[k];                           // :47
        const float u = U[k] * inv;                                // :50
        const float v = V[k] * inv;                                // :57
        const float t = M::dot(e2, qv[k]) * inv;                   // :63
        const bool miss = (det[k] < RT_EPS) | (u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f);
        const bool closer = nearest ? ((t > 0.0f) & (t < best_t[k])) : (best_t[k] < t);   // :84 / nearest-hit extension
        const bool upd = (!miss) & closer;
        best_t[k] = upd ? t : best_t[k];
        best_i[k] = upd ? tri_index : best_i[k];
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// BIN: conservative classification of one triangle against the whole ray family of a tile.
//
// Ray family of a wave: every lens origin o in the box oc +- orad and, for every in-image
// pixel of the tile, its focal point F (a box fc +- frad over the 64 pixels); a ray is the line
// through o with direction w/|w|, w = F - o (ThinLensCamera.cuh:44-50).  Writing the three
// quantities of HitTriangle with the UNNORMALISED direction,
//     det' = w.(e2 x e1)      U' = (o - v0).(w x e2)      V' = w.((o - v0) x e1)
// (det, U, V of Kernels.cuh:39-57 are these divided by |w|), and o = oc + do, F = fc + dF,
// the polynomials expand EXACTLY -- the dependency between origin and direction is resolved
// analytically, which is what makes the bounds tight for in-focus geometry -- to
//     det' = wc.N + (dF - do).N                                   N = e2 x e1, wc = fc - oc
//     U'   = tvc.(wc x e2) + do.(G x e2) + dF.(e2 x tvc) + do.(dF x e2)     tvc = oc - v0
//     V'   = wc.(tvc x e1) + do.(e1 x G) + dF.(tvc x e1) + dF.(do x e1)     G = fc - v0
// so with |do_i| <= orad_i, |dF_i| <= frad_i the radii are plain absolute-value sums.  For every
// ray of the family the values the REFERENCE arithmetic computes (either math mode) satisfy
//     |det_c |w| - det'| , |U_c |w| - U'| , |V_c |w| - V'|  <=  c * (magnitude sums)
// with c = 4e-6 (~67 ulp) covering every rounding of the reference's evaluation (normalisation
// of w included, ~20 ulp) and of this one.  With *_hi / *_lo the interval ends and
// lmin <= |w| <= lmax, the triangle can be dropped for the whole tile when
//     det_hi < eps * lmin                          every ray culled (:42)
//     U_hi < -1e-6 * det_hi      (det_hi > 0)      every unculled ray has u < 0
//     U_lo > 1.0002 * det_hi                       every unculled ray has u > 1
//     V_hi < -1e-6 * det_hi                        every remaining ray has v < 0
//     U_lo + V_lo > 1.0002 * det_hi                every remaining ray has u + v > 1
//     S_hi < -2e-4 * det_hi                        the same, from S' = det' - U' - V' bounded as ONE polynomial:
//                                                  S' = Sc + dF.(N - e2 x tvc - tvc x e1) - do.(N + G x e2 + e1 x G) - bilinear terms;
//                                                  per ray U' + V' > det' + 2e-4 det_hi >= 1.0002 det'.  The gradients of the
//                                                  three polynomials largely cancel in the sum (S' / det' is the third
//                                                  barycentric coordinate), so this is the rule that drops a triangle whose
//                                                  v1-v2 edge separates it from the family: the line above needs the footprint
//                                                  to be small against BOTH other coordinates' ranges.
// because the per-ray rules proven above test_triangle() are homogeneous in |w| > 0.  Any NaN
// makes the comparisons false -> the triangle is kept and the exact tests decide.
// tests/test_gpu_parity.py::test_binning_* and tools/stress_binning.py compare BIN against the
// full scan bit for bit.
// ------------------------------------------------------------------------------------
struct TileFamily {
  float oc[3], orad[3];   // lens origin box (wave-uniform)
  float A;                // radius of the lens DISK inside that box: the aperture part of orad[0], orad[1]
  float fc[3], frad[3];   // focal point box over the tile's pixels (wave-uniform)
  float lmin, lmax;       // bounds of |F - o| over the family
  bool usable;            // false: bounds not finite -> keep every triangle
};

__device__ __forceinline__ float uniform(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
// Wave-wide min/max through ds_bpermute (__shfl_xor).  A DPP/readlane butterfly was measured
// SLOWER (C3 166.5 vs 163.3 us): the kernel is VALU-issue-bound, and the bpermute round trips run
// on the otherwise idle LDS crossbar while other waves use the VALU.
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Wave-uniform bounds of the focal points of the wave's in-image pixels (`focal` is this
// lane's focal point exactly as its rays use it).
struct FocalBounds {
  float lo[3], hi[3];
  bool ok;                // every in-image lane had a finite focal point
  bool any;               // the wave has at least one in-image lane
};

template <class SL = SlackProduct>
__device__ __forceinline__ FocalBounds focal_bounds(const TraceParams& p, V3 focal, bool inside) {
  FocalBounds b;
  const float fl[3] = {focal.x, focal.y, focal.z};
  bool finite = true;
  // A full tile (lane = x + 8 y) takes the bounds from its four corner pixels' focal points -- computed exactly as the
  // rays use them, like every lane's -- widened by what a focal point of the tile can lie off the corners' bilinear
  // interpolant (p.tile_curv, host) and by the roundings of the lanes' own evaluations (cx, cy, the matrix product, the
  // exact normalize, the fma: < 1e-6 (|focal| (1 + |cx| + |cy|) + |pos|) between a lane and the ideal function, twice): 12
  // v_readlane instead of six 6-step wave reductions.  Partial tiles (image edge) and cameras the host does not vouch for
  // (p.tile_curv <= 0) reduce over their in-image lanes.
  const bool corners = p.tile_curv > 0.0f && __builtin_amdgcn_ballot_w64(inside) == ~0ull;      // wave-uniform
  const float dev = p.tile_curv + RT_SLK(4e-6f) * p.tile_round;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    finite = finite && (__builtin_fabsf(fl[i]) <= FLT_MAX);         // false for NaN/inf
    if (corners) {
      const int v = __builtin_bit_cast(int, fl[i]);
      const float c0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 0)), c1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 7));
      const float c2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 56)), c3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 63));
      b.lo[i] = fminf(fminf(c0, c1), fminf(c2, c3)) - dev;
      b.hi[i] = fmaxf(fmaxf(c0, c1), fmaxf(c2, c3)) + dev;
    } else {
      b.lo[i] = uniform(wave_min(inside ? fl[i] : FLT_MAX));        // out-of-image lanes do not constrain
      b.hi[i] = uniform(wave_max(inside ? fl[i] : -FLT_MAX));
    }
  }
  b.ok = __builtin_amdgcn_ballot_w64(inside && !finite) == 0ull;
  b.any = __builtin_amdgcn_ballot_w64(inside) != 0ull;
  return b;
}

// Block level of the classification: the union of the four waves' focal bounds (bbox: 4 x 8 floats of LDS; contains a
// __syncthreads(), so every wave of the block calls it).
__device__ __forceinline__ FocalBounds block_focal_union(const FocalBounds& wb, float* bbox, uint32_t wave, uint32_t lane) {
  if (lane == 0u) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { bbox[wave * 8u + i] = wb.lo[i]; bbox[wave * 8u + 3 + i] = wb.hi[i]; }
    bbox[wave * 8u + 6] = wb.ok ? 1.0f : 0.0f;
    bbox[wave * 8u + 7] = wb.any ? 1.0f : 0.0f;
  }
  __syncthreads();
  FocalBounds bb;
  bb.ok = true; bb.any = false;
#pragma unroll
  for (int i = 0; i < 3; ++i) { bb.lo[i] = FLT_MAX; bb.hi[i] = -FLT_MAX; }
  for (uint32_t w = 0; w < 4u; ++w) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      bb.lo[i] = fminf(bb.lo[i], bbox[w * 8u + i]);
      bb.hi[i] = fmaxf(bb.hi[i], bbox[w * 8u + 3 + i]);
    }
    bb.ok = bb.ok && (bbox[w * 8u + 6] != 0.0f);
    bb.any = bb.any || (bbox[w * 8u + 7] != 0.0f);
  }
  return bb;
}

// Ray family over every sample of every pixel inside the focal bounds.  o = pos + (dx*aperture,
// dy*aperture, 0) with |dx|,|dy| <= 1.0000003 (sr <= 1, build-owned sincos within 2 ulp of [-1,1]).
template <class SL = SlackProduct>
__device__ __forceinline__ TileFamily make_family(const TraceParams& p, const FocalBounds& b) {
  TileFamily f;
  const float A = __builtin_fabsf(p.aperture) * RT_SLKM(2e-6f);
  f.oc[0] = p.cam[9]; f.oc[1] = p.cam[10]; f.oc[2] = p.cam[11];
  f.orad[0] = A + RT_SLK(1e-6f) * __builtin_fabsf(f.oc[0]);
  f.orad[1] = A + RT_SLK(1e-6f) * __builtin_fabsf(f.oc[1]);
  f.orad[2] = RT_SLK(1e-6f) * __builtin_fabsf(f.oc[2]);
  f.A = A;
  float lmin2 = 0.0f, lmax2 = 0.0f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float lo = b.lo[i], hi = b.hi[i];
    f.fc[i] = 0.5f * (lo + hi);
    f.frad[i] = 0.5f * (hi - lo) * RT_SLKM(1e-5f) + RT_SLK(1e-6f) * (__builtin_fabsf(lo) + __builtin_fabsf(hi));
    const float wc = f.fc[i] - f.oc[i];
    const float wr = f.frad[i] + f.orad[i] + RT_SLK(2e-7f) * (__builtin_fabsf(f.fc[i]) + __builtin_fabsf(f.oc[i]));
    const float amin = fmaxf(__builtin_fabsf(wc) - wr, 0.0f), amax = __builtin_fabsf(wc) + wr;
    lmin2 = __builtin_fmaf(amin, amin, lmin2);
    lmax2 = __builtin_fmaf(amax, amax, lmax2);
  }
  // conservative bounds, not results: the raw v_sqrt_f32 (1 ulp; a denormal operand may read as 0)
  // under the 2e-6 slack and an absolute 1e-18 instead of two 17-instruction correctly rounded sqrtf
  f.lmin = __builtin_amdgcn_sqrtf(lmin2) * RT_SLKM(-2e-6f);
  f.lmax = __builtin_amdgcn_sqrtf(lmax2) * RT_SLKM(2e-6f) + RT_SLK(1e-18f);
  f.usable = b.ok && b.any && (A <= FLT_MAX) && (f.lmax <= FLT_MAX);
  return f;
}

// Per-sample forms (FORMS = true, large-scene kernels).  For ONE ray the lens origin is known: do =
// o - oc exactly (up to the rounding already inside a_r below), only the focal point keeps its box.
// The same expansion then bounds the ray's own det', U', V' by AFFINE functions of (do.x, do.y),
//     det_hi(do) = detc - do.N + DR      U_hi/lo(do) = Uc + do.(G x e2) +- UR      V_hi/lo(do) = Vc + do.(e1 x G) +- VR
// with radii that keep every dF term, the bilinear do x dF terms at their family-wide bound and the
// same rounding allowance c.  The drop rules above, applied to that single ray, become three forms
//     F1 = U_hi + 1e-6 det_hi        F2 = V_hi + 1e-6 det_hi        F3 = 1.0002 det_hi - U_lo - V_lo
// (if det_hi <= 0 the ray is culled and any verdict is right): F_i(do) < 0 for some i  =>  the
// reference's test misses for this ray.  The focal point enters U', V', det' linearly (dF.(e2 x tvc),
// dF.(tvc x e1), dF.N) apart from the small bilinear do x dF terms, and each lane knows its own
// dF = F - fc: the linear parts are evaluated per lane from the forms' gradients g_i (kept as fp16 after a
// per-form power-of-two scaling, their quantisation and the rounding of dF charged to the constant terms), so that the focal BOX only
// bounds the bilinear terms -- which is what lifts the rejection from 61 % to ~88 % of C4's tests.
// forms[] = {F1.c0, F1.cx, F1.cy, F2.c0, F2.cx, F2.cy, F3.c0, F3.cx, F3.cy, g1.xyz, g2.xyz, g3.xyz};
// the trace loop evaluates F_i = c0 + g.dF (per lane and candidate) + cx do.x + cy do.y (per sample) and only enters the
// Moeller-Trumbore stages when some ray of the wave survives -- at C4 89 % of the candidate tests
// of a sample batch are such wave-wide misses (the candidate list covers the whole lens, one
// batch only 256 points of it).
// true = every ray of the family certainly misses this triangle (see the block comment)
// SURE (small-scene kernels): *sure_hit = every ray of the family certainly HITS this triangle in the
// reference's own arithmetic -- the mirror image of the drop rules, from the same interval ends.  With det_lo,
// U_lo, V_lo, U_hi + V_hi the ends that already contain the rounding allowance of the reference's evaluation,
//     det_lo > 1.0001 eps lmax          not culled (:42): det >= det'_lo / |w| > eps
//     U_lo >= 1e-4 det_hi (> 0)         u = fl(U * fl(1/det)) >= 0 (:51), product of two positive numbers
//     V_lo >= 1e-4 det_hi               v >= 0 (:58)
//     U_hi + V_hi <= 0.9999 det_lo      u + v <= 0.9999 (1 + 4 ulp) < 1, hence also u <= 1 (:51,:58)
//       or  S_lo >= 1e-4 det_hi         the same claim from S' = det' - U' - V' bounded as one polynomial: per ray
//                                       U' + V' <= det' - 1e-4 det_hi <= 0.9999 det'  (the tighter of the two by far: the
//                                       separate ends ignore that U', V' and det' move together across the family)
//     |e2|.|tv x e1| lmax < 1e37 det_lo t = dot(e2, qv) * inv (:63) is finite, so -FLT_MAX < t records the hit (:84)
// Any NaN makes a comparison false -> not sure.
// WHICH hit wins (farthest, Kernels.cuh:84) is decided the same way.  The reference's t = dot(e2, (o - v0) x e1) / det
// (:63) equals Nt |w| / det' with Nt = -(o - v0).N affine in the lens offset alone; two candidates of one ray share
// |w|, so A is farther than j iff qA = NtA / det'A > qj = Ntj / det'j.  q[0] is a lower bound of q over the family
// (meaningful when sure), q[1] an upper bound over the rays that may hit at all (+inf when det' may reach 0: t is
// unbounded there); both contain the allowance c for the reference's evaluation of the numerator.
// A tile in which one certainly-hit triangle A has qA_lo above every other candidate's q_hi (by 1e-4 relative, against
// the two roundings of the quotient) needs neither rays nor intersection arithmetic under the reference's flat
// shading (Kernels.cuh:95-99 uses the winner's vertices only, `hitpoint` is unused): every sample's radiance is A's
// colour.  Its samples keep their RNG draws and their additions, nothing else.  (Not with spheres, smooth normals or
// the nearest-hit rule, which need t, u, v.)
// dbg (harness only, null in every product call): the interval ends the verdicts are taken from --
// {det_lo, det_hi, U_lo, U_hi, V_lo, V_hi, Nt_lo, Nt_hi, S_lo, S_hi} (Nt only with SURE; S' = det' - U' - V').
#ifndef RT_LISTS_WAVES
#define RT_LISTS_WAVES 5      // region_lists_kernel: waves per SIMD the allocator must allow (its VGPRs are taken from the trace waves it runs beside)
#endif
#ifndef RT_TRACE_THIRD_BLOCK
#define RT_TRACE_THIRD_BLOCK true
#endif
#ifndef RT_TRACE_THIRD_WAVE
#define RT_TRACE_THIRD_WAVE (!PRE)
#endif
template <bool FORMS = false, bool SURE = false, class SL = SlackProduct, bool THIRD = !FORMS>
__device__ __forceinline__ bool tile_misses_triangle(const TileFamily& f, V3 v0, V3 e1, V3 e2, float* forms = nullptr,
                                                     bool* sure_hit = nullptr, float* q = nullptr, float* dbg = nullptr,
                                                     float* pair = nullptr) {
  // rounding allowance relative to the magnitude sums (DESIGN.md 4.1 "Rounding budget": <= ~20 half-ulps are
  // needed, 67 / 84 are charged).  RT_BIN_SLACK_SCALE exists for the teeth test of the adversarial campaign only
  // (tools/stress_boundaries.py against a build with the allowance scaled down must FIND mismatches).
  const float c = FORMS ? RT_SLK(5e-6f) : RT_SLK(4e-6f);     // + the evaluation of the forms themselves
  const float e1v[3] = {e1.x, e1.y, e1.z}, e2v[3] = {e2.x, e2.y, e2.z}, v0v[3] = {v0.x, v0.y, v0.z};
  float E1[3], E2[3], wc[3], W[3], dw[3], tvc[3], T[3], G[3], a[3], r[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    E1[i] = __builtin_fabsf(e1v[i]);
    E2[i] = __builtin_fabsf(e2v[i]);
    a[i] = f.orad[i] + RT_SLK(2e-7f) * (__builtin_fabsf(f.oc[i]) + __builtin_fabsf(v0v[i]));   // |do| incl. rounding of o - v0
    r[i] = f.frad[i];
    wc[i] = f.fc[i] - f.oc[i];
    dw[i] = r[i] + a[i] + RT_SLK(2e-7f) * (__builtin_fabsf(f.fc[i]) + __builtin_fabsf(f.oc[i]));   // |dF - do|
    W[i] = __builtin_fabsf(wc[i]) + dw[i];                          // >= |w_i| for every ray
    tvc[i] = f.oc[i] - v0v[i];
    T[i] = __builtin_fabsf(tvc[i]) + a[i];                          // >= |(o - v0)_i|
    G[i] = f.fc[i] - v0v[i];
  }
  float detc = 0.0f, det_rad = 0.0f, Uc = 0.0f, U_rad = 0.0f, Vc = 0.0f, V_rad = 0.0f;
  float S_rad = 0.0f, cs[3];                                        // S' = det' - U' - V' bounded as ONE polynomial (see below)
  float DR = 0.0f, UR = 0.0f, VR = 0.0f;                            // FORMS: radii for a known origin
  float tmag = 0.0f;                                                // SURE: >= |dot(e2, (o - v0) x e1)|
  float Ntc = 0.0f, Nt_rad = 0.0f;                                  // SURE: Nt = -(o - v0).N at the lens centre, radius over the lens
  float wn = 0.0f;                                                  // SURE, pair: the magnitude sum of det' (its rounding allowance is c * wn)
  float Nv[3] = {0.0f, 0.0f, 0.0f}, Gu[3] = {0.0f, 0.0f, 0.0f}, Gv[3] = {0.0f, 0.0f, 0.0f};
  float Eu[3] = {0.0f, 0.0f, 0.0f}, Ev[3] = {0.0f, 0.0f, 0.0f}, qd[3] = {0.0f, 0.0f, 0.0f};
  float cn[3], cu[3], cv[3];                                        // |coefficient| of do_i in det', U', V' (lens terms)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int j = (i + 1) % 3, k = (i + 2) % 3;                     // cross(x, y)_i = x_j*y_k - y_j*x_k
    const float N_i = e2v[j] * e1v[k] - e1v[j] * e2v[k];            // (e2 x e1)_i
    const float Nabs = E2[j] * E1[k] + E1[j] * E2[k];
    const float wxe2 = wc[j] * e2v[k] - e2v[j] * wc[k];             // (wc x e2)_i
    const float Wxe2 = W[j] * E2[k] + E2[j] * W[k];                 // >= |(w x e2)_i|
    const float Gxe2 = G[j] * e2v[k] - e2v[j] * G[k];               // (G x e2)_i
    const float e2xt = e2v[j] * tvc[k] - tvc[j] * e2v[k];           // (e2 x tvc)_i
    const float rxe2 = r[j] * E2[k] + E2[j] * r[k];                 // >= |(dF x e2)_i|
    const float txe1 = tvc[j] * e1v[k] - e1v[j] * tvc[k];           // (tvc x e1)_i
    const float Txe1 = T[j] * E1[k] + E1[j] * T[k];                 // >= |((o - v0) x e1)_i|
    const float e1xG = e1v[j] * G[k] - G[j] * e1v[k];               // (e1 x G)_i
    const float axe1 = a[j] * E1[k] + E1[j] * a[k];                 // >= |(do x e1)_i|
    const float Gabs = (__builtin_fabsf(G[j]) * E2[k] + E2[j] * __builtin_fabsf(G[k])) +
                       (E1[j] * __builtin_fabsf(G[k]) + __builtin_fabsf(G[j]) * E1[k]);   // rounding of G x e2, e1 x G
    detc += wc[i] * N_i;
    det_rad += dw[i] * __builtin_fabsf(N_i) + c * (W[i] * Nabs);
    Uc += tvc[i] * wxe2;
    U_rad += a[i] * (__builtin_fabsf(Gxe2) + rxe2) + r[i] * __builtin_fabsf(e2xt) + c * (T[i] * Wxe2 + a[i] * Gabs);
    Vc += wc[i] * txe1;
    V_rad += a[i] * __builtin_fabsf(e1xG) + r[i] * (__builtin_fabsf(txe1) + axe1) + c * (W[i] * Txe1 + a[i] * Gabs);
    cn[i] = __builtin_fabsf(N_i); cu[i] = __builtin_fabsf(Gxe2) + rxe2; cv[i] = __builtin_fabsf(e1xG);
    // S' = det' - U' - V' (det' times the third barycentric coordinate) = Sc + dF.(N - e2 x tvc - tvc x e1) - do.(N + G x e2 + e1 x G)
    // - the two bilinear terms: the three gradients largely cancel (for a ray family inside the triangle's plane footprint
    // |N - ...| is the gradient of ONE edge function), which the sum of the separate interval ends cannot see.
    if constexpr (THIRD) {
      cs[i] = __builtin_fabsf((N_i + Gxe2) + e1xG);
      S_rad += (r[i] * __builtin_fabsf((N_i - e2xt) - txe1) + a[i] * cs[i]) + (a[i] * rxe2 + r[i] * axe1) +
               c * ((W[i] * Nabs + T[i] * Wxe2) + (W[i] * Txe1 + 2.0f * (a[i] * Gabs)));
    } else {
      cs[i] = 0.0f;
    }
    if constexpr (SURE) {
      tmag += E2[i] * Txe1;
      Ntc -= tvc[i] * N_i;
      Nt_rad += a[i] * __builtin_fabsf(N_i);
      if (pair != nullptr) { pair[2 + i] = N_i; wn += W[i] * Nabs; }
    }
    if constexpr (FORMS) {
      // a_r: what is left of |do_i| once the sample's own origin is used -- the roundings of o = pos + off
      // and of o - v0 (the aperture part A of orad is the known do itself)
      const float a_r = RT_SLK(1e-6f) * __builtin_fabsf(f.oc[i]) + RT_SLK(2e-7f) * (__builtin_fabsf(f.oc[i]) + __builtin_fabsf(v0v[i]));
      // The terms LINEAR in dF -- dF.N, dF.(e2 x tvc), dF.(tvc x e1) -- are not bounded over the box but
      // evaluated per lane from its own dF = F - fc (gradients Eu, Ev, N below); only the bilinear
      // do x dF terms keep their family-wide bound.
      const float dw_r = a_r + RT_SLK(2e-7f) * (__builtin_fabsf(f.fc[i]) + __builtin_fabsf(f.oc[i]));
      DR += dw_r * __builtin_fabsf(N_i) + c * (W[i] * Nabs);
      UR += a_r * __builtin_fabsf(Gxe2) + a[i] * rxe2 + c * (T[i] * Wxe2 + a[i] * Gabs);
      VR += a_r * __builtin_fabsf(e1xG) + r[i] * axe1 + c * (W[i] * Txe1 + a[i] * Gabs);
      Nv[i] = N_i; Gu[i] = Gxe2; Gv[i] = e1xG; Eu[i] = e2xt; Ev[i] = txe1;
      // what a lane's dF can be off by: the rounding of F - fc itself (the storage of the gradients is charged below)
      qd[i] = RT_SLK(2e-7f) * (__builtin_fabsf(f.fc[i]) + r[i]);
    }
  }
  if constexpr (FORMS) {
    DR *= RT_SLKM(1e-5f); UR *= RT_SLKM(1e-5f); VR *= RT_SLKM(1e-5f);
    const float dh = detc + DR;                                     // det_hi at do = 0
    forms[0] = (Uc + UR) + 1e-6f * dh;                              // F1 = U_hi + 1e-6 det_hi
    forms[1] = Gu[0] - 1e-6f * Nv[0];
    forms[2] = Gu[1] - 1e-6f * Nv[1];
    forms[3] = (Vc + VR) + 1e-6f * dh;                              // F2 = V_hi + 1e-6 det_hi
    forms[4] = Gv[0] - 1e-6f * Nv[0];
    forms[5] = Gv[1] - 1e-6f * Nv[1];
    forms[6] = 1.0002f * dh - (Uc - UR) - (Vc - VR);                // F3 = 1.0002 det_hi - U_lo - V_lo
    forms[7] = -1.0002f * Nv[0] - Gu[0] - Gv[0];
    forms[8] = -1.0002f * Nv[1] - Gu[1] - Gv[1];
#pragma unroll
    for (int i = 0; i < 3; ++i) {                                   // gradients with respect to the lane's dF
      const float g1 = Eu[i] + 1e-6f * Nv[i], g2 = Ev[i] + 1e-6f * Nv[i], g3 = 1.0002f * Nv[i] - Eu[i] - Ev[i];
      forms[9 + i] = g1; forms[12 + i] = g2; forms[15 + i] = g3;
      forms[0] += qd[i] * __builtin_fabsf(g1) * RT_SLKM(1e-5f);
      forms[3] += qd[i] * __builtin_fabsf(g2) * RT_SLKM(1e-5f);
      forms[6] += qd[i] * __builtin_fabsf(g3) * RT_SLKM(1e-5f);
    }
    // The gradients are kept as fp16 (v_fma_mix_f32 reads the halves in place: no unpacking in the per-sample loop).  Only
    // the SIGN of a form matters, so each form is first scaled by the power of two that brings its largest gradient
    // component into [2^13, 2^14) -- exact, and far from fp16's overflow -- and then rounded to nearest: a component is
    // off by at most 2^-11 of itself (normal range) or 2^-25 (below 2^-14), times |dF_i| <= r_i; charged to the constant
    // term, not scaled with the rounding allowances (it is a bound on a known quantisation).
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float m = fmaxf(fmaxf(__builtin_fabsf(forms[9 + 3 * k]), __builtin_fabsf(forms[10 + 3 * k])), __builtin_fabsf(forms[11 + 3 * k]));
      int n = 14 - __builtin_amdgcn_frexp_expf(m);
      n = (m > 0.0f && m <= FLT_MAX) ? (n < -100 ? -100 : n > 100 ? 100 : n) : 0;
      float quant = 0.0f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float gs = __builtin_ldexpf(forms[9 + 3 * k + j], n);
        const float gq = static_cast<float>(static_cast<_Float16>(gs));
        quant += (0.00048828125f * __builtin_fabsf(gs) + 2.98023224e-8f) * r[j];
        forms[9 + 3 * k + j] = gq;
      }
      forms[3 * k] = __builtin_ldexpf(forms[3 * k], n) + quant * 1.001f;
      forms[3 * k + 1] = __builtin_ldexpf(forms[3 * k + 1], n);
      forms[3 * k + 2] = __builtin_ldexpf(forms[3 * k + 2], n);
    }
  }
  // The lens is a DISK of radius A, the sums above took it as the box [-A, A]^2: a term do.g (do_z = 0) was charged
  // A (|g_x| + |g_y|) where A |g_xy| suffices, and max + 0.4143 min >= sqrt(max^2 + min^2).  Take the difference back,
  // a little less than all of it (0.585 < 1 - 0.41422).  (The rounding parts of a[], the bilinear do x dF term of V' and
  // the magnitude bounds W, T keep the box.)
  const float disk = 0.585f * f.A;
  const float dn = disk * fminf(cn[0], cn[1]);
  det_rad -= dn;
  U_rad -= disk * fminf(cu[0], cu[1]);
  V_rad -= disk * fminf(cv[0], cv[1]);
  if constexpr (SURE) Nt_rad -= dn;
  S_rad -= disk * fminf(cs[0], cs[1]);
  det_rad = det_rad * RT_SLKM(1e-5f);
  U_rad = U_rad * RT_SLKM(1e-5f);
  V_rad = V_rad * RT_SLKM(1e-5f);
  S_rad = S_rad * RT_SLKM(1e-5f);
  const float det_hi = detc + det_rad;
  const float U_lo = Uc - U_rad, U_hi = Uc + U_rad, V_lo = Vc - V_rad, V_hi = Vc + V_rad;
  const float Sc = (detc - Uc) - Vc;
  // (THIRD = false -- the wave-level call of the dense-scene kernels, whose per-sample form F3 is this rule per ray and whose
  // 128-VGPR budget the extra sums overflow: 24 -> 92 bytes of scratch per lane -- leaves the S rules out: -inf / +inf)
  const float S_lo = THIRD ? Sc - S_rad : -__builtin_inff(), S_hi = THIRD ? Sc + S_rad : __builtin_inff();   // bounds of the reference's det' - U' - V' per ray
  const float neg = det_hi * -1e-6f, big = det_hi * 1.0002f;
  if (dbg != nullptr) {
    dbg[0] = detc - det_rad; dbg[1] = det_hi; dbg[2] = U_lo; dbg[3] = U_hi; dbg[4] = V_lo; dbg[5] = V_hi; dbg[8] = S_lo; dbg[9] = S_hi;
  }
  if constexpr (SURE) {
    const float det_lo = detc - det_rad;
    *sure_hit = (det_lo > (RT_EPS * 1.0001f) * f.lmax) && (U_lo >= 1e-4f * det_hi) && (V_lo >= 1e-4f * det_hi) &&
                (((U_hi + V_hi) <= 0.9999f * det_lo) || (S_lo >= 1e-4f * det_hi)) && (tmag * f.lmax < 1e37f * det_lo);
    const float nt_rad = (Nt_rad + c * tmag) * RT_SLKM(1e-5f);
    const float nt_lo = Ntc - nt_rad, nt_hi = Ntc + nt_rad;
    if (dbg != nullptr) { dbg[6] = nt_lo; dbg[7] = nt_hi; }
    const float inv_lo = __builtin_amdgcn_rcpf(det_lo), inv_hi = __builtin_amdgcn_rcpf(det_hi);   // (1 ulp: far inside the 1e-4 margin of the comparison)
    q[0] = (nt_lo >= 0.0f) ? nt_lo * inv_hi : nt_lo * inv_lo;
    q[1] = (det_lo > 0.0f) ? ((nt_hi >= 0.0f) ? nt_hi * inv_lo : nt_hi * inv_hi) : __builtin_inff();
    if (pair != nullptr) {          // what pair_farther() needs of this triangle: the polynomials' centres, gradient and allowances
      pair[0] = Ntc; pair[1] = detc;                                 // Nt = Ntc - do.N, det' = detc + (dF - do).N; pair[2..4] = N
      pair[5] = c * tmag; pair[6] = c * wn;                          // allowances for the reference's Nt |w| and det |w|
      pair[7] = fmaxf(__builtin_fabsf(nt_lo), __builtin_fabsf(nt_hi));   // >= |Nt| over the family
      pair[8] = fmaxf(__builtin_fabsf(det_lo), __builtin_fabsf(det_hi)); // >= |det'|
    }
  }
  const bool all_culled = det_hi < RT_EPS * f.lmin;
  const bool pos = det_hi > 0.0f;
  const bool out = pos && ((U_hi < neg) || (U_lo > big) || (V_hi < neg) || ((U_lo + V_lo) > big) || (S_hi < det_hi * -2e-4f));
  return all_culled || out;
}

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
// Focal bounds of the pixel rectangle [x0, x0 + p.macro_w) x [y0, y0 + p.macro_h) of the band (clipped to it), by all 256
// threads of the block: every pixel's focal point exactly as the trace kernel computes it.  s_box: 4 x 8 floats of LDS.
template <bool FMA>
__device__ __forceinline__ FocalBounds macro_focal_bounds(const TraceParams& p, uint32_t x0, uint32_t y0, float (*s_box)[8]) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t x1 = (x0 + p.macro_w < p.W) ? x0 + p.macro_w : p.W;
  const uint32_t y1 = (y0 + p.macro_h < p.rows) ? y0 + p.macro_h : p.rows;
  const uint32_t w = x1 - x0, count_px = w * (y1 - y0);
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  bool finite = true;
  for (uint32_t i = threadIdx.x; i < count_px; i += 256u) {
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
  for (uint32_t v = 0; v < 4u; ++v) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      bb.lo[c] = fminf(bb.lo[c], s_box[v][c]);
      bb.hi[c] = fmaxf(bb.hi[c], s_box[v][3 + c]);
    }
    bb.ok = bb.ok && (s_box[v][6] != 0.0f);
  }
  return bb;
}

template <bool FMA>
__global__ __launch_bounds__(256) void macro_bin_kernel(TraceParams p) {
  __shared__ float s_box[4][8];
  __shared__ uint32_t s_cnt[2][4];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const FocalBounds bb = macro_focal_bounds<FMA>(p, blockIdx.x * p.macro_w, blockIdx.y * p.macro_h, s_box);
  const TileFamily fam = make_family(p, bb);
  uint32_t* const out = p.macro_lists + (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * (p.macro_cap + 1u);
  const uint32_t n = p.n_tris;
  uint32_t total = 0, step = 0;
  bool overflow = false;
  for (uint32_t base = 0; base < n; base += 256u, ++step) {
    const uint32_t tri = base + threadIdx.x;
    const bool valid = tri < n;
    const uint32_t ti = valid ? tri : (n - 1u);
    const float4 A0 = p.tri_a[2u * ti], A1 = p.tri_a[2u * ti + 1u];
    const float bz = p.tri_b[ti];
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
    if (total + step_total > p.macro_cap) { overflow = true; break; }   // block-uniform
    const uint32_t pos = total + before + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                                __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
    if (keep) out[1u + pos] = tri;                                // ascending order across waves and steps
    total += step_total;
  }
  if (threadIdx.x == 0u) out[0] = overflow ? 0xFFFFFFFFu : total;
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
template <bool FMA, class SL>
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
  const uint32_t quad_bad = (static_cast<uint32_t>(badm >> (lane & 28u)) & 0xFu);                 // lanes 0..31 mirror 32..63
  tile_b.ok = quad_bad == 0u;
  tile_b.any = valid;
  const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
  region_b.any = (vm & 0xFFFFFFFFull) != 0ull;
  region_b.ok = ((badm & vm) & 0xFFFFFFFFull) == 0ull;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    region_b.lo[i] = group_min<32>(rlo[i]);
    region_b.hi[i] = group_max<32>(rhi[i]);
  }
}

// Is A farther than B for every ray of the family that hits both?  The reference's t = dot(e2, qv) / det (Kernels.cuh:63) is
// Nt |w| / det' with Nt = Ntc - do.N and det' = detc + (dF - do).N, both per triangle; two candidates of one ray share |w|, so
// with both det' > 0 (A certainly hit, B hit by this ray)  t_A > t_B  <=>  D = Nt_A det'_B - Nt_B det'_A > 0.
//     D = Dc + dF.gF + do.gO + (do x dF).(N_B x N_A)      gF = nA N_B - nB N_A      gO = -gF - dB N_A + dA N_B
// -- affine in the lens offset and the focal offset but for one small bilinear term, where the per-triangle q intervals
// of two triangles at similar depth (a box on the floor, a light under the ceiling) overlap however tight they are.  Charged:
// the radii over |do_i| <= orad_i, |dF_i| <= frad_i; the reference's evaluation of the four factors (the allowances of the
// q bounds, cross-multiplied with the bounds of the other factor); 1e-4 relative for the roundings of the two quotients and
// of this evaluation, as the q comparison does.  a, b: pair[] of tile_misses_triangle.  Any NaN: false.
template <class SL>
__device__ __forceinline__ bool pair_farther(const TileFamily& f, const float* a, const float* b) {
  const float nA = a[0], dA = a[1], nB = b[0], dB = b[1];
  const float Dc = nA * dB - nB * dA;
  float rad = 0.0f, gFv[3], gOv[3], cr[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    gFv[i] = nA * b[2 + i] - nB * a[2 + i];
    gOv[i] = (dA * b[2 + i] - dB * a[2 + i]) - gFv[i];
    cr[i] = __builtin_fabsf(b[2 + j] * a[2 + k] - a[2 + j] * b[2 + k]);   // |(N_B x N_A)_i|
    rad += f.frad[i] * __builtin_fabsf(gFv[i]) + f.orad[i] * __builtin_fabsf(gOv[i]) + cr[i] * (f.orad[j] * f.frad[k] + f.orad[k] * f.frad[j]);
  }
  const float rnd = (a[5] * b[8] + a[7] * b[6]) + (b[5] * a[8] + b[7] * a[6]) + a[5] * b[6] + b[5] * a[6];
  const float mag = a[7] * b[8] + b[7] * a[8];
  return (Dc - (rad + rnd) * RT_SLKM(1e-5f)) - 1e-4f * mag > 0.0f;
}

// 8-lane groups: max in every lane of the group (quad_perm x 2, row_half_mirror)
__device__ __forceinline__ float max8(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  return v;
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
  const uint32_t t8 = lane >> 3, j = lane & 7u;
  FocalBounds mine;                                                  // tile t8's box: from lane 4 * t8
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    mine.lo[i] = __shfl(tb.lo[i], static_cast<int>(4u * t8), 64);
    mine.hi[i] = __shfl(tb.hi[i], static_cast<int>(4u * t8), 64);
  }
  const unsigned long long okm = __builtin_amdgcn_ballot_w64(tb.ok), anym = __builtin_amdgcn_ballot_w64(tb.any);
  mine.ok = ((okm >> (4u * t8)) & 1ull) != 0ull;
  mine.any = ((anym >> (4u * t8)) & 1ull) != 0ull;
  const bool tile_valid = mine.any;                                  // the tile has pixels in the band
  const TileFamily fam = make_family<SL>(p, mine);
  const uint32_t tslot = ((ry * 2u + (t8 >> 2)) * gx + rx) * 4u + (t8 & 3u);
  uint32_t* const saved = p.tile_lists + static_cast<size_t>(tile_valid ? tslot : 0u) * (1u + p.bin_list);
  const bool slot_live = (ry * 2u + (t8 >> 2)) < gy;                 // (a slot right of the image exists and gets an empty list)
  const float NEG = -__builtin_inff();
  uint32_t count = 0;
  bool haveA = false;
  float Q = NEG, M1 = NEG, M2 = NEG;
  uint32_t A = 0, I1 = 0xFFFFFFFFu;
  const uint32_t gsh = lane & 56u;                                   // first lane of this 8-lane group
  const bool one_pass = cnt <= 8u;                                   // every candidate of the region has a lane: pairs can be compared
  bool others_ok = true;                                             // one_pass: every other kept triangle is certainly nearer than A
  for (uint32_t c0 = 0; c0 < cnt; c0 += 8u) {                        // (wave-uniform trip count)
    const bool has = tile_valid && c0 + j < cnt;
    const uint32_t tri = cand[(c0 + j < cnt) ? c0 + j : 0u];
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
    const uint32_t gm = static_cast<uint32_t>(__builtin_amdgcn_ballot_w64(keep) >> gsh) & 0xFFu;
    const uint32_t pos = count + static_cast<uint32_t>(__builtin_popcount(gm & ((1u << j) - 1u)));
    if (keep) saved[1u + pos] = tri;                                 // ascending: candidates and passes ascend
    count += static_cast<uint32_t>(__builtin_popcount(gm));
    const bool cd = keep && sure && fam.usable;
    const float Qs = max8(cd ? q[0] : NEG);
    const uint32_t gbm = static_cast<uint32_t>(__builtin_amdgcn_ballot_w64(cd && q[0] == Qs) >> gsh) & 0xFFu;
    if (gbm != 0u && (!haveA || Qs > Q)) { haveA = true; Q = Qs; A = cand[c0 + static_cast<uint32_t>(__builtin_ctz(gbm))]; }
    const float qh = keep ? ((q[1] == q[1]) ? q[1] : __builtin_inff()) : NEG;
    if (one_pass) {
      // A's rivals one by one: nearer by the q intervals, or -- where those overlap -- by the pairwise bound (pair_farther)
      const uint32_t la = gbm != 0u ? static_cast<uint32_t>(__builtin_ctz(gbm)) : 0xFFFFFFFFu;
      bool lane_ok = !keep || j == la || (gbm != 0u && qh < Qs - 1e-4f * (__builtin_fabsf(qh) + __builtin_fabsf(Qs)));
      const bool need = gbm != 0u && !lane_ok;
      if (__builtin_amdgcn_ballot_w64(need) != 0ull) {               // (wave-uniform: rare)
        float pa[9];
        const int src = static_cast<int>(gsh + (la & 7u));
#pragma unroll
        for (int i = 0; i < 9; ++i) pa[i] = __shfl(pr[i], src, 64);
        lane_ok = lane_ok || (need && pair_farther<SL>(fam, pa, pr));
      }
      others_ok = (static_cast<uint32_t>(__builtin_amdgcn_ballot_w64(!lane_ok) >> gsh) & 0xFFu) == 0u;
    }
    const float m1s = max8(qh);
    const uint32_t gtm = static_cast<uint32_t>(__builtin_amdgcn_ballot_w64(keep && qh == m1s) >> gsh) & 0xFFu;
    const uint32_t l1 = gtm != 0u ? static_cast<uint32_t>(__builtin_ctz(gtm)) : 0xFFFFFFFFu;
    const float m2s = max8((keep && j != l1) ? qh : NEG);
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
    const bool lower = (ry * 2u + (t8 >> 2)) >= p.cost_split_brow;
    const uint32_t nu = static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(rays && !lower)));
    const uint32_t nl = static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(rays && lower)));
    if (lane == 0u) {
      if (nu != 0u) atomicAdd(p.half_cost, nu);
      if (nl != 0u) atomicAdd(p.half_cost + 1, nl);
    }
  }
}

// ------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------
// The trace kernel.  grid = (ceil(W/32), ceil(rows/8)), block = 256 threads.
// Dynamic LDS: BIN ? 4 waves * bin_list * 40 bytes (104 with the per-sample forms, PRE) + block_list * 4
// + 160 bytes for the block-level pre-cull : min(n_tris, chunk) * 36 bytes  (trace_lds_bytes()).
// ------------------------------------------------------------------------------------
// FUSE: the launch runs p.iters consecutive iterations of the host loop (RayTracerImpl.cu:246-249)
// of p.samples samples each: the per-iteration `render += accu` (:141-143) keeps its order of
// additions, so the buffers end bit-identical to p.iters separate launches -- without their
// state traffic, tile family and classification.  Used between two update points of a Trace.
// PRE: large-scene kernels (BIN && !ONEPASS) with the per-sample forms; a separate instantiation because
// the first classification then moves in front of the sample loop and the forms cost registers and code
// that sparser scenes do not earn back (300-1000 triangles at 1080p: +6-10 % with them, C4: -13 %).
template <bool FMA, int K, bool FILTER, bool STATS, bool BIN, bool ONEPASS, bool FUSE = false, bool PRE = false>
__global__ __launch_bounds__(256, (ONEPASS && K == 2) ? 5 : RT_TRACE_WAVES(K)) void trace_kernel(const TraceParams p) {
  using M = Math<FMA>;
  extern __shared__ float4 s_mem[];

  // (workgroup shape, measured at C3 with the lists rebuilt: 256 threads = one block of four tiles 62.5 us per step; one wave
  //  per workgroup 85.0; two / four blocks per workgroup 67.2 / 75.2; 8 instead of 7 waves per SIMD at 64 VGPRs 63.2)
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t bx = blockIdx.x;                                  // the block of 32 x 8 pixels this workgroup traces
  uint32_t by = blockIdx.y;
  const uint32_t gxb = gridDim.x;                                  // blocks per block row
  if constexpr (BIN && ONEPASS) {
    if (p.row_il != 0u) by = (blockIdx.y / p.row_il) * (2u * p.row_il) + p.row_phase * p.row_il + blockIdx.y % p.row_il;
  }
  const uint32_t px = bx * 32u + wave * 8u + (lane & 7u);
  const uint32_t ly = by * 8u + (lane >> 3);
  const bool inside = px < p.W && ly < p.rows;
  const uint32_t cxp = inside ? px : 0u, cyp = inside ? ly : 0u;   // out-of-image lanes shadow pixel 0
  const size_t pix = static_cast<size_t>(cxp) + static_cast<size_t>(cyp) * p.W;   // Kernels.cuh:128

#ifdef RT_TIMELINE
  // experiment builds only: per-wave timestamps (shader clock) + where the wave ran
  const size_t tl_slot = ((static_cast<size_t>(by) * gxb + bx) * 4u + wave) * 16u;
  auto tl_mark = [&](uint32_t i) {
    __builtin_amdgcn_sched_barrier(0);
    if (p.timeline != nullptr && lane == 0u) p.timeline[tl_slot + i] = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
  };
  if (p.timeline != nullptr && lane == 0u) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    p.timeline[tl_slot + 6] = hw;
    p.timeline[tl_slot + 7] = (static_cast<unsigned long long>(xcc) << 32) | (__builtin_amdgcn_s_memrealtime() & 0xffffffffull);
  }
  tl_mark(0);
#else
  auto tl_mark = [](uint32_t) {};
#endif
  Rng rng;                                                         // :131
  auto load_rng = [&] {
    rng.d = p.rng[0 * static_cast<size_t>(p.npix) + pix];
    rng.v0 = p.rng[1 * static_cast<size_t>(p.npix) + pix];
    rng.v1 = p.rng[2 * static_cast<size_t>(p.npix) + pix];
    rng.v2 = p.rng[3 * static_cast<size_t>(p.npix) + pix];
    rng.v3 = p.rng[4 * static_cast<size_t>(p.npix) + pix];
    rng.v4 = p.rng[5 * static_cast<size_t>(p.npix) + pix];
  };
  // Dense-scene kernels with the per-sample forms (PRE) classify BEFORE any ray exists; six state registers held across the
  // block- and wave-level classification are what made the 128-VGPR kernel spill (32 bytes of scratch per lane): their state
  // is requested behind the classification instead (one exposed load latency per wave, hidden by the other three waves).
  constexpr bool RNG_LATE = PRE && BIN && !ONEPASS;
  // The third-edge rules of tile_misses_triangle inside this kernel (block and wave level of dense scenes; the macro level and the
  // small scenes' list builders always have them): at the block level always; at the wave level of the instantiations with the
  // per-sample forms not inside the forms call (K = 4: 24 -> 92 bytes of scratch per lane) but as a separate call in front
  // of it.  C4, interleaved on one device: macro level only 3.97 ms, + block level 3.75 ms, + wave level inside the forms
  // call 4.55 ms, as a call of its own 3.68 ms.
  constexpr bool THIRD_BLOCK = RT_TRACE_THIRD_BLOCK, THIRD_WAVE = RT_TRACE_THIRD_WAVE;
  if constexpr (!RNG_LATE) load_rng();


  // Small scenes (ONEPASS): the tile's candidate list and its certain-winner verdict were built ahead of this launch by
  // tile_lists_kernel (p.tile_lists; null only for a scene without triangles), so the wave knows the verdict before
  // anything else: a tile with a certain winner needs no pinhole ray either (its samples keep their RNG draws and
  // additions only).
  const bool sure_ok = BIN && ONEPASS && (p.flags & (TRACE_NEAREST_HIT | TRACE_NO_SURE_HIT)) == 0u && p.n_spheres == 0u && p.tri_n == nullptr;
  bool loaded_sure = false;
  uint32_t list_word = 0u;                                          // count | winner << 10 | certain << 31
  if constexpr (BIN && ONEPASS) {
    if (p.tile_lists != nullptr) {
      const size_t slot0 = (static_cast<size_t>(by) * gxb + bx) * 4u + wave;
      // wave-uniform by construction; readfirstlane tells the compiler (scalar loop control below)
      list_word = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(p.tile_lists[slot0 * (1u + p.bin_list)])));
      loaded_sure = sure_ok && (list_word >> 31) != 0u;
    }
  }
  V3 po = {0.0f, 0.0f, 0.0f}, pd = {0.0f, 0.0f, 0.0f}, focal = {0.0f, 0.0f, 0.0f};
  if (!loaded_sure) {                                               // wave-uniform
    pinhole<FMA>(p, cxp, p.row0 + cyp, po, pd);
    focal = focal_point<FMA>(p, pd);
  }

  const uint32_t n = p.n_tris;
  // false: the reference's rule (keep the farthest t, negative t accepted, Kernels.cuh:73,84);
  // true: build-defined extension, keep the nearest t > 0.  Wave-uniform.
  const bool nearest = (p.flags & TRACE_NEAREST_HIT) != 0u;
  float ax = 0.0f, ay = 0.0f, az = 0.0f;                           // accu, :133
  unsigned long long st_exit[4] = {0, 0, 0, 0};                    // STATS: lane-tests by exit point
  unsigned long long st_skip[4] = {0, 0, 0, 0};                    // STATS: wave-triangles skipped after A/B/C, reaching D
  unsigned long long st_bin[2] = {0, 0};                           // STATS: candidates kept, classification rounds
  unsigned long long st_pre = 0;                                   // STATS: candidate tests of a sample batch skipped by the per-sample forms

  // ---- full-scan staging (BIN == false) --------------------------------------------------
  const uint32_t cap = n < p.chunk ? n : p.chunk;                  // triangles resident in LDS
  float4* const sA = s_mem;                                        // 2 float4 per triangle
  float* const sB = reinterpret_cast<float*>(s_mem + 2u * cap);    // 1 float per triangle
  const bool single_chunk = n <= p.chunk;
  if constexpr (!BIN) {
    if (single_chunk) {
      for (uint32_t i = threadIdx.x; i < 2u * n; i += 256u) sA[i] = p.tri_a[i];
      for (uint32_t i = threadIdx.x; i < n; i += 256u) sB[i] = p.tri_b[i];
      __syncthreads();
    }
  }

  // ---- per-wave candidate list (BIN == true) ---------------------------------------------
  const uint32_t L = p.bin_list;
  float4* const cA = s_mem + static_cast<size_t>(wave) * (2u * L);                         // 2 float4 per candidate
  float* const cB = reinterpret_cast<float*>(s_mem + 4u * 2u * L) + wave * L;
  int* const cI = reinterpret_cast<int*>(s_mem + 4u * 2u * L) + 4u * L + wave * L;
  // PRETEST (large-scene kernels, TRACE_PRETEST): one 64-byte block more per candidate behind the index
  // array -- the three per-sample forms of tile_misses_triangle<true> (9 floats) and their focal-point
  // gradients (9 fp16 in 5 dwords), contiguous so that one address register and four ds_read_b128
  // with immediate offsets fetch them.  L is a multiple of 2 here.
  constexpr bool PRETEST = PRE && BIN && !ONEPASS;
  const bool pretest = PRETEST && (p.flags & TRACE_PRETEST) != 0u;   // wave-uniform
  float4* const cP = s_mem + 4u * 2u * L + 2u * L + static_cast<size_t>(wave) * (4u * L);   // after cA (8L float4), cB + cI (2L float4)
  const uint32_t list_floats4 = pretest ? (4u * 2u * L + 2u * L + 16u * L) : (4u * 2u * L + 2u * L);   // float4 units before the block list

  TileFamily fam;
  bool list_complete = false;       // the list in LDS covers the whole scene (classification done once)
  uint32_t list_count = 0;
  // ONEPASS: one triangle that every ray of the tile's family certainly hits is certainly the farthest hit of every ray
  // (wave-uniform): the sample loop then needs neither rays nor tests (tile_misses_triangle<.., SURE>)
  bool sure_hit_tile = false;
  uint32_t sure_winner = 0;         // triangle index of the certain winner
  // Block-level pre-cull (scenes larger than the per-wave list): the 256 threads classify every
  // triangle ONCE against the union of the block's four tile families and keep the survivors'
  // indices, in ascending order, in LDS; each wave then only refines that short list against its
  // own tile.  4x fewer classifications and triangle-list reads than every wave scanning the scene.
  const uint32_t Lb = p.block_list;
  uint32_t* const bI = reinterpret_cast<uint32_t*>(s_mem + list_floats4);                    // Lb indices
  uint32_t* const bcnt = bI + Lb;                                                           // 2 x 4 wave counts
  float* const bbox = reinterpret_cast<float*>(bcnt + 8);                                   // 4 waves x (lo[3], hi[3], ok, any)
  uint32_t src_count = n;           // triangles the wave-level classification walks over
  bool src_is_block_list = false;
  // macro level: this block's macro tile already excludes most of the scene (macro_bin_kernel)
  const uint32_t* mI = nullptr;     // ascending triangle indices of the macro tile, or null = whole scene
  if constexpr (BIN && !ONEPASS) {
    if (p.macro_lists != nullptr) {
      const uint32_t mt = (blockIdx.y * 8u / p.macro_h) * p.macro_nx + (blockIdx.x * 32u / p.macro_w);
      const uint32_t* const ml = p.macro_lists + static_cast<size_t>(mt) * (p.macro_cap + 1u);
      const uint32_t mc = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ml[0])));
      if (mc != 0xFFFFFFFFu) { mI = ml + 1; src_count = mc; }
    }
  }
  const uint32_t n_src = src_count;  // triangles the block-level pre-cull walks over
  // (small-scene kernels load their tiles' lists and need no ray family at all)
  if constexpr (BIN && ONEPASS) fam.usable = false;
  if constexpr (BIN && !ONEPASS) {
    tl_mark(8);                                                    // loads issued, pinhole + focal point done
    const FocalBounds wb = focal_bounds(p, focal, inside);
    tl_mark(9);
    fam = make_family(p, wb);
    tl_mark(10);
    {
      if (Lb != 0u) {
        const FocalBounds bb = block_focal_union(wb, bbox, wave, lane);
        const TileFamily bfam = make_family(p, bb);
        uint32_t total = 0;
        bool overflow = false;
        uint32_t step = 0;
        for (uint32_t base = 0; base < n_src; base += 256u, ++step) {
          const uint32_t e = base + threadIdx.x;
          const bool valid = e < n_src;
          const uint32_t ei = valid ? e : (n_src - 1u);
          const uint32_t tri = mI != nullptr ? mI[ei] : ei;
          const uint32_t ti = tri;
          const float4 A0 = p.tri_a[2u * ti], A1 = p.tri_a[2u * ti + 1u];
          const float bz = p.tri_b[ti];
          bool keep = valid;
          if (bfam.usable)
            keep = valid && !tile_misses_triangle<false, false, SlackProduct, THIRD_BLOCK>(bfam, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z});
          const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
          uint32_t* const slot = bcnt + (step & 1u) * 4u;          // double-buffered: one barrier per step
          if (lane == 0u) slot[wave] = static_cast<uint32_t>(__builtin_popcountll(m));
          __syncthreads();
          const uint32_t c0 = slot[0], c1 = slot[1], c2 = slot[2], c3 = slot[3];
          const uint32_t before = (wave > 0u ? c0 : 0u) + (wave > 1u ? c1 : 0u) + (wave > 2u ? c2 : 0u);
          const uint32_t step_total = c0 + c1 + c2 + c3;
          if (total + step_total > Lb) { overflow = true; break; }   // block-uniform
          const uint32_t pos = total + before + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                                      __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
          if (keep) bI[pos] = tri;                                 // ascending order across waves and steps
          total += step_total;
        }
        __syncthreads();
        if (!overflow) { src_count = total; src_is_block_list = true; }
      }
    }
  }

  // classify triangles [from, n) until the list is full; returns the first unclassified index
  auto classify = [&](uint32_t from, auto with_forms) -> uint32_t {
    constexpr bool WF = decltype(with_forms)::value;                // forms only from the call before the sample loop
    uint32_t count = 0, base = from;
    while (base < src_count && count < L) {
      const uint32_t e = base + lane;
      const bool valid = e < src_count;
      const uint32_t ei = valid ? e : (src_count - 1u);
      const uint32_t tri = src_is_block_list ? bI[ei] : (mI != nullptr ? mI[ei] : ei);
      const uint32_t ti = tri;
      float4 A0, A1;
      float bz;
      A0 = p.tri_a[2u * ti]; A1 = p.tri_a[2u * ti + 1u];
      bz = p.tri_b[ti];
      bool keep = valid;
      float forms[18] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f,
                         0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};   // all-zero forms never reject
      if (fam.usable) {
        if constexpr (PRETEST && WF) {
          if (pretest) {
            // The rules on S' = det' - U' - V' as a call of their own in front of the forms: inside the forms call their sums
            // overflow the register budget (92 instead of 24 bytes of scratch, 4.47 instead of 3.75 ms at C4); the triangle's
            // values are laundered in between so that the two calls share no live ranges.  C4: 9.5 -> 8.8 candidates per
            // tile, 3.75 -> 3.68 ms.
            const bool miss3 = tile_misses_triangle<false, false, SlackProduct, true>(fam, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z});
            asm volatile("" : "+v"(A0.x), "+v"(A0.y), "+v"(A0.z), "+v"(A0.w), "+v"(A1.x), "+v"(A1.y), "+v"(A1.z), "+v"(A1.w), "+v"(bz));
            keep = valid && !miss3 && !tile_misses_triangle<true>(fam, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z}, forms);
          }
          else keep = valid && !tile_misses_triangle<false, false, SlackProduct, THIRD_WAVE>(fam, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z});
        } else {
          keep = valid && !tile_misses_triangle<false, false, SlackProduct, THIRD_WAVE>(fam, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z});
        }
      }
      const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
      if (count + static_cast<uint32_t>(__builtin_popcountll(m)) > L) break;   // does not fit: this step opens the next round
      const uint32_t pos = count + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                             __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
      if (keep) {                                                  // ascending order is preserved
        cA[2u * pos] = A0;
        cA[2u * pos + 1u] = A1;
        cB[pos] = bz;
        cI[pos] = static_cast<int>(tri);
        if constexpr (PRETEST && WF) {
          if (pretest) {
            auto pk = [&](float hi, float lo) {                        // two fp16 (the values are fp16-exact already) in one word
              const uint32_t h = __builtin_bit_cast(uint16_t, static_cast<_Float16>(hi)), l = __builtin_bit_cast(uint16_t, static_cast<_Float16>(lo));
              return __builtin_bit_cast(float, (h << 16) | l);
            };
            cP[4u * pos] = make_float4(forms[0], forms[1], forms[2], forms[3]);
            cP[4u * pos + 1u] = make_float4(forms[4], forms[5], forms[6], forms[7]);
            cP[4u * pos + 2u] = make_float4(forms[8], pk(forms[9], forms[10]), pk(forms[11], forms[12]), pk(forms[13], forms[14]));
            cP[4u * pos + 3u] = make_float4(pk(forms[15], forms[16]), pk(forms[17], 0.0f), 0.0f, 0.0f);
          }
        }
      }
      count += static_cast<uint32_t>(__builtin_popcountll(m));
      base += 64u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // this wave's ds_writes before its ds_reads
    __builtin_amdgcn_wave_barrier();
    list_count = count;
    if constexpr (STATS) { st_bin[0] += count; st_bin[1] += 1; }
    return base < src_count ? base : src_count;
  };

  // ONEPASS: the scene has no more triangles than the list holds (host-checked).  The tile's list -- a function of the
  // tile's ray family only, not of the samples -- comes from tile_lists_kernel: the wave gathers the records of the listed
  // triangles into its LDS slot; a tile with a certain winner needs no records at all.
  if constexpr (BIN && ONEPASS) {
    const size_t slot = (static_cast<size_t>(by) * gxb + bx) * 4u + wave;
    const uint32_t count = list_word & 0x3FFu;                         // bit 31: the tile has a certain winner, bits 10..19: its triangle
    sure_hit_tile = loaded_sure;
    sure_winner = (list_word >> 10) & 0x3FFu;
    if (p.tile_lists != nullptr) {
      const uint32_t* const saved = p.tile_lists + slot * (1u + L);
      for (uint32_t base = 0; base < (sure_hit_tile ? 0u : count); base += 64u) {
        const uint32_t e = base + lane;
        if (e < count) {
          const uint32_t tri = saved[1u + e];
          cA[2u * e] = p.tri_a[2u * tri];
          cA[2u * e + 1u] = p.tri_a[2u * tri + 1u];
          cB[e] = p.tri_b[tri];
          cI[e] = static_cast<int>(tri);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // this wave's ds_writes before its ds_reads
    __builtin_amdgcn_wave_barrier();
    list_count = count;
    if constexpr (STATS) { st_bin[0] += count; st_bin[1] += 1; }
    list_complete = true;
  }
  // Large scenes: the first classification also runs BEFORE any ray exists (few live registers, and
  // only here are the per-sample forms computed).  A tile whose candidates fit the list -- nearly
  // all of them -- never classifies again; one that overflows falls back to rounds inside the
  // sample loop (lists rebuilt per batch, without forms).
  bool forms_ready = false;
  if constexpr (PRETEST) {
    const uint32_t next0 = classify(0u, std::true_type{});
    list_complete = next0 >= src_count;
    forms_ready = pretest && list_complete;
  }
  if constexpr (RNG_LATE) load_rng();
  tl_mark(1);                                                      // family + classification done
  float4 sure_col = make_float4(0.0f, 0.0f, 0.0f, 0.0f);           // the winner's colour -- or, with the table, its p.samples-fold sum
  const bool sure_sums = BIN && ONEPASS && p.sure_table != nullptr;
  if constexpr (BIN && ONEPASS) { if (sure_hit_tile) sure_col = sure_sums ? p.sure_table[sure_winner] : p.tri_color[sure_winner]; }
  if constexpr (STATS && BIN && ONEPASS) {                         // tiles with a certain winner; the others by list length: 0, 1, 2, more
    if (lane == 0u && p.stats != nullptr)
      atomicAdd(p.stats + (sure_hit_tile ? 13 : list_count == 0u ? 11 : list_count == 1u ? 12 : list_count == 2u ? 14 : 15), 1ull);
  }
  const uint32_t iters = FUSE ? p.iters : 1u;
  float rx = 0.0f, ry = 0.0f, rz = 0.0f, rw = 0.0f;                // FUSE: the pixel's RenderBuffer value so far
  uint32_t cnt_first = 0u;
  for (uint32_t it = 0; it < iters; ++it) {                        // FUSE: the host loop's iterations, :246
  if constexpr (FUSE) { ax = 0.0f; ay = 0.0f; az = 0.0f; }          // accu, :133
  uint32_t traced_samples = p.samples;
  if constexpr (BIN && ONEPASS) {
    // The tile's winner is hit by every ray of its family: every sample's radiance is that triangle's colour
    // (Kernels.cuh:95-99), whatever the lens sample -- no ray, no test.  What the samples still do to the state is kept
    // exactly: the three draws of each lens sample (Random.cuh:15-16) and the additions of :137 in sample order.
    if (sure_hit_tile) {                                           // wave-uniform
      rtd::rng_discard(rng, 3u * p.samples);
      if (sure_sums) { ax = sure_col.x; ay = sure_col.y; az = sure_col.z; }      // (the same additions, done once per triangle)
      else for (uint32_t s = 0; s < p.samples; ++s) { ax += sure_col.x; ay += sure_col.y; az += sure_col.z; }
      if constexpr (STATS) st_pre += (p.samples + static_cast<uint32_t>(K) - 1u) / static_cast<uint32_t>(K);
      traced_samples = 0u;
    }
  }
  for (uint32_t s0 = 0; s0 < traced_samples; s0 += K) {            // :134, K samples per pass
    if (s0 == static_cast<uint32_t>(K)) tl_mark(2);                // first batch done (includes the wait for the RNG state)
    const uint32_t valid_k = (p.samples - s0 < static_cast<uint32_t>(K)) ? p.samples - s0 : static_cast<uint32_t>(K);
    V3 o[K], d[K];
    float best_t[K];
    int best_i[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (static_cast<uint32_t>(k) < valid_k) get_ray<FMA>(p, focal, rng, o[k], d[k]);   // :136
      else { o[k] = {0.0f, 0.0f, 0.0f}; d[k] = {0.0f, 0.0f, -1.0f}; }   // padding ray, result discarded (a constant: the pinhole ray need not stay live)
      best_t[k] = nearest ? FLT_MAX : -FLT_MAX;                    // :73
      best_i[k] = -1;
    }

    const unsigned long long lanes_in = __builtin_amdgcn_ballot_w64(inside);   // (wave constant: hoisted)
    float dox[K], doy[K];                                           // PRETEST: each ray's lens offset do = o - oc
    float dFx = 0.0f, dFy = 0.0f, dFz = 0.0f;                       // PRETEST: this lane's focal point minus the tile's box centre
    if constexpr (PRETEST) {
#pragma unroll
      for (int k = 0; k < K; ++k) { dox[k] = o[k].x - p.cam[9]; doy[k] = o[k].y - p.cam[10]; }
      dFx = focal.x - fam.fc[0]; dFy = focal.y - fam.fc[1]; dFz = focal.z - fam.fc[2];
    }
    if constexpr (BIN && ONEPASS) {
      for (uint32_t j = 0; j < list_count; ++j) {                  // ascending triangle order
        const float4 A0 = cA[2u * j], A1 = cA[2u * j + 1u];
        test_triangle<FMA, K, FILTER, STATS>(A0, A1, [&] { return cB[j]; }, cI[j], o, d, best_t, best_i,
                                             nearest, inside, valid_k, st_exit, st_skip);
      }
    } else if constexpr (BIN) {
      uint32_t base = 0;
      do {
        uint32_t next = src_count;
        if (!list_complete) {
          next = classify(base, std::false_type{});
          if (base == 0u && next >= src_count) list_complete = true;     // (!PRE: the first classification happens here)
        }
        // does ANY ray of the wave survive the per-sample forms of candidate j?  (wave-uniform)
        auto forms_alive = [&](uint32_t j) -> bool {
          // per-sample forms of this candidate at each ray's own lens origin: does ANY ray of the wave survive?
          const float4 f0 = cP[4u * j], f1 = cP[4u * j + 1u], q2 = cP[4u * j + 2u], q3 = cP[4u * j + 3u];
          const float f2 = q2.x;
          // gradients: fp16 pairs read in place by v_fma_mix_f32 (op_sel picks the half): no unpack instructions
          typedef _Float16 h2 __attribute__((ext_vector_type(2)));
          const h2 w0 = __builtin_bit_cast(h2, q2.y), w1 = __builtin_bit_cast(h2, q2.z), w2 = __builtin_bit_cast(h2, q2.w),
                   w3 = __builtin_bit_cast(h2, q3.x), w4 = __builtin_bit_cast(h2, q3.y);      // .y = high half
          // per lane and candidate: constant term + gradient . (this lane's focal point - box centre)
          const float b1 = __builtin_fmaf(static_cast<float>(w1.y), dFz, __builtin_fmaf(static_cast<float>(w0.x), dFy, __builtin_fmaf(static_cast<float>(w0.y), dFx, f0.x)));
          const float b2 = __builtin_fmaf(static_cast<float>(w2.x), dFz, __builtin_fmaf(static_cast<float>(w2.y), dFy, __builtin_fmaf(static_cast<float>(w1.x), dFx, f0.w)));
          const float b3 = __builtin_fmaf(static_cast<float>(w4.y), dFz, __builtin_fmaf(static_cast<float>(w3.x), dFy, __builtin_fmaf(static_cast<float>(w3.y), dFx, f1.z)));
          // a ray is skipped when its smallest form is negative; the candidate when that holds for every ray of the wave:
          // per sample one compare, the lane masks combined on the scalar unit (a NaN form never skips: NaN < 0 is false)
          unsigned long long all_neg = ~0ull;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const float F1 = __builtin_fmaf(f0.z, doy[k], __builtin_fmaf(f0.y, dox[k], b1));
            const float F2 = __builtin_fmaf(f1.y, doy[k], __builtin_fmaf(f1.x, dox[k], b2));
            const float F3 = __builtin_fmaf(f2, doy[k], __builtin_fmaf(f1.w, dox[k], b3));
            const float worst = __builtin_fminf(__builtin_fminf(F1, F2), F3);
            all_neg &= __builtin_amdgcn_ballot_w64(worst < 0.0f) | ((static_cast<uint32_t>(k) < valid_k) ? 0ull : ~0ull);
          }
          const bool alive = (~all_neg & lanes_in) != 0ull;
          if constexpr (STATS) { if (!alive) st_pre += 1; }
          return alive;
        };
        auto run_tests = [&](uint32_t j) {
          const float4 A0 = cA[2u * j], A1 = cA[2u * j + 1u];
          test_triangle<FMA, K, FILTER, STATS>(A0, A1, [&] { return cB[j]; }, cI[j], o, d, best_t, best_i,
                                               nearest, inside, valid_k, st_exit, st_skip);
        };
        if (forms_ready) {
          // Two candidates per trip: both records are requested before either form is evaluated, so that a wave waits for
          // LDS once per pair -- the loop is latency-bound at 4 waves per SIMD: C4 4.64 -> 4.15 ms.  (Written out: the same
          // loop as a generic group of N with a flag array measured 4.39 / 4.42 / 4.43 ms for N = 2 / 3 / 4.)  The exact
          // tests stay in ascending order.
          uint32_t j = 0;
          for (; j + 1u < list_count; j += 2u) {
            const bool a0 = forms_alive(j), a1 = forms_alive(j + 1u);
            if (a0) run_tests(j);
            if (a1) run_tests(j + 1u);
          }
          if (j < list_count && forms_alive(j)) run_tests(j);
        } else {
          for (uint32_t j = 0; j < list_count; ++j) run_tests(j);   // ascending triangle order
        }
        base = next;
        if (!list_complete) __builtin_amdgcn_wave_barrier();       // list is rewritten by the next round
      } while (!list_complete && base < src_count);
    } else {
      for (uint32_t c0 = 0; c0 < n; c0 += p.chunk) {
        const uint32_t cn = (n - c0 < p.chunk) ? n - c0 : p.chunk;
        if (!single_chunk) {
          __syncthreads();                                         // everyone done with the previous chunk
          for (uint32_t i = threadIdx.x; i < 2u * cn; i += 256u) sA[i] = p.tri_a[2u * c0 + i];
          for (uint32_t i = threadIdx.x; i < cn; i += 256u) sB[i] = p.tri_b[c0 + i];
          __syncthreads();
        }
        for (uint32_t j = 0; j < cn; ++j) {                        // :75, ascending order
          const float4 A0 = sA[2u * j], A1 = sA[2u * j + 1u];
          test_triangle<FMA, K, FILTER, STATS>(A0, A1, [&] { return sB[j]; }, static_cast<int>(c0 + j), o, d,
                                               best_t, best_i, nearest, inside, valid_k, st_exit, st_skip);
        }
      }
    }

    // spheres continue the same farthest-hit scan, then shade in sample order (:95-104, :137)
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (static_cast<uint32_t>(k) < valid_k) {
        float dist = best_t[k];
        int win = best_i[k];
        for (uint32_t si = 0; si < p.n_spheres; ++si) {
          float t = 0.0f;
          if (hit_sphere<FMA>(o[k], d[k], p.spheres[si], t) && (nearest ? (t > 0.0f && t < dist) : dist < t)) {
            dist = t;
            win = static_cast<int>(n + si);
          }
        }
        float r, g, b;
        if (win >= 0) {
          if (win < static_cast<int>(n)) {
            if (p.tri_n != nullptr) {
              // build-defined smooth shading: vertex normals interpolated at the winner's barycentrics;
              // u, v are recomputed from the winner's record (same arithmetic as the scan: same bits)
              const float4 A0 = p.tri_a[2 * win], A1 = p.tri_a[2 * win + 1];
              float t = 0.0f, u = 0.0f, v = 0.0f;
              int stage;
              (void)hit_triangle_exact<FMA>(o[k], d[k], {A1.z, A1.w, p.tri_b[win]}, {A0.w, A1.x, A1.y},
                                            {A0.x, A0.y, A0.z}, RT_EPS, t, u, v, stage);
              const float4 n0 = p.tri_n[3 * win], n1 = p.tri_n[3 * win + 1], n2 = p.tri_n[3 * win + 2];
              const float w = (1.0f - u) - v;
              V3 m;
              if constexpr (FMA) {
                m.x = __builtin_fmaf(v, n2.x, __builtin_fmaf(u, n1.x, w * n0.x));
                m.y = __builtin_fmaf(v, n2.y, __builtin_fmaf(u, n1.y, w * n0.y));
                m.z = __builtin_fmaf(v, n2.z, __builtin_fmaf(u, n1.z, w * n0.z));
              } else {
                m.x = (w * n0.x + u * n1.x) + v * n2.x;
                m.y = (w * n0.y + u * n1.y) + v * n2.y;
                m.z = (w * n0.z + u * n1.z) + v * n2.z;
              }
              const V3 nn = M::normalize(m);
              r = rtd::absf(nn.x); g = rtd::absf(nn.y); b = rtd::absf(nn.z);
            } else {
              const float4 col = p.tri_color[win];
              r = col.x; g = col.y; b = col.z;
            }
          } else {
            const float4 sph = p.spheres[win - static_cast<int>(n)];
            const V3 hp = {M::madd1(d[k].x, dist, o[k].x), M::madd1(d[k].y, dist, o[k].y),
                           M::madd1(d[k].z, dist, o[k].z)};                 // Ray::point, Ray.cuh:41-44
            const V3 nn = M::normalize(rtd::sub(hp, {sph.x, sph.y, sph.z}));
            r = rtd::absf(nn.x); g = rtd::absf(nn.y); b = rtd::absf(nn.z);
          }
        } else {                                                    // :103, background (0.15,0.11,0.13)
          if constexpr (FMA) {
            r = __builtin_fmaf(d[k].x, 0.2f, 0.15f * 0.8f);
            g = __builtin_fmaf(d[k].y, 0.2f, 0.11f * 0.8f);
            b = __builtin_fmaf(d[k].z, 0.2f, 0.13f * 0.8f);
          } else {
            r = 0.15f * 0.8f + d[k].x * 0.2f;
            g = 0.11f * 0.8f + d[k].y * 0.2f;
            b = 0.13f * 0.8f + d[k].z * 0.2f;
          }
        }
        ax += r; ay += g; az += b;                                  // :137
      }
    }
  }

  if constexpr (FUSE) {                                            // end of iteration `it`: :140-143
    if (it == 0u && !(p.flags & TRACE_ZERO_ACC) && inside) {        // wave-uniform but for `inside`
      const float4 r0 = p.render[pix];
      rx = r0.x; ry = r0.y; rz = r0.z; rw = r0.w;
      cnt_first = p.counts[pix];
    }
    rx += ax; ry += ay; rz += az;
  }
  }                                                                // iterations
  tl_mark(3);                                                      // all samples done
  // The epilogue forms its addresses afresh from the pixel index: kept live from the prologue's RNG loads they are
  // four 64-bit values that no longer fit the 96-VGPR budget (spilled: 130 MB of scratch traffic per C3 launch).
  uint32_t pix_lo = static_cast<uint32_t>(pix);                     // (a band has < 2^32 pixels)
  asm volatile("" : "+v"(pix_lo));
  const size_t pix_e = pix_lo;
  if (inside) {
    // Accumulators are read here, not prefetched at kernel start: five registers held across
    // the whole kernel cost more (spills at the 96-VGPR budget of 5 waves/SIMD) than the exposed
    // read latency of a finished wave (measured C3 169.4 -> 164.9 us, progressive launches equal).
    float4 acc_in = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t cnt_in = 0u;
    if constexpr (!FUSE) {
      if (!(p.flags & TRACE_ZERO_ACC)) {                            // wave-uniform
        acc_in = p.render[pix_e];
        cnt_in = p.counts[pix_e];
      }
    }
    const uint32_t cnt = FUSE ? cnt_first + iters * p.samples : cnt_in + p.samples;   // :140
    p.counts[pix_e] = cnt;
    float4 acc = acc_in;
    if constexpr (FUSE) {
      acc = make_float4(rx, ry, rz, rw);
    } else {
      acc.x += ax; acc.y += ay; acc.z += az;                        // :141-143, alpha untouched (:144)
    }
    p.render[pix_e] = acc;
    if (p.flags & TRACE_EMIT_IMAGE) {                               // fused rt::ConverterKernel, :164-168
      const float c = static_cast<float>(cnt);
      uint32_t bgra;
      // a certain-winner tile on cleared accumulators: the table holds the word every pixel of it gets (wave-uniform branch)
      if (!FUSE && sure_sums && sure_hit_tile && (p.flags & TRACE_ZERO_ACC)) bgra = __builtin_bit_cast(uint32_t, sure_col.w);
      else bgra = rtd::pack_color(255.0f * (acc.x / c), 255.0f * (acc.y / c), 255.0f * (acc.z / c));
      p.image[pix_e] = bgra;
      // update hand-off: the same value straight into the caller-visible pinned host image (posted
      // PCIe writes, one 256-byte row segment per wave store) -- no device-to-host copy afterwards
      if (p.image_host != nullptr) p.image_host[pix_e] = bgra;
    }
    p.rng[0 * static_cast<size_t>(p.npix) + pix_e] = rng.d;           // :146
    p.rng[1 * static_cast<size_t>(p.npix) + pix_e] = rng.v0;
    p.rng[2 * static_cast<size_t>(p.npix) + pix_e] = rng.v1;
    p.rng[3 * static_cast<size_t>(p.npix) + pix_e] = rng.v2;
    p.rng[4 * static_cast<size_t>(p.npix) + pix_e] = rng.v3;
    p.rng[5 * static_cast<size_t>(p.npix) + pix_e] = rng.v4;
  }
  tl_mark(4);                                                      // stores issued
  if constexpr (STATS) {
    if (lane == 0 && p.stats != nullptr) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        atomicAdd(p.stats + e, st_exit[e]);
        atomicAdd(p.stats + 4 + e, st_skip[e]);
      }
      atomicAdd(p.stats + 8, st_bin[0]);
      atomicAdd(p.stats + 9, st_bin[1]);
      atomicAdd(p.stats + 10, st_pre);
    }
  }
}

}  // namespace rtk
