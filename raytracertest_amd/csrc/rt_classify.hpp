// rt_classify.hpp -- the conservative classification of a triangle against the ray family of a region of the frame
// (a wave tile, a block, a macro tile, a region of the small scenes' list builder): TileFamily, the focal bounds it is made
// from, tile_misses_triangle (drop rules, certainly-hit proof, q bounds, per-sample forms) and pair_farther.  Claims,
// rounding budget and evidence: CLASSIFICATION.md.  Included through rt_trace.hpp.
#pragma once
#include "rt_rays.hpp"

namespace rtk {

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


}  // namespace rtk
