// rt_dense.hpp -- dense scenes (4 096 ... 50 000 triangles, the per-sample forms): the per-wave candidate lists as a structure of
// their own in HBM, built by wave_lists_kernel, read by trace_kernel<.., PRE, HBM> (rt_trace.hpp).
//
// rt::Radiance tests every triangle for every ray (RayTracer/Kernels.cuh:75-92).  Rounds 1-3 classified inside the trace kernel:
// every wave of EVERY launch walked macro list -> block list -> its own list and kept the survivors' records and per-sample
// forms in LDS.  The lists depend on camera, scene and frame, not on the samples -- like the macro lists and the small scenes'
// tile lists they are an acceleration structure -- so wave_lists_kernel runs that three-level classification once per key (the
// same focal_bounds, make_family, block pre-cull and tile_misses_triangle calls, the same forms) and writes per 8x8 tile a
// header and the survivors, ascending, 64 bytes each; the trace kernel's HBM instantiation contains no classification and no
// barrier: a wave copies its tile's records into its LDS slot and runs the candidate loop of round 3.  Accumulating launches
// of a Trace (and, with list reuse across Traces, every launch of an unchanged view) skip the build.
// A tile whose survivors exceed the list's capacity is marked and falls back to the exact tests over its macro tile's list
// (correct by construction: a superset, ascending); tests force that path with a tiny capacity (rt_options.bin_list).
// (Measured and dropped, HISTORY.md round 4: the candidate loop reading the records as wave-uniform scalar loads -- s_load
// into SGPRs that the VALU takes as operands, no LDS at all -- is bit-identical and 13 % slower: 16 waves x 8.8 candidates x
// 128 bytes cycle through a 16 KiB scalar cache that two CUs share.)
//
// Layout of p.wave_lists, per tile slot of the (half-)launch grid, (1 + wave_cap) records of 16 dwords (64 bytes):
//   header     [0] count (0xFFFFFFFF = overflow)  [1..3] fc = centre of the tile's focal box (the forms' dF = F - fc)
//   candidate  [0..13] the forms exactly as the LDS slot holds them (9 floats, 5 words of fp16 gradient pairs)  [14] triangle index
// (the triangle's 36-byte record and its colour are gathered from the scene's own tables -- 640 KB at C4, L2-resident -- when
//  the wave loads its list: what travels through HBM per build is the forms)
#pragma once
#include "rt_trace.hpp"

namespace rtk {

// ------------------------------------------------------------------------------------
// Can ANY lens sample of ANY pixel of the tile pass all three per-sample forms at once?
//
// The trace loop skips a candidate for a ray when min(F1, F2, F3) < 0, F_i = c_i + g_i . dF + n_i . do with the lane's
// focal offset dF (|dF_k| <= frad_k over the tile) and the sample's lens offset do (|do| <= R: the lens is a disk).  The
// tile-level rules bound each quantity on its own, so a triangle whose image in lens space -- the intersection of the three
// half-planes -- lies outside the lens while each half-plane alone still cuts it is kept, evaluated for every sample batch
// and rejected every time.  With every form taken at its largest over a focal (sub-)box, C_i = c_i + g_i . centre + sum_k |g_ik| r_k, a sample
// that passes exists only if the disk meets {x : C_i + n_i . x >= 0 for all i}; by Helly's theorem in the plane that fails
// iff it fails for the disk and TWO of the half-planes, i.e. iff for some pair the wedge H_i /\ H_j is farther than R from the
// lens centre: the closest point of a wedge is the centre itself, the foot of the perpendicular on one edge (if the other
// constraint holds there) or the apex.  A candidate this test drops is one the forms would skip for every sample of every
// lane: no result changes (the forms' own guarantee, CLASSIFICATION.md), the lists get shorter.  R and the comparison carry
// 1e-3 relative slack (the roundings of this evaluation and of the loop's fp32 forms are ~1e-6); any NaN keeps the candidate.
// forms[]: as tile_misses_triangle<FORMS> leaves them (scaled per form by a power of two: the geometry is scale-free).
// ------------------------------------------------------------------------------------
#ifndef RT_LENS_SPLIT
#define RT_LENS_SPLIT 2       // the focal box is tested in RT_LENS_SPLIT^2 sub-boxes (its two widest axes split): any partition of
                              // the box covers every focal point of the tile, and a smaller box fattens the wedges less
#endif
// one (sub-)box: form i at its largest over it is C[i]; true = the disk of radius^2 R2 may meet all three half-planes.
// Division-free: distances are compared as (numerator)^2 > R^2 (denominator)^2.  (The pair constants |n_i|^2, n_i . n_j, det^2
// are recomputed per sub-box; hoisting them buys nothing.  What did push the build kernel into scratch -- 68 bytes per lane,
// 340 MB of HBM traffic per C4 launch -- was forms[] indexed by a run-time axis in lens_can_pass_forms: see the selects there.)
__device__ __forceinline__ bool lens_meets_wedges(const float (&C)[3], const float (&nx)[3], const float (&ny)[3], float R2) {
  bool surely_outside = false;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int i = a, j = (a + 1) % 3;
    const float li = nx[i] * nx[i] + ny[i] * ny[i], lj = nx[j] * nx[j] + ny[j] * ny[j], dij = nx[i] * nx[j] + ny[i] * ny[j];
    const float det = nx[i] * ny[j] - ny[i] * nx[j];
    const float det2 = det * det;
    const bool par = !(det2 > 1e-12f * (li * lj));                  // edges too close to parallel (or NaN): no verdict from this pair
    const float Ci = C[i], Cj = C[j];
    const bool centre_inside = !(Ci < 0.0f) && !(Cj < 0.0f);       // (or NaN: keep)
    // the closest point of the wedge H_i /\ H_j to the lens centre is the foot of the perpendicular on edge i (if C_i < 0 and the
    // foot satisfies j: C_j |n_i|^2 - C_i n_i.n_j >= 0), the foot on edge j, or the apex; the wedge is outside the disk when every
    // one of those that belongs to it is: C_i^2 > R^2 |n_i|^2, |C_i n_j - C_j n_i|^2 > R^2 det^2
    const bool foot_i = (Ci < 0.0f) && (li > 0.0f) && (Cj * li - Ci * dij >= 0.0f);
    const bool foot_j = (Cj < 0.0f) && (lj > 0.0f) && (Ci * lj - Cj * dij >= 0.0f);
    const bool far_i = Ci * Ci > R2 * li, far_j = Cj * Cj > R2 * lj;
    const float wx = Ci * nx[j] - Cj * nx[i], wy = Ci * ny[j] - Cj * ny[i];
    const float apex2 = wx * wx + wy * wy;
    const bool far_apex = (apex2 > R2 * det2) && (apex2 <= 3.0e38f);
    const bool outside = !centre_inside && !par && far_apex && (!foot_i || far_i) && (!foot_j || far_j);
    surely_outside = surely_outside || outside;
  }
  return !surely_outside;
}

__device__ __forceinline__ bool lens_can_pass_forms(const float (&forms)[18], const float (&frad)[3], float lens_radius) {
  float nx[3], ny[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { nx[i] = forms[3 * i + 1]; ny[i] = forms[3 * i + 2]; }
  const float R = lens_radius * 1.001f + 1e-30f, R2 = R * R * 1.001f;
  // the two widest axes of the focal box are split, the third keeps its radius (wave-uniform choice)
  // (selected with compares, not indexed: a register array indexed by a run-time value lives in scratch)
  const bool n0 = frad[0] <= frad[1] && frad[0] <= frad[2], n1 = !n0 && frad[1] <= frad[2];      // the narrow axis: 0, 1, else 2
  constexpr int S = RT_LENS_SPLIT;
  const float ra = n0 ? frad[1] : frad[0], rb = (n0 || n1) ? frad[2] : frad[1], rn = n0 ? frad[0] : n1 ? frad[1] : frad[2];
  const float ha = ra / S, hb = rb / S;                              // sub-box radii (1.001 below covers the roundings of the centres)
  float c0[3], ga[3], gb[3], spread[3];                              // per form: constant, gradient along the split axes, radius term of a sub-box
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float g0 = forms[9 + 3 * i], g1 = forms[10 + 3 * i], g2 = forms[11 + 3 * i];
    ga[i] = n0 ? g1 : g0; gb[i] = (n0 || n1) ? g2 : g1;
    const float gn = n0 ? g0 : n1 ? g1 : g2;
    c0[i] = forms[3 * i];
    spread[i] = (__builtin_fabsf(ga[i]) * ha + __builtin_fabsf(gb[i]) * hb + __builtin_fabsf(gn) * rn) * 1.001f;
  }
  bool any = false;
  for (int qa = 0; qa < S; ++qa) {
    for (int qb = 0; qb < S; ++qb) {
      const float ca = (2.0f * qa + 1.0f - S) * ha, cb = (2.0f * qb + 1.0f - S) * hb;       // sub-box centre relative to the box centre
      float C[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) C[i] = (c0[i] + (ga[i] * ca + gb[i] * cb)) + spread[i];
      any = any || lens_meets_wedges(C, nx, ny, R2);
    }
  }
  return any;
}

// ------------------------------------------------------------------------------------
// wave_lists_kernel: one WAVE per block of 32 x 8 pixels (its four 8 x 8 tiles), four blocks per workgroup; no barrier.
// grid = ceil(blocks / 4) workgroups of 256 threads, dynamic LDS = 4 waves x block_list x 4 bytes.
//
// (Until r04_g one wave per TILE, four waves sharing the block list through LDS: every wave paid a whole pass of the forms
// -- ~850 instructions on 64 lanes -- for the ~12 entries of its block list, 1 500 instructions per tile, 8 % of a C4
// step's instructions and most of what a rebuilt step costs over one with the lists kept.)  Here
//   1. the wave visits its four tiles in turn, lane = pixel, for their focal boxes (focal_bounds: the same calls, the same
//      values as the tile's own wave computed), the block's box is their union;
//   2. block level, lane = entry of the macro tile's list: the union family, survivors ascending into this wave's LDS list;
//   3. tile level, lane = tile * 16 + entry: sixteen entries of the block list per pass against each of the four tiles' OWN
//      families (per lane: selected from the four boxes), third-edge rules + forms + the joint lens rule, compaction per
//      16-lane group, survivors straight into the tile's records.
// The verdict functions, their inputs and the order of the survivors are those of the one-wave-per-tile build: same lists.
// ------------------------------------------------------------------------------------
template <bool FMA>
__global__ __launch_bounds__(256, 4) void wave_lists_kernel(const TraceParams p) {
  extern __shared__ float4 s_mem[];
  // The build runs in front of its half's trace kernel and beside the OTHER half's, whose long-lived waves would win every
  // arbitration against it: at the highest wave priority the 0.3 ms chain of dependent loads and classifications shortens
  // (C4 3.24 -> 3.10 ms per step; the trace waves lose what the build gains, the step's critical path is what shrinks).
  __builtin_amdgcn_s_setprio(3);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t gxb = (p.W + 31u) / 32u, gyb = (p.rows + 7u) / 8u;
  const uint32_t blk = blockIdx.x * 4u + wave;
  if (blk >= gxb * gyb) return;                                      // wave-uniform; no barrier below
  const uint32_t bx = blk % gxb, by = blk / gxb;

  // ---- 1. the four tiles' focal boxes (wave-uniform values) and their union
  FocalBounds wb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const uint32_t px = bx * 32u + static_cast<uint32_t>(t) * 8u + (lane & 7u), ly = by * 8u + (lane >> 3);
    const bool inside = px < p.W && ly < p.rows;
    V3 po, pd;
    pinhole<FMA>(p, inside ? px : 0u, p.row0 + (inside ? ly : 0u), po, pd);
    wb[t] = focal_bounds(p, focal_point<FMA>(p, pd), inside);
  }
  FocalBounds bb;
  bb.ok = wb[0].ok && wb[1].ok && wb[2].ok && wb[3].ok;
  bb.any = wb[0].any || wb[1].any || wb[2].any || wb[3].any;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    bb.lo[i] = fminf(fminf(fminf(fminf(FLT_MAX, wb[0].lo[i]), wb[1].lo[i]), wb[2].lo[i]), wb[3].lo[i]);       // (block_focal_union's order)
    bb.hi[i] = fmaxf(fmaxf(fmaxf(fmaxf(-FLT_MAX, wb[0].hi[i]), wb[1].hi[i]), wb[2].hi[i]), wb[3].hi[i]);
  }

  const uint32_t Lb = p.block_list;
  uint32_t* const bI = reinterpret_cast<uint32_t*>(s_mem) + static_cast<size_t>(wave) * Lb;
  uint32_t src_count = p.n_tris;
  bool src_is_block_list = false;
  const uint32_t* mI = nullptr;
  if (p.macro_lists != nullptr) {
    const uint32_t mt = (by * 8u / p.macro_h) * p.macro_nx + (bx * 32u / p.macro_w);
    const uint32_t* const ml = p.macro_lists + static_cast<size_t>(mt) * (p.macro_cap + 1u);
    const uint32_t mc = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ml[0])));
    if (mc != 0xFFFFFFFFu) { mI = ml + 1; src_count = mc; }
  }
  const uint32_t n_src = src_count;

  // ---- 2. block level, lane = entry of the macro tile's list
  if (Lb != 0u) {
    const TileFamily bfam = make_family(p, bb);
    uint32_t total = 0;
    bool overflow = false;
    for (uint32_t base = 0; base < n_src; base += 64u) {
      const uint32_t e = base + lane;
      const bool valid = e < n_src;
      const uint32_t ei = valid ? e : (n_src - 1u);
      const uint32_t tri = mI != nullptr ? mI[ei] : ei;
      const float4 A0 = p.tri_a[2u * tri], A1 = p.tri_a[2u * tri + 1u];
      const float bz = p.tri_b[tri];
      bool keep = valid;
      if (bfam.usable)
        keep = valid && !tile_misses_triangle<false, false, SlackProduct, true>(bfam, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z});
      const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
      const uint32_t kept = static_cast<uint32_t>(__builtin_popcountll(m));
      if (total + kept > Lb) { overflow = true; break; }             // wave-uniform: the tiles walk the macro tile's list instead
      const uint32_t pos = total + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
      if (keep) bI[pos] = tri;                                       // ascending
      total += kept;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");            // this wave's ds_writes before its ds_reads
    __builtin_amdgcn_wave_barrier();
    if (!overflow) { src_count = total; src_is_block_list = true; }
  }

  // ---- 3. tile level, lane = tile * 16 + entry
  const uint32_t t4 = lane >> 4, c = lane & 15u;
  FocalBounds mine = wb[0];                                          // selected with compares, not indexed (scratch otherwise)
#pragma unroll
  for (int t = 1; t < 4; ++t) {
    const bool is = t4 == static_cast<uint32_t>(t);
#pragma unroll
    for (int i = 0; i < 3; ++i) { mine.lo[i] = is ? wb[t].lo[i] : mine.lo[i]; mine.hi[i] = is ? wb[t].hi[i] : mine.hi[i]; }
    mine.ok = is ? wb[t].ok : mine.ok;
    mine.any = is ? wb[t].any : mine.any;
  }
  const TileFamily fam = make_family(p, mine);
  const size_t slot = (static_cast<size_t>(by) * gxb + bx) * 4u + t4;
  uint32_t* const out = p.wave_lists + slot * (1u + p.wave_cap) * kWaveRec;
  const uint32_t gsh = lane & 48u;                                   // first lane of this tile's 16-lane group
  uint32_t count = 0;                                                // (the same in the 16 lanes of a group)
  bool overflow = false;
  for (uint32_t base = 0; base < src_count; base += 16u) {            // (wave-uniform trip count)
    const uint32_t e = base + c;
    const bool valid = e < src_count && !overflow;
    const uint32_t ei = e < src_count ? e : (src_count - 1u);
    const uint32_t tri = src_is_block_list ? bI[ei] : (mI != nullptr ? mI[ei] : ei);
    const float4 A0 = p.tri_a[2u * tri], A1 = p.tri_a[2u * tri + 1u];
    const float bz = p.tri_b[tri];
    bool keep = valid;
    float forms[18] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f,
                       0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};     // all-zero forms never reject
    if (fam.usable) {
      // (one call: no rays are alive here, so the S rules fit beside the forms -- inside the classifying trace kernel of round 3
      //  they were a call of their own in front of it, for the registers; the verdict is the same conjunction either way)
      const bool miss = tile_misses_triangle<true, false, SlackProduct, true>(fam, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z}, forms);
      keep = valid && !miss;
#ifndef RT_NO_LENS_JOINT
      keep = keep && lens_can_pass_forms(forms, fam.frad, fam.A);
#endif
    }
    const uint32_t gm = static_cast<uint32_t>(__builtin_amdgcn_ballot_w64(keep) >> gsh) & 0xFFFFu;
    const uint32_t kept = static_cast<uint32_t>(__builtin_popcount(gm));
    if (count + kept > p.wave_cap) { overflow = true; keep = false; }   // this tile only: its header says so, its records are void
    const uint32_t pos = count + static_cast<uint32_t>(__builtin_popcount(gm & ((1u << c) - 1u)));
    if (keep) {                                                      // ascending order is preserved
      auto pk = [&](float hi, float lo) {                            // two fp16 (the values are fp16-exact already) in one word
        const uint32_t h = __builtin_bit_cast(uint16_t, static_cast<_Float16>(hi)), l = __builtin_bit_cast(uint16_t, static_cast<_Float16>(lo));
        return __builtin_bit_cast(float, (h << 16) | l);
      };
      float4* const r = reinterpret_cast<float4*>(out + (1u + pos) * kWaveRec);
      r[0] = make_float4(forms[0], forms[1], forms[2], forms[3]);
      r[1] = make_float4(forms[4], forms[5], forms[6], forms[7]);
      r[2] = make_float4(forms[8], pk(forms[9], forms[10]), pk(forms[11], forms[12]), pk(forms[13], forms[14]));
      r[3] = make_float4(pk(forms[15], forms[16]), pk(forms[17], 0.0f), __builtin_bit_cast(float, tri), 0.0f);
    }
    if (!overflow) count += kept;
  }
  if (c == 0u) {
    reinterpret_cast<float4*>(out)[0] = make_float4(__builtin_bit_cast(float, overflow ? kWaveOverflow : count), fam.fc[0], fam.fc[1], fam.fc[2]);
  }
}

}  // namespace rtk
