// rt_dense.hpp -- dense scenes (4 096 ... 50 000 triangles, the per-sample forms): the per-wave candidate lists as a structure of
// their own in HBM, built by wave_lists_kernel and read by dense_trace_kernel through the SCALAR cache.
//
// rt::Radiance tests every triangle for every ray (RayTracer/Kernels.cuh:75-92).  Rounds 1-3 classified inside the trace kernel:
// every wave of every launch walked macro list -> block list -> its own list, kept the survivors' records and per-sample forms
// in LDS (104 bytes per candidate, 40 KiB per block) and read them back -- wave-uniform values -- through ds_read_b128 into
// VGPRs for every sample batch: 25 vector registers of operands that are the same in all 64 lanes, an LDS round trip per
// candidate in a loop that is latency-bound at 4 waves per SIMD, 24 bytes of scratch per lane.  Here
//   * wave_lists_kernel runs that three-level classification ONCE per (camera, scene, frame) -- the same focal_bounds,
//     make_family, block pre-cull and tile_misses_triangle calls, the same forms -- and writes per 8x8 tile a header and the
//     survivors, ascending, 128 bytes each; like the macro lists the result is keyed and kept: accumulating launches of a
//     Trace (and, with list reuse across Traces, every launch of an unchanged view) skip it;
//   * dense_trace_kernel contains no classification, no LDS and no barrier: a candidate's forms and record are wave-uniform
//     loads from read-only memory, i.e. s_load_dwordx16 through the scalar cache into SGPRs, which VALU instructions take as
//     operands directly.  The exact tests (test_triangle) and everything a ray computes are the code of rt_trace.hpp.
// A tile whose survivors exceed the list's capacity is marked and falls back to the exact tests over its macro tile's list
// (correct by construction: a superset, ascending); tests force that path with a tiny capacity (rt_options.bin_list).
//
// Layout of p.wave_lists, per tile slot of the (half-)launch grid, (1 + wave_cap) records of 32 dwords:
//   header     [0] count (0xFFFFFFFF = overflow)  [1..3] fc = centre of the tile's focal box (the forms' dF = F - fc)
//   candidate  [0..13] the forms as the LDS path stored them (9 floats, 5 words of fp16 gradient pairs)  [14] triangle index
//              [16..23] (e2.xyz, e1.x), (e1.yz, v0.xy)  [24] v0.z
#pragma once
#include "rt_trace.hpp"

namespace rtk {

constexpr uint32_t kWaveRec = 32u;                 // dwords per record
constexpr uint32_t kWaveOverflow = 0xFFFFFFFFu;

#ifndef RT_DENSE_WAVES
#define RT_DENSE_WAVES 4      // __launch_bounds__ 2nd argument of dense_trace_kernel
#endif

// ------------------------------------------------------------------------------------
// wave_lists_kernel: grid = (ceil(W/32), ceil(rows/8)), 256 threads, dynamic LDS = block_list * 4 + 160 bytes.
// ------------------------------------------------------------------------------------
template <bool FMA>
__global__ __launch_bounds__(256) void wave_lists_kernel(const TraceParams p) {
  extern __shared__ float4 s_mem[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t px = blockIdx.x * 32u + wave * 8u + (lane & 7u), ly = blockIdx.y * 8u + (lane >> 3);
  const bool inside = px < p.W && ly < p.rows;
  V3 po, pd;
  pinhole<FMA>(p, inside ? px : 0u, p.row0 + (inside ? ly : 0u), po, pd);
  const V3 focal = focal_point<FMA>(p, pd);

  const uint32_t Lb = p.block_list;
  uint32_t* const bI = reinterpret_cast<uint32_t*>(s_mem);
  uint32_t* const bcnt = bI + Lb;
  float* const bbox = reinterpret_cast<float*>(bcnt + 8);
  uint32_t src_count = p.n_tris;
  bool src_is_block_list = false;
  const uint32_t* mI = nullptr;
  if (p.macro_lists != nullptr) {
    const uint32_t mt = (blockIdx.y * 8u / p.macro_h) * p.macro_nx + (blockIdx.x * 32u / p.macro_w);
    const uint32_t* const ml = p.macro_lists + static_cast<size_t>(mt) * (p.macro_cap + 1u);
    const uint32_t mc = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ml[0])));
    if (mc != 0xFFFFFFFFu) { mI = ml + 1; src_count = mc; }
  }
  const uint32_t n_src = src_count;
  const FocalBounds wb = focal_bounds(p, focal, inside);
  const TileFamily fam = make_family(p, wb);
  if (Lb != 0u) {                                                    // block level: the union of the four tiles' families
    const FocalBounds bb = block_focal_union(wb, bbox, wave, lane);
    const TileFamily bfam = make_family(p, bb);
    uint32_t total = 0, step = 0;
    bool overflow = false;
    for (uint32_t base = 0; base < n_src; base += 256u, ++step) {
      const uint32_t e = base + threadIdx.x;
      const bool valid = e < n_src;
      const uint32_t ei = valid ? e : (n_src - 1u);
      const uint32_t tri = mI != nullptr ? mI[ei] : ei;
      const float4 A0 = p.tri_a[2u * tri], A1 = p.tri_a[2u * tri + 1u];
      const float bz = p.tri_b[tri];
      bool keep = valid;
      if (bfam.usable)
        keep = valid && !tile_misses_triangle<false, false, SlackProduct, true>(bfam, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z});
      const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
      uint32_t* const slot = bcnt + (step & 1u) * 4u;                // double-buffered: one barrier per step
      if (lane == 0u) slot[wave] = static_cast<uint32_t>(__builtin_popcountll(m));
      __syncthreads();
      const uint32_t c0 = slot[0], c1 = slot[1], c2 = slot[2], c3 = slot[3];
      const uint32_t before = (wave > 0u ? c0 : 0u) + (wave > 1u ? c1 : 0u) + (wave > 2u ? c2 : 0u);
      const uint32_t step_total = c0 + c1 + c2 + c3;
      if (total + step_total > Lb) { overflow = true; break; }       // block-uniform
      const uint32_t pos = total + before + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
      if (keep) bI[pos] = tri;                                       // ascending order across waves and steps
      total += step_total;
    }
    __syncthreads();
    if (!overflow) { src_count = total; src_is_block_list = true; }
  }

  // ---- wave level, lane = candidate of the block list: third-edge rules, then the forms; survivors straight into the tile's records
  const size_t slot = (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 4u + wave;
  uint32_t* const out = p.wave_lists + slot * (1u + p.wave_cap) * kWaveRec;
  uint32_t count = 0;
  bool overflow = false;
  for (uint32_t base = 0; base < src_count; base += 64u) {           // (wave-uniform trip count)
    const uint32_t e = base + lane;
    const bool valid = e < src_count;
    const uint32_t ei = valid ? e : (src_count - 1u);
    const uint32_t tri = src_is_block_list ? bI[ei] : (mI != nullptr ? mI[ei] : ei);
    float4 A0 = p.tri_a[2u * tri], A1 = p.tri_a[2u * tri + 1u];
    float bz = p.tri_b[tri];
    bool keep = valid;
    float forms[18] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f,
                       0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};     // all-zero forms never reject
    if (fam.usable) {
      // (two calls that share no live ranges, as in the trace kernel of round 3: the S rules, then the forms)
      const bool miss3 = tile_misses_triangle<false, false, SlackProduct, true>(fam, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z});
      asm volatile("" : "+v"(A0.x), "+v"(A0.y), "+v"(A0.z), "+v"(A0.w), "+v"(A1.x), "+v"(A1.y), "+v"(A1.z), "+v"(A1.w), "+v"(bz));
      keep = valid && !miss3 && !tile_misses_triangle<true>(fam, {A1.z, A1.w, bz}, {A0.w, A1.x, A1.y}, {A0.x, A0.y, A0.z}, forms);
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
    if (count + static_cast<uint32_t>(__builtin_popcountll(m)) > p.wave_cap) { overflow = true; break; }
    const uint32_t pos = count + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
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
      r[4] = A0;
      r[5] = A1;
      r[6] = make_float4(bz, 0.0f, 0.0f, 0.0f);
    }
    count += static_cast<uint32_t>(__builtin_popcountll(m));
  }
  if (lane == 0u) {
    reinterpret_cast<float4*>(out)[0] = make_float4(__builtin_bit_cast(float, overflow ? kWaveOverflow : count), fam.fc[0], fam.fc[1], fam.fc[2]);
  }
}

// ------------------------------------------------------------------------------------
// dense_trace_kernel: grid = (ceil(W/32), ceil(rows/8)), 256 threads, no LDS.  One lane = one pixel, K samples in registers
// per pass, candidates in ascending order (first-scanned wins ties, Kernels.cuh:84) -- trace_kernel<.., PRE> without its
// classification.  FUSE: p.iters consecutive iterations of the host loop in one launch (see trace_kernel).
// ------------------------------------------------------------------------------------
template <bool FMA, int K, bool FUSE>
__global__ __launch_bounds__(256, RT_DENSE_WAVES) void dense_trace_kernel(const TraceParams p) {
  using M = Math<FMA>;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)));
  const uint32_t px = blockIdx.x * 32u + wave * 8u + (lane & 7u), ly = blockIdx.y * 8u + (lane >> 3);
  const bool inside = px < p.W && ly < p.rows;
  const uint32_t cxp = inside ? px : 0u, cyp = inside ? ly : 0u;     // out-of-image lanes shadow pixel 0
  const size_t pix = static_cast<size_t>(cxp) + static_cast<size_t>(cyp) * p.W;   // Kernels.cuh:128

  // the tile's list: read-only, wave-uniform addresses -> scalar loads
  const size_t slot = (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 4u + wave;
  const uint4* __restrict__ const recs = reinterpret_cast<const uint4*>(p.wave_lists) + slot * (1u + p.wave_cap) * (kWaveRec / 4u);
  const uint4 hdr = recs[0];
  const uint32_t list_count = hdr.x;
  const float fcx = __builtin_bit_cast(float, hdr.y), fcy = __builtin_bit_cast(float, hdr.z), fcz = __builtin_bit_cast(float, hdr.w);

  Rng rng;                                                           // :131
  rng.d = p.rng[0 * static_cast<size_t>(p.npix) + pix];
  rng.v0 = p.rng[1 * static_cast<size_t>(p.npix) + pix];
  rng.v1 = p.rng[2 * static_cast<size_t>(p.npix) + pix];
  rng.v2 = p.rng[3 * static_cast<size_t>(p.npix) + pix];
  rng.v3 = p.rng[4 * static_cast<size_t>(p.npix) + pix];
  rng.v4 = p.rng[5 * static_cast<size_t>(p.npix) + pix];
  V3 po, pd;
  pinhole<FMA>(p, cxp, p.row0 + cyp, po, pd);
  const V3 focal = focal_point<FMA>(p, pd);
  const uint32_t n = p.n_tris;
  const bool nearest = (p.flags & TRACE_NEAREST_HIT) != 0u;          // wave-uniform
  const unsigned long long lanes_in = __builtin_amdgcn_ballot_w64(inside);
  const float dFx = focal.x - fcx, dFy = focal.y - fcy, dFz = focal.z - fcz;   // this lane's focal point minus the tile's box centre
  unsigned long long st_unused[4] = {0, 0, 0, 0};

  // overflow fallback: the exact tests over the macro tile's list (or the whole scene)
  const uint32_t* mI = nullptr;
  uint32_t slow_count = n;
  if (list_count == kWaveOverflow && p.macro_lists != nullptr) {
    const uint32_t mt = (blockIdx.y * 8u / p.macro_h) * p.macro_nx + (blockIdx.x * 32u / p.macro_w);
    const uint32_t* const ml = p.macro_lists + static_cast<size_t>(mt) * (p.macro_cap + 1u);
    const uint32_t mc = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ml[0])));
    if (mc != 0xFFFFFFFFu) { mI = ml + 1; slow_count = mc; }
  }

  float ax = 0.0f, ay = 0.0f, az = 0.0f;                             // accu, :133
  const uint32_t iters = FUSE ? p.iters : 1u;
  float rx = 0.0f, ry = 0.0f, rz = 0.0f, rw = 0.0f;                  // FUSE: the pixel's RenderBuffer value so far
  uint32_t cnt_first = 0u;
  for (uint32_t it = 0; it < iters; ++it) {                          // FUSE: the host loop's iterations, :246
    if constexpr (FUSE) { ax = 0.0f; ay = 0.0f; az = 0.0f; }
    for (uint32_t s0 = 0; s0 < p.samples; s0 += K) {                 // :134, K samples per pass
      const uint32_t valid_k = (p.samples - s0 < static_cast<uint32_t>(K)) ? p.samples - s0 : static_cast<uint32_t>(K);
      V3 o[K], d[K];
      float best_t[K];
      int best_i[K];
      float dox[K], doy[K];                                          // each ray's lens offset do = o - oc
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (static_cast<uint32_t>(k) < valid_k) get_ray<FMA>(p, focal, rng, o[k], d[k]);   // :136
        else { o[k] = {0.0f, 0.0f, 0.0f}; d[k] = {0.0f, 0.0f, -1.0f}; }   // padding ray, result discarded
        best_t[k] = nearest ? FLT_MAX : -FLT_MAX;                    // :73
        best_i[k] = -1;
        dox[k] = o[k].x - p.cam[9]; doy[k] = o[k].y - p.cam[10];
      }
      if (list_count != kWaveOverflow) {
        for (uint32_t j = 0; j < list_count; ++j) {                  // ascending triangle order
          const uint4* __restrict__ const r = recs + (1u + j) * (kWaveRec / 4u);
          const uint4 u0 = r[0], u1 = r[1], u2 = r[2], u3 = r[3];
          auto f = [](uint32_t x) { return __builtin_bit_cast(float, x); };
          typedef _Float16 h2 __attribute__((ext_vector_type(2)));
          const h2 w0 = __builtin_bit_cast(h2, u2.y), w1 = __builtin_bit_cast(h2, u2.z), w2 = __builtin_bit_cast(h2, u2.w),
                   w3 = __builtin_bit_cast(h2, u3.x), w4 = __builtin_bit_cast(h2, u3.y);      // .y = high half
          // per lane and candidate: constant term + gradient . (this lane's focal point - box centre)
          const float b1 = __builtin_fmaf(static_cast<float>(w1.y), dFz, __builtin_fmaf(static_cast<float>(w0.x), dFy, __builtin_fmaf(static_cast<float>(w0.y), dFx, f(u0.x))));
          const float b2 = __builtin_fmaf(static_cast<float>(w2.x), dFz, __builtin_fmaf(static_cast<float>(w2.y), dFy, __builtin_fmaf(static_cast<float>(w1.x), dFx, f(u0.w))));
          const float b3 = __builtin_fmaf(static_cast<float>(w4.y), dFz, __builtin_fmaf(static_cast<float>(w3.x), dFy, __builtin_fmaf(static_cast<float>(w3.y), dFx, f(u1.z))));
          unsigned long long all_neg = ~0ull;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const float F1 = __builtin_fmaf(f(u0.z), doy[k], __builtin_fmaf(f(u0.y), dox[k], b1));
            const float F2 = __builtin_fmaf(f(u1.y), doy[k], __builtin_fmaf(f(u1.x), dox[k], b2));
            const float F3 = __builtin_fmaf(f(u2.x), doy[k], __builtin_fmaf(f(u1.w), dox[k], b3));
            const float worst = __builtin_fminf(__builtin_fminf(F1, F2), F3);
            all_neg &= __builtin_amdgcn_ballot_w64(worst < 0.0f) | ((static_cast<uint32_t>(k) < valid_k) ? 0ull : ~0ull);
          }
          if ((~all_neg & lanes_in) != 0ull) {                       // some ray of the wave survives the forms: the reference's tests
            const uint4 a0 = r[4], a1 = r[5];
            const uint32_t bzw = r[6].x;
            const float4 A0 = make_float4(f(a0.x), f(a0.y), f(a0.z), f(a0.w)), A1 = make_float4(f(a1.x), f(a1.y), f(a1.z), f(a1.w));
            test_triangle<FMA, K, true, false>(A0, A1, [&] { return f(bzw); }, static_cast<int>(u3.z), o, d, best_t, best_i,
                                               nearest, inside, valid_k, st_unused, st_unused);
          }
        }
      } else {
        for (uint32_t e = 0; e < slow_count; ++e) {                  // ascending: the macro lists ascend
          const uint32_t tri = mI != nullptr ? static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(mI[e]))) : e;
          const float4 A0 = p.tri_a[2u * tri], A1 = p.tri_a[2u * tri + 1u];
          const float bz = p.tri_b[tri];
          test_triangle<FMA, K, true, false>(A0, A1, [&] { return bz; }, static_cast<int>(tri), o, d, best_t, best_i,
                                             nearest, inside, valid_k, st_unused, st_unused);
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
      if (it == 0u && !(p.flags & TRACE_ZERO_ACC) && inside) {
        const float4 r0 = p.render[pix];
        rx = r0.x; ry = r0.y; rz = r0.z; rw = r0.w;
        cnt_first = p.counts[pix];
      }
      rx += ax; ry += ay; rz += az;
    }
  }
  if (inside) {
    float4 acc_in = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t cnt_in = 0u;
    if constexpr (!FUSE) {
      if (!(p.flags & TRACE_ZERO_ACC)) {                              // wave-uniform
        acc_in = p.render[pix];
        cnt_in = p.counts[pix];
      }
    }
    const uint32_t cnt = FUSE ? cnt_first + iters * p.samples : cnt_in + p.samples;   // :140
    p.counts[pix] = cnt;
    float4 acc = acc_in;
    if constexpr (FUSE) {
      acc = make_float4(rx, ry, rz, rw);
    } else {
      acc.x += ax; acc.y += ay; acc.z += az;                          // :141-143, alpha untouched (:144)
    }
    p.render[pix] = acc;
    if (p.flags & TRACE_EMIT_IMAGE) {                                 // fused rt::ConverterKernel, :164-168
      const float c = static_cast<float>(cnt);
      const uint32_t bgra = rtd::pack_color(255.0f * (acc.x / c), 255.0f * (acc.y / c), 255.0f * (acc.z / c));
      p.image[pix] = bgra;
      if (p.image_host != nullptr) p.image_host[pix] = bgra;
    }
    p.rng[0 * static_cast<size_t>(p.npix) + pix] = rng.d;             // :146
    p.rng[1 * static_cast<size_t>(p.npix) + pix] = rng.v0;
    p.rng[2 * static_cast<size_t>(p.npix) + pix] = rng.v1;
    p.rng[3 * static_cast<size_t>(p.npix) + pix] = rng.v2;
    p.rng[4 * static_cast<size_t>(p.npix) + pix] = rng.v3;
    p.rng[5 * static_cast<size_t>(p.npix) + pix] = rng.v4;
  }
}

}  // namespace rtk
