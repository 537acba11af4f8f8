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
#include "rt_lists.hpp"          // -> rt_classify.hpp -> rt_rays.hpp -> rt_device_math.hpp, rt_kernels.hpp

namespace rtk {

#ifndef RT_TRACE_MIN_WAVES
#define RT_TRACE_MIN_WAVES 4     // __launch_bounds__ 2nd argument: waves per SIMD the allocator must allow
                                 // (<= 128 VGPRs; measured C3 213 -> 193 us, C4 27.2 -> 24.3 ms vs the 136-VGPR build)
#endif
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES(K) RT_TRACE_MIN_WAVES
#endif
#ifndef RT_HBM_WAVES
#define RT_HBM_WAVES 4           // the dense-scene kernels that read their lists from HBM (A/B: 5 = at most 102 VGPRs)
#endif
#ifndef RT_SMALL_WG_WAVES
#define RT_SMALL_WG_WAVES 4      // small-scene trace kernels: waves per workgroup (4, 2 or 1; see trace_kernel)
#endif
// candidate records per wave in LDS (40 bytes each): TraceParams::bin_list, a multiple of 64

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
// HBM (PRE kernels only): the tiles' candidate lists + forms come from p.wave_lists (wave_lists_kernel, rt_dense.hpp) -- this
// instantiation contains no classification and no barrier; a tile marked as overflowing tests its macro tile's list.
constexpr uint32_t kWaveRec = 16u;                 // dwords per record of p.wave_lists
constexpr uint32_t kWaveOverflow = 0xFFFFFFFFu;
template <bool FMA, int K, bool FILTER, bool STATS, bool BIN, bool ONEPASS, bool FUSE = false, bool PRE = false, bool HBM = false>
__global__ __launch_bounds__((BIN && ONEPASS) ? 64 * RT_SMALL_WG_WAVES : 256, (ONEPASS && K == 2) ? 5 : HBM ? RT_HBM_WAVES : RT_TRACE_WAVES(K)) void trace_kernel(const TraceParams p) {
  static_assert(!HBM || (PRE && BIN && !ONEPASS && FILTER && !STATS), "lists from HBM: the default dense-scene kernels only");
  using M = Math<FMA>;
  extern __shared__ float4 s_mem[];
  // Small scenes: a 32 x 8 block of four tiles is traced by 4 / WGW workgroups of WGW waves (no wave of these kernels talks to
  // another one).  A CU holds at most 16 multi-wave workgroups (one barrier resource each, kept until the LAST wave of the
  // workgroup ends): in a frame that mixes short certain-winner waves with long ray-generating ones a four-wave workgroup is
  // soon down to its slowest wave and 16 of them leave wave slots empty.
  constexpr uint32_t WGW = (BIN && ONEPASS) ? RT_SMALL_WG_WAVES : 4u, WGS = 4u / WGW;

  // (workgroup shape, measured at C3 with the lists rebuilt: 256 threads = one block of four tiles 62.5 us per step; one wave
  //  per workgroup 85.0; two / four blocks per workgroup 67.2 / 75.2; 8 instead of 7 waves per SIMD at 64 VGPRs 63.2)
  const uint32_t lane = threadIdx.x & 63u, wl = threadIdx.x >> 6;   // wl: wave within its workgroup
  const uint32_t wave = (blockIdx.x % WGS) * WGW + wl;             // tile within the block of 32 x 8 pixels
  const uint32_t bx = blockIdx.x / WGS;                            // the block this workgroup traces (a part of)
  uint32_t by = blockIdx.y;
  const uint32_t gxb = gridDim.x / WGS;                            // blocks per block row
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
  float4* const cA = s_mem + static_cast<size_t>(wl) * (2u * L);                           // 2 float4 per candidate
  float* const cB = reinterpret_cast<float*>(s_mem + WGW * 2u * L) + wl * L;
  int* const cI = reinterpret_cast<int*>(s_mem + WGW * 2u * L) + WGW * L + wl * L;
  // PRETEST (large-scene kernels, TRACE_PRETEST): one 64-byte block more per candidate behind the index
  // array -- the three per-sample forms of tile_misses_triangle<true> (9 floats) and their focal-point
  // gradients (9 fp16 in 5 dwords), contiguous so that one address register and four ds_read_b128
  // with immediate offsets fetch them.  L is a multiple of 2 here.
  constexpr bool PRETEST = PRE && BIN && !ONEPASS;
  const bool pretest = PRETEST && (p.flags & TRACE_PRETEST) != 0u;   // wave-uniform
  float4* const cP = s_mem + 4u * 2u * L + 2u * L + static_cast<size_t>(wave) * (4u * L);   // after cA (8L float4), cB + cI (2L float4)
  const uint32_t list_floats4 = pretest ? (4u * 2u * L + 2u * L + 16u * L) : (4u * 2u * L + 2u * L);   // float4 units before the block list
  // HBM: the candidates' colours travel with them (3 floats each, behind the forms: this instantiation has no block list).  A
  // hit is remembered as its SLOT in the wave's list -- ascending like the triangle indices, so ties break the same way -- and
  // shaded from LDS: the per-sample gather p.tri_color[winner] was an exposed global-memory round trip per sample batch
  // (24 % of the dense-scene kernel's wave cycles sat in s_waitcnt: profiles/r04_c4_stalls.txt).
  float* const cC = reinterpret_cast<float*>(s_mem + list_floats4) + wave * (3u * L);

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
  bool hbm_overflow = false;        // HBM: the tile's list did not fit its slot -> exact tests over the macro tile's list
  if constexpr (HBM) {
    const size_t slot = (static_cast<size_t>(by) * gxb + bx) * 4u + wave;
    const float4* const rec = reinterpret_cast<const float4*>(p.wave_lists) + slot * (1u + p.wave_cap) * (kWaveRec / 4u);
    const float4 hdr = rec[0];
    const uint32_t count = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, hdr.x)));
    hbm_overflow = count == kWaveOverflow;
    fam.usable = true;
    fam.fc[0] = uniform(hdr.y); fam.fc[1] = uniform(hdr.z); fam.fc[2] = uniform(hdr.w);
    list_count = hbm_overflow ? 0u : count;
    for (uint32_t e = lane; e < list_count; e += 64u) {              // the records into this wave's LDS slot, as classify() leaves them
      const float4* const r = rec + (1u + e) * (kWaveRec / 4u);
      const float4 f0 = r[0], f1 = r[1], f2 = r[2], f3 = r[3];
      const uint32_t tri = __builtin_bit_cast(uint32_t, f3.z);
      const float4 a0 = p.tri_a[2u * tri], a1 = p.tri_a[2u * tri + 1u], c0 = p.tri_color[tri];   // (the scene's tables: L2-resident)
      const float bz = p.tri_b[tri];
      cP[4u * e] = f0; cP[4u * e + 1u] = f1; cP[4u * e + 2u] = f2; cP[4u * e + 3u] = make_float4(f3.x, f3.y, 0.0f, 0.0f);
      cA[2u * e] = a0; cA[2u * e + 1u] = a1;
      cB[e] = bz;
      cI[e] = static_cast<int>(tri);
      cC[3u * e] = c0.x; cC[3u * e + 1u] = c0.y; cC[3u * e + 2u] = c0.z;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // this wave's ds_writes before its ds_reads
    __builtin_amdgcn_wave_barrier();
    list_complete = true;
  }
  if constexpr (BIN && !ONEPASS && !HBM) {
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
  if constexpr (PRETEST && !HBM) {
    const uint32_t next0 = classify(0u, std::true_type{});
    list_complete = next0 >= src_count;
    forms_ready = pretest && list_complete;
  }
  if constexpr (HBM) forms_ready = true;
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
    unsigned long long pad_mask[K];                                 // PRETEST: all ones for the padding samples of a partial batch (they never keep a candidate alive)
#pragma unroll
    for (int k = 0; k < K; ++k) pad_mask[k] = (static_cast<uint32_t>(k) < valid_k) ? 0ull : ~0ull;
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
        if constexpr (!HBM) {
          if (!list_complete) {
            next = classify(base, std::false_type{});
            if (base == 0u && next >= src_count) list_complete = true;     // (!PRE: the first classification happens here)
          }
        }
        // does ANY ray of the wave survive the per-sample forms of candidate j?  (wave-uniform)
        auto forms_eval = [&](const float4 f0, const float4 f1, const float4 q2, const float4 q3) -> bool {
          // per-sample forms of this candidate at each ray's own lens origin: does ANY ray of the wave survive?
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
            all_neg &= __builtin_amdgcn_ballot_w64(worst < 0.0f) | pad_mask[k];
          }
          const bool alive = (~all_neg & lanes_in) != 0ull;
          if constexpr (STATS) { if (!alive) st_pre += 1; }
          return alive;
        };
        auto forms_alive = [&](uint32_t j) -> bool { return forms_eval(cP[4u * j], cP[4u * j + 1u], cP[4u * j + 2u], cP[4u * j + 3u]); };
        auto run_tests = [&](uint32_t j) {
          const float4 A0 = cA[2u * j], A1 = cA[2u * j + 1u];
          test_triangle<FMA, K, FILTER, STATS>(A0, A1, [&] { return cB[j]; }, HBM ? static_cast<int>(j) : cI[j], o, d, best_t, best_i,
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
      if constexpr (HBM) {
        if (hbm_overflow) {                                        // (wave-uniform, rare) every triangle of the macro tile's list, ascending
          for (uint32_t e = 0; e < n_src; ++e) {
            const uint32_t tri = mI != nullptr ? static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(mI[e]))) : e;
            const float4 A0 = p.tri_a[2u * tri], A1 = p.tri_a[2u * tri + 1u];
            const float bz = p.tri_b[tri];
            test_triangle<FMA, K, FILTER, STATS>(A0, A1, [&] { return bz; }, static_cast<int>(tri), o, d, best_t, best_i,
                                                 nearest, inside, valid_k, st_exit, st_skip);
          }
        }
      }
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
        // (HBM: a triangle hit is its slot in the wave's list -- unless the tile overflowed and tested triangles by index)
        const bool by_slot = HBM && !hbm_overflow;                  // wave-uniform
        const int first_sphere = by_slot ? 0x40000000 : static_cast<int>(n);
        for (uint32_t si = 0; si < p.n_spheres; ++si) {
          float t = 0.0f;
          if (hit_sphere<FMA>(o[k], d[k], p.spheres[si], t) && (nearest ? (t > 0.0f && t < dist) : dist < t)) {
            dist = t;
            win = first_sphere + static_cast<int>(si);
          }
        }
        float r, g, b;
        if (win >= 0) {
          if (win < first_sphere) {
            if (p.tri_n != nullptr) {
              // build-defined smooth shading: vertex normals interpolated at the winner's barycentrics;
              // u, v are recomputed from the winner's record (same arithmetic as the scan: same bits)
              if (by_slot) win = cI[win];
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
            } else if (by_slot) {
              r = cC[3 * win]; g = cC[3 * win + 1]; b = cC[3 * win + 2];
            } else {
              const float4 col = p.tri_color[win];
              r = col.x; g = col.y; b = col.z;
            }
          } else {
            const float4 sph = p.spheres[win - first_sphere];
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
