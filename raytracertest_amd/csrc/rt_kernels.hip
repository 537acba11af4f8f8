// rt_kernels.hip -- hand-written gfx950 kernels of the trace path.
//
//   rng_init_kernel    <- random::InitRandomStates        (RayTracer/Random.cu:10-30)
//   prep_triangles     <- per-triangle invariants hoisted out of rt::Radiance
//   trace_kernel       <- rt::TraceKernel + Radiance + HitTriangle + ThinLensCamera::GetRay
//                         (RayTracer/Kernels.cuh:29-147, ThinLensCamera.cuh:30-52,111-130)
//   convert_kernel     <- rt::ConverterKernel              (RayTracer/Kernels.cuh:149-169)
//   dbg_* kernels      <- single-function harnesses used by the parity tests
//
// Execution shape (MI355X: 256 CUs x 4 SIMD, wave64, 160 KiB LDS/CU):
//   * one lane = one pixel; a wave covers an 8x8 pixel tile (coherent rays -> the
//     wave-uniform skips below fire often), a 256-thread block covers 32x8 pixels so
//     every 128-byte line of the per-pixel buffers is written whole by one block;
//   * every lane keeps K samples of its pixel in registers and tests them against one
//     triangle at a time; the triangle record (v0, e1, e2) is read from LDS with three
//     wave-uniform ds_read_b128 (broadcast), amortised over 64*K rays;
//   * triangles are staged into LDS per block in chunks (whole scene when it fits),
//     always scanned in ascending order (first-scanned wins ties, Kernels.cuh:84);
//   * __ballot-driven wave-uniform skips after the culling test, the u test and the
//     v test; the IEEE division only runs for triangles some lane may really hit.
#include "rt_device_math.hpp"
#include "rt_kernels.hpp"

#include <float.h>

namespace rtk {

using rtd::Math;
using rtd::Rng;
using rtd::V3;

// ------------------------------------------------------------------------------------
// RNG state creation: curand_init(seed, subsequence = global pixel index, offset 0).
// v <- (T^(2^67))^p v as a product of the precomputed powers jump[k] = J^(2^k)
// (160 columns x 8 words each, 5 used), column loads are wave-uniform (scalar).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rng_init_kernel(uint32_t* __restrict__ rng, uint32_t npix,
                                                        uint32_t p0, Rng seeded,
                                                        const uint32_t* __restrict__ jump) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const bool inside = i < npix;
  const uint32_t p = p0 + (inside ? i : 0u);      // wraps like the reference's 32-bit x + y*W
  uint32_t v[5] = {seeded.v0, seeded.v1, seeded.v2, seeded.v3, seeded.v4};
  for (int k = 0; k < 32; ++k) {
    const bool bit = (p >> k) & 1u;
    if (__builtin_amdgcn_ballot_w64(bit) == 0ull) continue;      // no lane of this wave needs J^(2^k)
    const uint32_t* __restrict__ M = jump + static_cast<size_t>(k) * 160u * 8u;
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;
#pragma unroll
    for (int w = 0; w < 5; ++w) {
      const uint32_t bits = v[w];
      for (int b = 0; b < 32; ++b) {
        const uint32_t* __restrict__ col = M + (w * 32 + b) * 8;
        const uint32_t m = 0u - ((bits >> b) & 1u);
        r0 ^= col[0] & m; r1 ^= col[1] & m; r2 ^= col[2] & m; r3 ^= col[3] & m; r4 ^= col[4] & m;
      }
    }
    if (bit) { v[0] = r0; v[1] = r1; v[2] = r2; v[3] = r3; v[4] = r4; }
  }
  if (inside) {
    rng[0 * static_cast<size_t>(npix) + i] = seeded.d;
    rng[1 * static_cast<size_t>(npix) + i] = v[0];
    rng[2 * static_cast<size_t>(npix) + i] = v[1];
    rng[3 * static_cast<size_t>(npix) + i] = v[2];
    rng[4 * static_cast<size_t>(npix) + i] = v[3];
    rng[5 * static_cast<size_t>(npix) + i] = v[4];
  }
}

// ------------------------------------------------------------------------------------
// Per-triangle invariants.  e1 = v1 - v0 and e2 = v2 - v0 are the same fp32 subtractions
// HitTriangle performs per ray (Kernels.cuh:37-38); colour = abs(normalize(cross(e1,e2)))
// is the shade of a hit (Kernels.cuh:97-99), a function of the triangle only.
// Record layout (36 bytes per triangle, what the trace kernel stages into LDS):
//   tri_a[2i]   = (e2.x, e2.y, e2.z, e1.x)      stage A reads tri_a[2i], tri_a[2i+1]
//   tri_a[2i+1] = (e1.y, e1.z, v0.x, v0.y)      (two ds_read_b128, wave-uniform)
//   tri_b[i]    = v0.z                          stage B adds one ds_read_b32
// ------------------------------------------------------------------------------------
template <bool FMA>
__global__ __launch_bounds__(256) void prep_triangles_kernel(const float4* __restrict__ verts, uint32_t n,
                                                              float4* __restrict__ tri_a,
                                                              float* __restrict__ tri_b,
                                                              float4* __restrict__ color) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const float4 a = verts[3 * i + 0], b = verts[3 * i + 1], c = verts[3 * i + 2];
  const V3 v0 = {a.x, a.y, a.z};
  const V3 e1 = rtd::sub({b.x, b.y, b.z}, v0);
  const V3 e2 = rtd::sub({c.x, c.y, c.z}, v0);
  tri_a[2 * i + 0] = make_float4(e2.x, e2.y, e2.z, e1.x);
  tri_a[2 * i + 1] = make_float4(e1.y, e1.z, v0.x, v0.y);
  tri_b[i] = v0.z;
  const V3 nn = Math<FMA>::normalize(Math<FMA>::cross(e1, e2));
  color[i] = make_float4(rtd::absf(nn.x), rtd::absf(nn.y), rtd::absf(nn.z), 0.0f);
}

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

// ThinLensCamera::GetRay, ThinLensCamera.cuh:30-52; pd is the pixel's pinhole direction
template <bool FMA>
__device__ __forceinline__ void get_ray(const TraceParams& p, V3 pd, Rng& rng, V3& o, V3& d) {
  using M = Math<FMA>;
  float dx, dy;
  rtd::uniform_on_disk(rng, dx, dy);                                              // :41
  const V3 pos = {p.cam[9], p.cam[10], p.cam[11]};                                // Position(), :54-57
  const V3 off = {dx * p.aperture, dy * p.aperture, 0.0f};
  const V3 focal = {M::madd1(p.focal, pd.x, pos.x), M::madd1(p.focal, pd.y, pos.y),
                    M::madd1(p.focal, pd.z, pos.z)};                              // :44
  o = rtd::add(pos, off);                                                         // :47
  d = M::normalize(rtd::sub(focal, o));                                           // :50
}

// ------------------------------------------------------------------------------------
// The trace kernel.  grid = (ceil(W/32), ceil(rows/8)), block = 256 threads,
// dynamic LDS = min(n_tris, chunk) * 36 bytes.
//
// FILTER: three wave-uniform early-outs per triangle, decided with __ballot on
// CONSERVATIVE per-ray rejections -- a ray is only ever dropped when the reference's own
// test is certain to miss, and a triangle is skipped only when every ray of the wave is
// dropped; whenever any ray survives, stage D evaluates the reference's exact test
// (division included) for all lanes from the values already computed.  With
// u = fl(U*inv), v = fl(V*inv), inv = fl(1/det), det >= 1e-10 (not culled):
//   U > fl(det*1.0001)            => U/det > 1.00009            => u > 1      (miss, :51)
//   U < fl(det*-1e-6)             => U/det < -0.99e-6 (normal)  => u < 0      (miss, :51)
//   V < fl(det*-1e-6)             =>                               v < 0      (miss, :58)
//   U+V > fl(det*1.0001), with U,V >= -1e-6 det (not dropped above)
//                                 => u+v > 1.0001 - 4e-6 - roundoff > 1       (miss, :58)
// NaN/inf operands make every comparison false: the ray is kept and stage D decides.
// tests/test_gpu_parity.py::test_filter_off_equals_filter_on checks FILTER against the
// plain reference-order path bit for bit.
// ------------------------------------------------------------------------------------
#define RT_EPS 0.0000000001f
#ifndef RT_TRACE_MIN_WAVES
#define RT_TRACE_MIN_WAVES 1     // __launch_bounds__ 2nd argument: waves per SIMD the allocator must allow
#endif

template <bool FMA, int K, bool FILTER, bool STATS>
__global__ __launch_bounds__(256, RT_TRACE_MIN_WAVES) void trace_kernel(const TraceParams p) {
  using M = Math<FMA>;
  extern __shared__ float4 s_mem[];

  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t px = blockIdx.x * 32u + wave * 8u + (lane & 7u);
  const uint32_t ly = blockIdx.y * 8u + (lane >> 3);
  const bool inside = px < p.W && ly < p.rows;
  const uint32_t cxp = inside ? px : 0u, cyp = inside ? ly : 0u;   // out-of-image lanes shadow pixel 0
  const size_t pix = static_cast<size_t>(cxp) + static_cast<size_t>(cyp) * p.W;   // Kernels.cuh:128

  Rng rng;                                                         // :131
  rng.d = p.rng[0 * static_cast<size_t>(p.npix) + pix];
  rng.v0 = p.rng[1 * static_cast<size_t>(p.npix) + pix];
  rng.v1 = p.rng[2 * static_cast<size_t>(p.npix) + pix];
  rng.v2 = p.rng[3 * static_cast<size_t>(p.npix) + pix];
  rng.v3 = p.rng[4 * static_cast<size_t>(p.npix) + pix];
  rng.v4 = p.rng[5 * static_cast<size_t>(p.npix) + pix];

  V3 po, pd;
  pinhole<FMA>(p, cxp, p.row0 + cyp, po, pd);

  const uint32_t n = p.n_tris;
  const uint32_t cap = n < p.chunk ? n : p.chunk;                  // triangles resident in LDS
  float4* const sA = s_mem;                                        // 2 float4 per triangle
  float* const sB = reinterpret_cast<float*>(s_mem + 2u * cap);    // 1 float per triangle
  const bool single_chunk = n <= p.chunk;
  if (single_chunk) {
    for (uint32_t i = threadIdx.x; i < 2u * n; i += 256u) sA[i] = p.tri_a[i];
    for (uint32_t i = threadIdx.x; i < n; i += 256u) sB[i] = p.tri_b[i];
    __syncthreads();
  }

  float ax = 0.0f, ay = 0.0f, az = 0.0f;                           // accu, :133
  unsigned long long st_exit[4] = {0, 0, 0, 0};                    // STATS: lane-tests by exit point
  unsigned long long st_skip[4] = {0, 0, 0, 0};                    // STATS: wave-triangles skipped after A/B/C, reaching D

  for (uint32_t s0 = 0; s0 < p.samples; s0 += K) {                 // :134, K samples per pass
    V3 o[K], d[K];
    float best_t[K];
    int best_i[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (s0 + k < p.samples) get_ray<FMA>(p, pd, rng, o[k], d[k]);   // :136
      else { o[k] = po; d[k] = pd; }                               // padding ray, result discarded
      best_t[k] = -FLT_MAX;                                        // :73
      best_i[k] = -1;
    }

    for (uint32_t c0 = 0; c0 < n; c0 += p.chunk) {
      const uint32_t cn = (n - c0 < p.chunk) ? n - c0 : p.chunk;
      if (!single_chunk) {
        __syncthreads();                                           // everyone done with the previous chunk
        for (uint32_t i = threadIdx.x; i < 2u * cn; i += 256u) sA[i] = p.tri_a[2u * c0 + i];
        for (uint32_t i = threadIdx.x; i < cn; i += 256u) sB[i] = p.tri_b[c0 + i];
        __syncthreads();
      }
      for (uint32_t j = 0; j < cn; ++j) {                          // :75, ascending order
        const float4 A0 = sA[2u * j + 0], A1 = sA[2u * j + 1];
        const V3 e2 = {A0.x, A0.y, A0.z}, e1 = {A0.w, A1.x, A1.y};
        const int tri_index = static_cast<int>(c0 + j);

        if constexpr (!FILTER) {
          const V3 v0 = {A1.z, A1.w, sB[j]};
#pragma unroll
          for (int k = 0; k < K; ++k) {
            float t = 0.0f, u = 0.0f, v = 0.0f;
            int stage;
            const bool h = hit_triangle_exact<FMA>(o[k], d[k], v0, e1, e2, RT_EPS, t, u, v, stage);
            if (h && best_t[k] < t) {                              // :84
              best_t[k] = t;
              best_i[k] = tri_index;
            }
            if constexpr (STATS) {
              const bool counted = inside && (s0 + k < p.samples);
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
          if (live == 0ull) { if constexpr (STATS) st_skip[0]++; continue; }   // whole wave culled

          // stage B: U = dot(origin - v0, pv) (:49-50), conservative u rejection
          const V3 v0 = {A1.z, A1.w, sB[j]};
          V3 tv[K];
          float U[K], thi[K], tlo[K];
          live = 0ull;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            tv[k] = rtd::sub(o[k], v0);
            U[k] = M::dot(tv[k], pv[k]);
            thi[k] = det[k] * 1.0001f;
            tlo[k] = det[k] * -1e-6f;
            mk[k] &= __builtin_amdgcn_ballot_w64(!(U[k] > thi[k])) &
                     __builtin_amdgcn_ballot_w64(!(U[k] < tlo[k]));
            live |= mk[k];
          }
          if (live == 0ull) { if constexpr (STATS) st_skip[1]++; continue; }

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
          if (live == 0ull) { if constexpr (STATS) st_skip[2]++; continue; }
          if constexpr (STATS) st_skip[3]++;

          // stage D: the reference's exact tests (:42-63, :84) wherever a ray may hit
#pragma unroll
          for (int k = 0; k < K; ++k) {
            if (mk[k] != 0ull) {
              const float inv = 1.0f / det[k];                     // :47
              const float u = U[k] * inv;                          // :50
              const float v = V[k] * inv;                          // :57
              const float t = M::dot(e2, qv[k]) * inv;             // :63
              const bool miss = (det[k] < RT_EPS) | (u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f);
              const bool upd = (!miss) & (best_t[k] < t);          // :84
              best_t[k] = upd ? t : best_t[k];
              best_i[k] = upd ? tri_index : best_i[k];
            }
          }
        }
      }
    }

    // spheres continue the same farthest-hit scan, then shade in sample order (:95-104, :137)
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (s0 + k < p.samples) {
        float dist = best_t[k];
        int win = best_i[k];
        for (uint32_t si = 0; si < p.n_spheres; ++si) {
          float t = 0.0f;
          if (hit_sphere<FMA>(o[k], d[k], p.spheres[si], t) && dist < t) {
            dist = t;
            win = static_cast<int>(n + si);
          }
        }
        float r, g, b;
        if (win >= 0) {
          if (win < static_cast<int>(n)) {
            const float4 col = p.tri_color[win];
            r = col.x; g = col.y; b = col.z;
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

  if (inside) {
    p.counts[pix] += p.samples;                                     // :140
    float4 acc = p.render[pix];
    acc.x += ax; acc.y += ay; acc.z += az;                          // :141-143, alpha untouched (:144)
    p.render[pix] = acc;
    p.rng[0 * static_cast<size_t>(p.npix) + pix] = rng.d;           // :146
    p.rng[1 * static_cast<size_t>(p.npix) + pix] = rng.v0;
    p.rng[2 * static_cast<size_t>(p.npix) + pix] = rng.v1;
    p.rng[3 * static_cast<size_t>(p.npix) + pix] = rng.v2;
    p.rng[4 * static_cast<size_t>(p.npix) + pix] = rng.v3;
    p.rng[5 * static_cast<size_t>(p.npix) + pix] = rng.v4;
  }
  if constexpr (STATS) {
    if (lane == 0 && p.stats != nullptr) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        atomicAdd(p.stats + e, st_exit[e]);
        atomicAdd(p.stats + 4 + e, st_skip[e]);
      }
    }
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
  get_ray<FMA>(p, pd, r, o, d);
  float* out = rays + 6 * static_cast<size_t>(i);
  out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = d.x; out[4] = d.y; out[5] = d.z;
  s[0] = r.d; s[1] = r.v0; s[2] = r.v1; s[3] = r.v2; s[4] = r.v3; s[5] = r.v4;
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

hipError_t launch_rng_init(uint32_t* rng, uint32_t npix, uint32_t p0, const uint32_t seeded[6],
                           const uint32_t* jump, hipStream_t st) {
  Rng s = {seeded[0], seeded[1], seeded[2], seeded[3], seeded[4], seeded[5]};
  hipLaunchKernelGGL(rng_init_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, st, rng, npix, p0, s, jump);
  return hipGetLastError();
}

hipError_t launch_prep_triangles(bool fma, const float4* verts, uint32_t n, float4* tri_a, float* tri_b,
                                 float4* color, hipStream_t st) {
  if (n == 0) return hipSuccess;
  if (fma) hipLaunchKernelGGL(prep_triangles_kernel<true>, dim3(cdiv(n, 256)), dim3(256), 0, st, verts, n, tri_a, tri_b, color);
  else hipLaunchKernelGGL(prep_triangles_kernel<false>, dim3(cdiv(n, 256)), dim3(256), 0, st, verts, n, tri_a, tri_b, color);
  return hipGetLastError();
}

template <bool FMA, bool FILTER, bool STATS>
static void launch_trace_k(const TraceParams& p, int K, dim3 grid, size_t lds, hipStream_t st) {
  switch (K) {
    case 1: hipLaunchKernelGGL((trace_kernel<FMA, 1, FILTER, STATS>), grid, dim3(256), lds, st, p); break;
    case 2: hipLaunchKernelGGL((trace_kernel<FMA, 2, FILTER, STATS>), grid, dim3(256), lds, st, p); break;
    default: hipLaunchKernelGGL((trace_kernel<FMA, 4, FILTER, STATS>), grid, dim3(256), lds, st, p); break;
  }
}

uint32_t trace_lds_bytes(const TraceParams& p) {
  const uint32_t staged = p.n_tris < p.chunk ? p.n_tris : p.chunk;
  return staged * 36u;
}

hipError_t launch_trace(const TraceParams& p, bool fma, bool filter, int K, hipStream_t st) {
  if (p.rows == 0 || p.W == 0 || p.samples == 0) return hipSuccess;
  const dim3 grid(cdiv(p.W, 32), cdiv(p.rows, 8));
  const size_t lds = trace_lds_bytes(p);
  if (p.stats != nullptr) {          // instrumented build of the same kernel (not the timed path)
    if (fma) { if (filter) launch_trace_k<true, true, true>(p, K, grid, lds, st); else launch_trace_k<true, false, true>(p, K, grid, lds, st); }
    else { if (filter) launch_trace_k<false, true, true>(p, K, grid, lds, st); else launch_trace_k<false, false, true>(p, K, grid, lds, st); }
  } else {
    if (fma) { if (filter) launch_trace_k<true, true, false>(p, K, grid, lds, st); else launch_trace_k<true, false, false>(p, K, grid, lds, st); }
    else { if (filter) launch_trace_k<false, true, false>(p, K, grid, lds, st); else launch_trace_k<false, false, false>(p, K, grid, lds, st); }
  }
  return hipGetLastError();
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

}  // namespace rtk
