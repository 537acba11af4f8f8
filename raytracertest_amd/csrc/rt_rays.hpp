// rt_rays.hpp -- ray generation and the reference's exact intersection tests (RayTracer/Kernels.cuh:29-65,
// ThinLensCamera.cuh:30-52,111-130; the build-defined ray-sphere test).  Shared by the trace kernels, the list builders
// (pinhole rays of the tile corners) and the dbg harnesses.  Included through rt_trace.hpp.
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


}  // namespace rtk
