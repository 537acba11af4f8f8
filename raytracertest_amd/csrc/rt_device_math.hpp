// rt_device_math.hpp -- gfx950 device arithmetic of the trace path.
//
// Numeric spec (DESIGN.md "numeric spec"): IEEE-754 binary32, correctly rounded / and
// sqrt (hipcc default), denormals kept, the whole library compiled with
// -ffp-contract=off so the ONLY fused multiply-adds are the explicit ones below.
// Math<true>  ("fma", default): the a*b+c shapes nvcc's default -fmad=true would fuse in
//                               the reference's GPU build (RayTracer/RayTracer.vcxproj:65-70).
// Math<false> ("strict"):       every source-level operation rounds on its own.
// Citations are to the reference: RayTracer/Kernels.cuh, ThinLensCamera.cuh, Ray.cuh,
// Random.cuh, Random.cu, DeviceUtils.cuh.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtd {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ float absf(float f) { return (f < 0.0f) ? -f : f; }   // Kernels.cuh:16-19

// ---- correctly rounded sqrt and reciprocal for operands in [2^-96, 2^96] ------------------------
// The compiler's IEEE expansions of sqrtf (17 instructions) and 1.0f/y (12) spend 7 and 5 of
// them on operand scaling (denormal and overflow ranges) and special values.  For mid-range
// operands those steps are no-ops; what remains is reproduced here instruction for instruction.
__device__ __forceinline__ bool midrange(float x) {                 // 2^-96 <= x <= 2^96 (false for NaN, inf, <= 0)
  return (__builtin_bit_cast(uint32_t, x) - 0x0F800000u) <= (0x6F800000u - 0x0F800000u);
}
__device__ __forceinline__ float sqrt_midrange(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);                        // v_sqrt_f32, 1 ulp
  const float s_dn = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
  const float s_up = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
  const float r_dn = __builtin_fmaf(-s_dn, s, x);
  const float r_up = __builtin_fmaf(-s_up, s, x);
  float r = (r_dn <= 0.0f) ? s_dn : s;
  r = (r_up > 0.0f) ? s_up : r;
  return r;
}
__device__ __forceinline__ float rcp_midrange(float y) {
  float r = __builtin_amdgcn_rcpf(y);                               // v_rcp_f32, 1 ulp
  const float e0 = __builtin_fmaf(-y, r, 1.0f);
  r = __builtin_fmaf(e0, r, r);
  const float e1 = __builtin_fmaf(-y, r, 1.0f);                     // numerator 1: the quotient estimate is r itself
  const float q = __builtin_fmaf(e1, r, r);
  const float e2 = __builtin_fmaf(-y, q, 1.0f);
  return __builtin_fmaf(e2, r, q);
}

template <bool FMA>
struct Math {
  // x*y - z*w
  static __device__ __forceinline__ float msub2(float x, float y, float z, float w) {
    if constexpr (FMA) return __builtin_fmaf(x, y, -(z * w));
    else return (x * y) - (z * w);
  }
  // a*b + c*d
  static __device__ __forceinline__ float madd2(float a, float b, float c, float d) {
    if constexpr (FMA) return __builtin_fmaf(a, b, c * d);
    else return (a * b) + (c * d);
  }
  // a*b + c
  static __device__ __forceinline__ float madd1(float a, float b, float c) {
    if constexpr (FMA) return __builtin_fmaf(a, b, c);
    else return (a * b) + c;
  }
  // glm::cross
  static __device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return {msub2(a.y, b.z, b.y, a.z), msub2(a.z, b.x, b.z, a.x), msub2(a.x, b.y, b.x, a.y)};
  }
  // glm::dot(vec3): (a.x*b.x + a.y*b.y) + a.z*b.z
  static __device__ __forceinline__ float dot(V3 a, V3 b) {
    if constexpr (FMA) return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.x, b.x, a.y * b.y));
    else return ((a.x * b.x) + (a.y * b.y)) + (a.z * b.z);
  }
  // glm::normalize: v * (1 / sqrt(dot(v, v))), both operations correctly rounded.  When every
  // active lane's squared length lies in [2^-96, 2^96] (wave-uniform test; always, in practice) the
  // scaling and special-case steps of the generic expansions are skipped: same bits, 13 fewer
  // VALU instructions per ray (device-checked over every float of the range, rt_dbg_check_midrange).
  static __device__ __forceinline__ V3 normalize(V3 a) {
    const float dd = dot(a, a);
    float inv;
#ifndef RT_NO_MIDRANGE
    if (__builtin_amdgcn_ballot_w64(!midrange(dd)) == 0ull) inv = rcp_midrange(sqrt_midrange(dd));
    else
#endif
      inv = 1.0f / __builtin_sqrtf(dd);
    return {a.x * inv, a.y * inv, a.z * inv};
  }
  // glm mat4 * vec4, xyz: (m0*x + m1*y) + (m2*z + m3*w); M holds the 4 columns' xyz
  static __device__ __forceinline__ V3 mat_mul_point(const float* M, float x, float y, float z, float w) {
    return {madd2(M[0], x, M[3], y) + madd2(M[6], z, M[9], w),
            madd2(M[1], x, M[4], y) + madd2(M[7], z, M[10], w),
            madd2(M[2], x, M[5], y) + madd2(M[8], z, M[11], w)};
  }
};

// ---- build-owned sincos (replaces libdevice sinf/cosf; same constants and operation
// order as the documented spec, every step an explicit fma) ------------------------------
__host__ __device__ __forceinline__ void sincos_spec(float x, float& s, float& c) {
  const float k = __builtin_rintf(x * 0.63661977236758134308f);
  float r = __builtin_fmaf(k, -1.5703125f, x);
  r = __builtin_fmaf(k, -4.837512969970703125e-4f, r);
  r = __builtin_fmaf(k, -7.54978995489188216e-8f, r);
  const float z = r * r;
  const float ps = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z,
                                  -1.6666654611e-1f);
  const float sn = __builtin_fmaf(r * z, ps, r);
  const float pc = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z,
                                  4.166664568298827e-2f);
  const float cs = __builtin_fmaf(z * z, pc, __builtin_fmaf(-0.5f, z, 1.0f));
  // quadrant q = k mod 4: (sin, cos) = (sn, cs), (cs, -sn), (-sn, -cs), (-cs, sn); branch-free
  const int q = static_cast<int>(k) & 3;
  const float ss = (q & 1) ? cs : sn;
  const float cc = (q & 1) ? sn : cs;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
}

// ---- XORWOW (cuRAND's curandState_t generator restated; Random.cuh:15-16,23) ----------
struct Rng { uint32_t d, v0, v1, v2, v3, v4; };

__device__ __forceinline__ uint32_t rng_next(Rng& s) {
  const uint32_t t = s.v0 ^ (s.v0 >> 2);
  s.v0 = s.v1; s.v1 = s.v2; s.v2 = s.v3; s.v3 = s.v4;
  s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
  s.d += 362437u;
  return s.v4 + s.d;
}

// `draws` calls of rng_next whose values nobody needs (the samples of a tile with a certain winner): five draws are one full
// rotation of the five xorshift words, so the loop body updates them in place -- no register moves -- and the Weyl counter
// advances once.
__device__ __forceinline__ void rng_discard(Rng& s, uint32_t draws) {
  auto f = [](uint32_t x, uint32_t v) -> uint32_t { const uint32_t t = x ^ (x >> 2); return (v ^ (v << 4)) ^ (t ^ (t << 1)); };
  uint32_t i = 0;
  for (; i + 5u <= draws; i += 5u) {
    s.v0 = f(s.v0, s.v4); s.v1 = f(s.v1, s.v0); s.v2 = f(s.v2, s.v1); s.v3 = f(s.v3, s.v2); s.v4 = f(s.v4, s.v3);
  }
  for (; i < draws; ++i) {
    const uint32_t nv = f(s.v0, s.v4);
    s.v0 = s.v1; s.v1 = s.v2; s.v2 = s.v3; s.v3 = s.v4; s.v4 = nv;
  }
  s.d += 362437u * draws;
}

// curand_uniform: x * 2^-32 + 2^-33, (0, 1]
__device__ __forceinline__ float rng_uniform(Rng& s) {
  // x * 2^-32 is exact, so the single fma rounds exactly like the reference's mul + add
  return __builtin_fmaf(static_cast<float>(rng_next(s)), 2.3283064e-10f, 2.3283064e-10f / 2.0f);
}

// random::UnifromOnDisk, Random.cuh:13-19 (the mistyped pi literal is the reference's)
__device__ __forceinline__ void uniform_on_disk(Rng& s, float& dx, float& dy) {
  const float t = (2.0f * 3.14156545f) * rng_uniform(s);
  const float u1 = rng_uniform(s);
  const float u2 = rng_uniform(s);
  const float u = u1 + u2;
  const float sr = (u > 1.0f) ? 2.0f - u : u;
  float sn, cs;
  sincos_spec(t, sn, cs);
  dx = sr * cs;
  dy = sr * sn;
}

// Corrected form of the reference's experimental normal packing (UnitTests/NormalPackingTest.cpp:10-23,
// Documentation/gpu.meshes.txt:20-33): n_k = floor(fract(packed * 256^k) * 256) / 127 - 1
__device__ __forceinline__ V3 unpack_normal(float packed) {
  auto field = [packed](float shift) {
    const float s = packed * shift;
    return __builtin_floorf((s - __builtin_floorf(s)) * 256.0f) / 127.0f - 1.0f;
  };
  return {field(1.0f), field(256.0f), field(65536.0f)};
}

// float -> rt::Channel as the GPU conversion does it: truncate toward zero, clamp, NaN -> 0
__device__ __forceinline__ uint32_t to_channel(float f) {
  if (!(f > 0.0f)) return 0u;
  if (f >= 255.0f) return 255u;
  return static_cast<uint32_t>(f);
}

// utils::GetColor, DeviceUtils.cuh:20-23: b | g<<8 | r<<16 | a<<24
__device__ __forceinline__ uint32_t pack_color(float r, float g, float b) {
  return (to_channel(b) << 0) | (to_channel(g) << 8) | (to_channel(r) << 16) | (255u << 24);
}

}  // namespace rtd
