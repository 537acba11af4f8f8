/*
 * oracle.h -- CPU restatement of the RayTracerTest per-pixel trace path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load or call it.  The shipped library
 * (raytracertest_amd/csrc) neither includes, links nor calls anything in oracle/.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference checkout).  Pinning status:
 *   - HitTriangle / Ray / normalize(cross)  : pinned bit-exactly by the reference's 8
 *     UnitTests/TriangleHitTest.cpp cases (tests/golden/triangle_hit_kats.json).
 *   - everything else (ray generation, lens sampling, RNG stream, farthest-hit select,
 *     shading, accumulation, BGRA conversion) : PARITY UNPINNED by the reference -- it
 *     ships no golden image, no recorded RGBA and seeds from time(nullptr)
 *     (RayTracer/Random.cu:45).  The reference cannot be compiled here (needs nvcc,
 *     cuRAND, glm, gtest; none present), so the restatement is reviewed line by line.
 *
 * Third-party arithmetic that is not under the reference tree is restated from its
 * published algorithm:
 *   - cuRAND XORWOW (CUDA Toolkit 11.7, RayTracer/RayTracer.vcxproj:32): seeding
 *     scramble, xorshift/Weyl step, 2^67-step subsequence jump, curand_uniform mapping.
 *   - glm 0.9.9.x: cross/dot/normalize/angleAxis/quat product/mat4_cast/mat4*vec4.
 *   - libdevice sinf/cosf/tanf: replaced by a build-owned polynomial sincos (documented
 *     in DESIGN.md "numeric spec"); tan(fov/2) = sin/cos of that, hoisted per launch.
 *
 * Two arithmetic modes, both IEEE-754 binary32 with correctly rounded / and sqrt:
 *   contract = 0 ("strict")  every source-level * + - rounds separately (what a host
 *                            compile of the reference source without FMA computes);
 *   contract = 1 ("fma")     the a*b+c shapes listed in DESIGN.md are single fused
 *                            multiply-adds (what the reference's GPU build does under
 *                            nvcc's default -fmad=true, RayTracer.vcxproj:65-70; the exact
 *                            fusion pattern nvcc picks is unknowable, ours is documented).
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- RNG: cuRAND XORWOW restated (Random.cu:22-27, Random.cuh:15-16,23) ---------- */
/* state layout used everywhere in this repo: s[0] = d (Weyl), s[1..5] = v[0..4]     */
void     orc_rng_seed(uint64_t seed, uint32_t s[6]);                 /* curand_init scratch part   */
void     orc_rng_init(uint64_t seed, uint64_t subsequence, uint32_t s[6]); /* curand_init(seed,sub,0) */
void     orc_rng_skip_subsequences(uint32_t s[6], uint64_t n);       /* raw state += n * 2^67 outputs */
uint32_t orc_rng_next(uint32_t s[6]);                                /* curand()                   */
float    orc_rng_uniform(uint32_t s[6]);                             /* curand_uniform() in (0,1]  */
void     orc_rng_jump_columns(uint32_t cols[160 * 5]);               /* T^(2^67) as 160 columns    */
void     orc_rng_step_linear_n(uint32_t v[5], uint64_t n);           /* v <- T^n v by stepping     */
void     orc_rng_matpow_apply(uint64_t n, uint32_t v[5]);            /* v <- T^n v by matrix power */

/* ---- build-owned transcendental spec ------------------------------------------- */
void  orc_sincos(float x, float* s, float* c);
float orc_tan_half(float fov_rad);            /* sin(fov/2)/cos(fov/2) */

/* ---- geometry (Kernels.cuh:29-65, Ray.cuh:12-44) -------------------------------- */
/* ray[6] = origin xyz, direction xyz.  eps_mode 0: kernel epsilon 1e-10f
 * (Kernels.cuh:42); 1: unit-test epsilon FLT_EPSILON (TriangleHitTest.cpp:70).     */
void orc_ray_make(const float o[3], const float d[3], int normalize, int contract, float ray[6]);
void orc_ray_point(const float ray[6], float t, int contract, float p[3]);
int  orc_hit_triangle(const float ray[6], const float v0[3], const float v1[3],
                      const float v2[3], int contract, int eps_mode,
                      float* t, float* u, float* v);
void orc_triangle_normal(const float a[3], const float b[3], const float c[3],
                         int contract, float n[3]);  /* normalize(cross(b-a, c-a)) */
int  orc_hit_sphere(const float ray[6], const float sph[4], int contract, float* t);

/* ---- camera (ThinLensCamera.cuh:16-28,30-52,111-141) ---------------------------- */
typedef struct orc_camera {
  float angles[2];   /* radians (mRotationAngles) */
  float fov;         /* radians (mFov)            */
  float focal;       /* mFocalLength              */
  float aperture;    /* mAperture                 */
  float M[16];       /* column-major mat4, M[col*4+row] (mCameraTransformation) */
} orc_camera;
void orc_camera_init(orc_camera* cam, const float angles[2], float fov_deg,
                     float focal, float aperture);
void orc_camera_rotate(orc_camera* cam, const float dangles[2]);
void orc_camera_set(orc_camera* cam, float fov_deg, float focal, float aperture);
void orc_camera_pinhole(const orc_camera* cam, uint32_t px, uint32_t py, uint32_t W,
                        uint32_t H, int contract, float ray[6]);
void orc_camera_get_ray(const orc_camera* cam, uint32_t px, uint32_t py, uint32_t W,
                        uint32_t H, uint32_t rng[6], int contract, float ray[6]);
void orc_uniform_on_disk(uint32_t rng[6], float xy[2]);   /* Random.cuh:13-19 */

/* ---- scene + frame --------------------------------------------------------------- */
typedef struct orc_scene {
  const float* tris;     /* n_tris * 12 floats: three float4 absolute vertices, .w ignored */
  uint32_t     n_tris;
  const float* spheres;  /* n_spheres * 4 floats: centre xyz, radius (build-defined)      */
  uint32_t     n_spheres;
  int32_t      hit_mode;  /* 0: reference rule (farthest hit, negative t accepted, Kernels.cuh:73,84);
                             1: nearest hit with t > 0 (build-defined extension) */
  int32_t      layout;    /* 0: absolute vertices (RayTracerImpl.cu:119-177);
                             1: (v0, e0 = v1-v0, e1 = v2-v0) rows with a packed vertex normal in each .w
                                (Documentation/gpu.meshes.txt:16-34; build-defined) */
  int32_t      shade_mode; /* 0: |face normal| (Kernels.cuh:97-99);
                              1: |interpolated vertex normal|, layout 1 only (build-defined) */
} orc_scene;

/* Corrected form of the reference's experimental packing (UnitTests/NormalPackingTest.cpp:10-23):
 * byte_k = floor(n_k*127 + 127.5); packed = byte_0/2^8 + byte_1/2^16 + byte_2/2^24;
 * unpack: n_k = floor(fract(packed * 256^k) * 256) / 127 - 1.  PARITY UNPINNED (the reference's
 * own test of its version cannot pass). */
float orc_pack_normal(const float n[3]);
void  orc_unpack_normal(float packed, float n[3]);

/* A frame is a row band [row0, row0+rows) of a W x H image.  Buffers hold the band only:
 * render rows*W*4 floats (RGBA, alpha never written, Kernels.cuh:141-144), counts rows*W,
 * rng rows*W*6 (array of {d,v0..v4}), image rows*W BGRA8-in-u32 (Common/Color.h:21-24). */
typedef struct orc_frame {
  uint32_t W, H, row0, rows;
  float*    render;
  uint32_t* counts;
  uint32_t* rng;
  uint32_t* image;
} orc_frame;

/* Probe of a ray family (test infrastructure of the product's conservative classification): the reference's HitTriangle
 * arithmetic (Kernels.cuh:39-63) and farthest-hit scan (:73-92) for the rays of n_pixels pixels x n_lens lens samples
 * (unit-disk points as Random.cuh:13-19 returns them; rays built as ThinLensCamera.cuh:41-51) against every triangle.
 * out[n_tris] must be initialised with orc_probe_init; several calls accumulate. */
typedef struct orc_probe_tri {
  uint32_t rays, hits, wins, nan_hits;      /* rays probed; HitTriangle returned true; Radiance's winner; hits with t = NaN */
  uint32_t nan_rays, form_rejects, form_wrong, pad;   /* rays whose det, U or V is NaN; rays the per-sample forms skip; of those, hit */
  double det_min, det_max, U_min, U_max, V_min, V_max;   /* over the rays: det |w|, U |w|, V |w| (w = focal point - lens point) */
  double q_min, q_max;                      /* over the hit rays: t / |w| */
  double S_min, S_max;                      /* over the rays: (det - U - V) |w|, det |w| times the third barycentric coordinate */
} orc_probe_tri;
void orc_probe_init(orc_probe_tri* out, uint32_t n_tris);
void orc_tile_probe(const orc_scene* sc, const orc_camera* cam, uint32_t W, uint32_t H, const uint32_t* pixels,
                    uint32_t n_pixels, const float* lens, uint32_t n_lens, int contract, const float* forms,
                    const float* fc, orc_probe_tri* out, uint32_t* no_hit_rays);
void orc_radiance(const orc_scene* sc, const float ray[6], int contract, float rgb[3]);
void orc_normalize(const float v[3], int contract, float out[3]);        /* glm::normalize */
void orc_frame_rng_init(orc_frame* f, uint64_t seed, int nthreads);      /* Random.cu:10-52   */
void orc_frame_clear(orc_frame* f);                                      /* RayTracerImpl.cu:242-243 */
void orc_trace_launch(const orc_scene* sc, const orc_camera* cam, orc_frame* f,
                      uint32_t sample_count, int contract, int nthreads); /* Kernels.cuh:110-147 */
void orc_convert(orc_frame* f);                                          /* Kernels.cuh:149-169 */
uint32_t orc_pack_color(float r, float g, float b);                      /* DeviceUtils.cuh:20-23 + Q5 clamp */

#ifdef __cplusplus
}
#endif
#endif
