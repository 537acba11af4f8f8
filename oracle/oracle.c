/*
 * oracle.c -- CPU restatement of the RayTracerTest per-pixel trace path (plain C11).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE -- see the header of oracle.h for who may use
 * it, what in it is pinned by the reference's own vectors and what is "parity unpinned".
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math, no intrinsics).
 */
#define _GNU_SOURCE
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__) && !defined(ORC_NO_CLONES)
/* one clone with hardware FMA (fmaf inlines to vfmadd), one generic (libm fmaf, exact too) */
#define ORC_CLONES __attribute__((target_clones("fma", "default")))
#else
#define ORC_CLONES
#endif
#define ORC_INL static inline __attribute__((always_inline))

typedef struct { float x, y, z; } v3;
typedef struct { v3 o, d; } ray_t;

static inline v3 sub3(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline v3 add3(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static inline float absf(float f) { return (f < 0.0f) ? -f : f; }   /* Kernels.cuh:16-19 */
static inline v3 ld3(const float* p) { v3 r = {p[0], p[1], p[2]}; return r; }
static inline void st3(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

/* =====================================================================================
 * RNG -- cuRAND XORWOW (curandState_t default generator), restated from the published
 * curand_kernel.h algorithm.  PARITY UNPINNED: no cuRAND here, no reference vectors.
 * ===================================================================================== */

/* curand(): xorshift on v[0..4] + Weyl sequence d */
static inline uint32_t xorwow_next(uint32_t s[6]) {
  uint32_t* v = s + 1;
  const uint32_t t = v[0] ^ (v[0] >> 2);
  v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
  v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
  s[0] += 362437u;
  return v[4] + s[0];
}
uint32_t orc_rng_next(uint32_t s[6]) { return xorwow_next(s); }

/* curand_uniform(): x * 2^-32 + 2^-33, in (0, 1] */
#define RNG_UNIFORM_OF(x) ((float)(x) * 2.3283064e-10f + (2.3283064e-10f / 2.0f))
float orc_rng_uniform(uint32_t s[6]) { return RNG_UNIFORM_OF(xorwow_next(s)); }

/* curand_init() scratch part: split the 64-bit seed, salt, multiply, offset Marsaglia's
 * constants */
void orc_rng_seed(uint64_t seed, uint32_t s[6]) {
  const uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u;
  const uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
  const uint32_t t0 = 1099087573u * s0;
  const uint32_t t1 = 2591861531u * s1;
  s[0] = 6615241u + t1 + t0;
  s[1] = 123456789u + t0;
  s[2] = 362436069u ^ t0;
  s[3] = 521288629u + t1;
  s[4] = 88675123u ^ t1;
  s[5] = 5783321u + t0;
}

/* The xorshift part is linear over GF(2) on the 160 bits of v[0..4].  A matrix is stored
 * as 160 columns of 5 words: column i is the image of basis vector e_i (bit i%32 of
 * word i/32).  curand_init(seed, subsequence, 0) advances v by subsequence * 2^67 steps
 * (d is unchanged: 362437 * 2^67 = 0 mod 2^32), i.e. v <- (T^(2^67))^subsequence v. */
typedef struct { uint32_t col[160][5]; } gf2mat;

static void step_v(uint32_t v[5]) {
  const uint32_t t = v[0] ^ (v[0] >> 2);
  v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
  v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
}

static void mat_vec(const gf2mat* A, const uint32_t x[5], uint32_t out[5]) {
  uint32_t r[5] = {0, 0, 0, 0, 0};
  for (int w = 0; w < 5; ++w) {
    uint32_t bits = x[w];
    while (bits) {
      const int b = __builtin_ctz(bits);
      bits &= bits - 1;
      const uint32_t* c = A->col[w * 32 + b];
      r[0] ^= c[0]; r[1] ^= c[1]; r[2] ^= c[2]; r[3] ^= c[3]; r[4] ^= c[4];
    }
  }
  memcpy(out, r, sizeof r);
}

static void mat_square(const gf2mat* A, gf2mat* out) {
  gf2mat tmp;
  for (int i = 0; i < 160; ++i) mat_vec(A, A->col[i], tmp.col[i]);
  *out = tmp;
}

static void mat_step(gf2mat* T) {
  for (int i = 0; i < 160; ++i) {
    uint32_t v[5] = {0, 0, 0, 0, 0};
    v[i / 32] = 1u << (i % 32);
    step_v(v);
    memcpy(T->col[i], v, sizeof v);
  }
}

static gf2mat g_jump[64];              /* g_jump[k] = (T^(2^67))^(2^k) */
static pthread_once_t g_jump_once = PTHREAD_ONCE_INIT;
static void jump_build(void) {
  gf2mat m;
  mat_step(&m);
  for (int i = 0; i < 67; ++i) mat_square(&m, &m);
  g_jump[0] = m;
  for (int k = 1; k < 64; ++k) mat_square(&g_jump[k - 1], &g_jump[k]);
}

void orc_rng_jump_columns(uint32_t cols[160 * 5]) {
  pthread_once(&g_jump_once, jump_build);
  memcpy(cols, g_jump[0].col, sizeof g_jump[0].col);
}

void orc_rng_init(uint64_t seed, uint64_t subsequence, uint32_t s[6]) {
  pthread_once(&g_jump_once, jump_build);
  orc_rng_seed(seed, s);
  for (int k = 0; k < 64 && (subsequence >> k); ++k)
    if ((subsequence >> k) & 1u) mat_vec(&g_jump[k], s + 1, s + 1);
}

/* advance a raw state {d, v0..v4} by `n` subsequences of 2^67 outputs (the skip-ahead of curand_init) */
void orc_rng_skip_subsequences(uint32_t s[6], uint64_t n) {
  pthread_once(&g_jump_once, jump_build);
  for (int k = 0; k < 64 && (n >> k); ++k)
    if ((n >> k) & 1u) mat_vec(&g_jump[k], s + 1, s + 1);
}

void orc_rng_step_linear_n(uint32_t v[5], uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) step_v(v);
}

void orc_rng_matpow_apply(uint64_t n, uint32_t v[5]) {
  gf2mat m;
  mat_step(&m);
  for (int k = 0; k < 64 && (n >> k); ++k) {
    if ((n >> k) & 1u) mat_vec(&m, v, v);
    mat_square(&m, &m);
  }
}

/* =====================================================================================
 * Build-owned sincos (replaces libdevice sinf/cosf/tanf; DESIGN.md "numeric spec").
 * Cody-Waite reduction by pi/2 with three constants, degree-7/8 minimax polynomials on
 * [-pi/4, pi/4], every step an explicit fmaf so that any IEEE-754 machine agrees bitwise.
 * ===================================================================================== */
ORC_INL void sincos_spec(float x, float* s, float* c) {
  const float k = rintf(x * 0.63661977236758134308f);
  float r = __builtin_fmaf(k, -1.5703125f, x);
  r = __builtin_fmaf(k, -4.837512969970703125e-4f, r);
  r = __builtin_fmaf(k, -7.54978995489188216e-8f, r);
  const float z = r * r;
  const float ps = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z,
                                  -1.6666654611e-1f);
  const float sn = __builtin_fmaf(r * z, ps, r);
  const float pc = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f),
                                  z, 4.166664568298827e-2f);
  const float cs = __builtin_fmaf(z * z, pc, __builtin_fmaf(-0.5f, z, 1.0f));
  const int q = (int)k & 3;
  *s = (q == 0) ? sn : (q == 1) ? cs : (q == 2) ? -sn : -cs;
  *c = (q == 0) ? cs : (q == 1) ? -sn : (q == 2) ? -cs : sn;
}

void orc_sincos(float x, float* s, float* c) { sincos_spec(x, s, c); }

float orc_tan_half(float fov_rad) {
  float s, c;
  sincos_spec(fov_rad / 2.0f, &s, &c);          /* ThinLensCamera.cuh:114 */
  return s / c;
}

/* random::UnifromOnDisk, Random.cuh:13-19 (pi is mistyped in the reference: Q2) */
ORC_INL float rng_uniform(uint32_t s[6]) { return RNG_UNIFORM_OF(xorwow_next(s)); }

ORC_INL void uniform_on_disk(uint32_t* rng, float xy[2]) {
  const float t = (2.0f * 3.14156545f) * rng_uniform(rng);           /* :15 */
  const float u1 = rng_uniform(rng);
  const float u2 = rng_uniform(rng);
  const float u = u1 + u2;                                           /* :16 */
  const float sr = (u > 1.0f) ? 2.0f - u : u;                        /* :17 */
  float sn, cs;
  sincos_spec(t, &sn, &cs);
  xy[0] = sr * cs;                                                   /* :18 */
  xy[1] = sr * sn;
}
void orc_uniform_on_disk(uint32_t rng[6], float xy[2]) { uniform_on_disk(rng, xy); }

/* =====================================================================================
 * Camera host side, ThinLensCamera.cuh:16-28,104-108,132-141 (host code: never fused)
 * ===================================================================================== */
static void camera_transform(orc_camera* cam) {
  float sx, cx, sy, cy;
  orc_sincos(cam->angles[0] * 0.5f, &sx, &cx);   /* glm::angleAxis(a, axis): (cos(a/2), axis*sin(a/2)) */
  orc_sincos(cam->angles[1] * 0.5f, &sy, &cy);
  const float qXw = cx, qXx = 1.0f * sx, qXy = 0.0f * sx, qXz = 0.0f * sx;   /* :138 */
  const float qYw = cy, qYx = 0.0f * sy, qYy = 1.0f * sy, qYz = 0.0f * sy;   /* :139 */
  /* glm quat product p*q with p = qY, q = qX (:140) */
  const float w = qYw * qXw - qYx * qXx - qYy * qXy - qYz * qXz;
  const float x = qYw * qXx + qYx * qXw + qYy * qXz - qYz * qXy;
  const float y = qYw * qXy + qYy * qXw + qYz * qXx - qYx * qXz;
  const float z = qYw * qXz + qYz * qXw + qYx * qXy - qYy * qXx;
  /* glm::mat4_cast */
  const float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z;
  const float qwx = w * x, qwy = w * y, qwz = w * z;
  float* M = cam->M;
  memset(M, 0, 16 * sizeof(float));
  M[0] = 1.0f - 2.0f * (qyy + qzz); M[1] = 2.0f * (qxy + qwz);        M[2] = 2.0f * (qxz - qwy);
  M[4] = 2.0f * (qxy - qwz);        M[5] = 1.0f - 2.0f * (qxx + qzz); M[6] = 2.0f * (qyz + qwx);
  M[8] = 2.0f * (qxz + qwy);        M[9] = 2.0f * (qyz - qwx);        M[10] = 1.0f - 2.0f * (qxx + qyy);
  M[15] = 1.0f;
}

static float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

void orc_camera_init(orc_camera* cam, const float angles[2], float fov_deg, float focal,
                     float aperture) {
  cam->angles[0] = angles[0]; cam->angles[1] = angles[1];
  cam->fov = radians(fov_deg);                                       /* :23 */
  cam->focal = focal;
  cam->aperture = aperture;
  camera_transform(cam);                                             /* :27 */
}

void orc_camera_rotate(orc_camera* cam, const float d[2]) {          /* :104-108 */
  cam->angles[0] += d[0]; cam->angles[1] += d[1];
  camera_transform(cam);
}

void orc_camera_set(orc_camera* cam, float fov_deg, float focal, float aperture) {
  cam->fov = radians(fov_deg);                                       /* :79-82 */
  cam->focal = focal;                                                /* :99-102 */
  cam->aperture = aperture;                                          /* :89-92 */
}

/* =====================================================================================
 * Mode-dependent arithmetic, instantiated twice
 * ===================================================================================== */
float orc_pack_normal(const float n[3]) {
  return floorf(n[0] * 127.0f + 127.5f) / 256.0f + floorf(n[1] * 127.0f + 127.5f) / 65536.0f +
         floorf(n[2] * 127.0f + 127.5f) / 16777216.0f;
}

void orc_unpack_normal(float packed, float n[3]) {
  float shift = 1.0f;
  for (int k = 0; k < 3; ++k, shift *= 256.0f) {
    const float s = packed * shift;
    n[k] = floorf((s - floorf(s)) * 256.0f) / 127.0f - 1.0f;
  }
}

#define ORC_FN(n) n##_strict
#define ORC_CONTRACT 0
#include "oracle_core.inc"
#undef ORC_FN
#undef ORC_CONTRACT

#define ORC_FN(n) n##_fma
#define ORC_CONTRACT 1
#include "oracle_core.inc"
#undef ORC_FN
#undef ORC_CONTRACT

/* ---- exported single-item entry points ------------------------------------------ */
void orc_ray_make(const float o[3], const float d[3], int normalize, int contract, float out[6]) {
  ray_t r;
  if (normalize) r = contract ? ray_fma(ld3(o), ld3(d)) : ray_strict(ld3(o), ld3(d));
  else { r.o = ld3(o); r.d = ld3(d); }
  st3(out, r.o); st3(out + 3, r.d);
}

static ray_t ld_ray(const float* p) { ray_t r; r.o = ld3(p); r.d = ld3(p + 3); return r; }

void orc_ray_point(const float ray[6], float t, int contract, float p[3]) {
  st3(p, contract ? point_fma(ld_ray(ray), t) : point_strict(ld_ray(ray), t));
}

int orc_hit_triangle(const float ray[6], const float v0[3], const float v1[3], const float v2[3],
                     int contract, int eps_mode, float* t, float* u, float* v) {
  const float eps = eps_mode ? FLT_EPSILON : 0.0000000001f;
  return contract ? hit_triangle_fma(ld_ray(ray), ld3(v0), ld3(v1), ld3(v2), eps, t, u, v)
                  : hit_triangle_strict(ld_ray(ray), ld3(v0), ld3(v1), ld3(v2), eps, t, u, v);
}

void orc_triangle_normal(const float a[3], const float b[3], const float c[3], int contract,
                         float n[3]) {
  const v3 e1 = sub3(ld3(b), ld3(a)), e2 = sub3(ld3(c), ld3(a));
  st3(n, contract ? normalize_fma(cross_fma(e1, e2)) : normalize_strict(cross_strict(e1, e2)));
}

int orc_hit_sphere(const float ray[6], const float sph[4], int contract, float* t) {
  return contract ? hit_sphere_fma(ld_ray(ray), sph, t) : hit_sphere_strict(ld_ray(ray), sph, t);
}

void orc_camera_pinhole(const orc_camera* cam, uint32_t px, uint32_t py, uint32_t W, uint32_t H,
                        int contract, float out[6]) {
  const float hh = orc_tan_half(cam->fov);
  const ray_t r = contract ? pinhole_fma(cam, hh, px, py, W, H) : pinhole_strict(cam, hh, px, py, W, H);
  st3(out, r.o); st3(out + 3, r.d);
}

void orc_camera_get_ray(const orc_camera* cam, uint32_t px, uint32_t py, uint32_t W, uint32_t H,
                        uint32_t rng[6], int contract, float out[6]) {
  const float hh = orc_tan_half(cam->fov);
  ray_t r;
  if (contract) r = get_ray_fma(cam, pinhole_fma(cam, hh, px, py, W, H), rng);
  else r = get_ray_strict(cam, pinhole_strict(cam, hh, px, py, W, H), rng);
  st3(out, r.o); st3(out + 3, r.d);
}

void orc_normalize(const float v[3], int contract, float out[3]) {
  const v3 a = {v[0], v[1], v[2]};
  const v3 n = contract ? normalize_fma(a) : normalize_strict(a);
  out[0] = n.x; out[1] = n.y; out[2] = n.z;
}

void orc_radiance(const orc_scene* sc, const float ray[6], int contract, float rgb[3]) {
  st3(rgb, contract ? radiance_fma(sc, ld_ray(ray)) : radiance_strict(sc, ld_ray(ray)));
}

void orc_probe_init(orc_probe_tri* out, uint32_t n_tris) {
  for (uint32_t i = 0; i < n_tris; ++i) {
    memset(out + i, 0, sizeof out[i]);
    out[i].det_min = out[i].U_min = out[i].V_min = out[i].q_min = out[i].S_min = INFINITY;
    out[i].det_max = out[i].U_max = out[i].V_max = out[i].q_max = out[i].S_max = -INFINITY;
  }
}

void orc_tile_probe(const orc_scene* sc, const orc_camera* cam, uint32_t W, uint32_t H, const uint32_t* pixels,
                    uint32_t n_pixels, const float* lens, uint32_t n_lens, int contract, const float* forms,
                    const float* fc, orc_probe_tri* out, uint32_t* no_hit_rays) {
  if (contract) tile_probe_fma(sc, cam, W, H, pixels, n_pixels, lens, n_lens, forms, fc, out, no_hit_rays);
  else tile_probe_strict(sc, cam, W, H, pixels, n_pixels, lens, n_lens, forms, fc, out, no_hit_rays);
}

/* =====================================================================================
 * Frame-level: RNG state creation, clear, one TraceKernel launch, conversion
 * ===================================================================================== */
typedef struct {
  const orc_scene* sc; const orc_camera* cam; orc_frame* f;
  uint32_t samples; int contract; uint32_t r0, r1; uint64_t seed; int job;
} work_t;

static void rng_rows(orc_frame* f, uint64_t seed, uint32_t r0, uint32_t r1) {
  /* random::InitRandomStates, Random.cu:10-30: curand_init(seed, x + y*W, 0).  Along a
   * row subsequence p+1 is one more jump: v_{p+1} = T^(2^67) v_p (same linear map). */
  for (uint32_t ly = r0; ly < r1; ++ly) {
    const uint64_t p0 = (uint64_t)(f->row0 + ly) * f->W;
    uint32_t s[6];
    orc_rng_init(seed, p0, s);
    for (uint32_t x = 0; x < f->W; ++x) {
      memcpy(f->rng + 6 * ((size_t)ly * f->W + x), s, sizeof s);
      mat_vec(&g_jump[0], s + 1, s + 1);
    }
  }
}

static void* worker(void* arg) {
  work_t* w = (work_t*)arg;
  if (w->job == 0) rng_rows(w->f, w->seed, w->r0, w->r1);
  else if (w->contract) trace_rows_fma(w->sc, w->cam, w->f, w->samples, w->r0, w->r1);
  else trace_rows_strict(w->sc, w->cam, w->f, w->samples, w->r0, w->r1);
  return NULL;
}

static void run_rows(work_t proto, int nthreads) {
  const uint32_t rows = proto.f->rows;
  if (nthreads < 1) nthreads = 1;
  if ((uint32_t)nthreads > rows) nthreads = (int)rows;
  if (nthreads <= 1) { proto.r0 = 0; proto.r1 = rows; worker(&proto); return; }
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
  work_t* ws = (work_t*)malloc(sizeof(work_t) * (size_t)nthreads);
  /* interleave small row blocks so threads finish together; each row is independent */
  for (int i = 0; i < nthreads; ++i) {
    ws[i] = proto;
    ws[i].r0 = (uint32_t)(((uint64_t)rows * (uint64_t)i) / (uint64_t)nthreads);
    ws[i].r1 = (uint32_t)(((uint64_t)rows * (uint64_t)(i + 1)) / (uint64_t)nthreads);
    pthread_create(&th[i], NULL, worker, &ws[i]);
  }
  for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
  free(th); free(ws);
}

void orc_frame_rng_init(orc_frame* f, uint64_t seed, int nthreads) {
  pthread_once(&g_jump_once, jump_build);
  work_t w; memset(&w, 0, sizeof w);
  w.f = f; w.seed = seed; w.job = 0;
  run_rows(w, nthreads);
}

void orc_frame_clear(orc_frame* f) {       /* ClearRenderBuffer / ClearSampleCountBuffer */
  const size_t n = (size_t)f->rows * f->W;
  memset(f->render, 0, n * 4 * sizeof(float));
  memset(f->counts, 0, n * sizeof(uint32_t));
}

void orc_trace_launch(const orc_scene* sc, const orc_camera* cam, orc_frame* f,
                      uint32_t sample_count, int contract, int nthreads) {
  work_t w; memset(&w, 0, sizeof w);
  w.sc = sc; w.cam = cam; w.f = f; w.samples = sample_count; w.contract = contract; w.job = 1;
  run_rows(w, nthreads);
}

/* utils::GetColor + the float -> uint8 conversion of its arguments, DeviceUtils.cuh:20-23,
 * Kernels.cuh:165-168.  Out-of-range float -> uint8 is UB in C++; the GPU conversion
 * truncates toward zero and clamps (negative and NaN -> 0): Q5.  Made explicit here. */
static uint32_t to_channel(float f) {
  if (!(f > 0.0f)) return 0u;             /* negatives, -0, NaN */
  if (f >= 255.0f) return 255u;
  return (uint32_t)f;                     /* truncation toward zero */
}

uint32_t orc_pack_color(float r, float g, float b) {
  return (to_channel(b) << 0) | (to_channel(g) << 8) | (to_channel(r) << 16) | (255u << 24);
}

void orc_convert(orc_frame* f) {
  const size_t n = (size_t)f->rows * f->W;
  for (size_t p = 0; p < n; ++p) {
    const float cnt = (float)f->counts[p];                             /* Kernels.cuh:164 */
    f->image[p] = orc_pack_color(255.0f * (f->render[4 * p + 0] / cnt),
                                 255.0f * (f->render[4 * p + 1] / cnt),
                                 255.0f * (f->render[4 * p + 2] / cnt));
  }
}
