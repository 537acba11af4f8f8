// rt_tracer.hip -- host runtime behind the C ABI of include/rt_mi355x.h.
//
// Mirrors rt::RayTracerImpl (RayTracer/RayTracerImpl.cuh:17-75, RayTracerImpl.cu): owns the
// device buffers, the RNG states, the scene, the camera, the render std::thread, launches
// the kernels and fires the callbacks.  HIP streams/events, pinned host image for the
// callbacks (the PBO interop is cut), no CPU fallback: without a HIP device creation fails.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "rt_device_math.hpp"
#include "rt_kernels.hpp"
#include "rt_rng_host.hpp"

namespace {

std::mutex g_err_mu;
std::string g_last_error;

// The environment switches of the library, read ONCE per process: every one of them is exercised by
// a test (tests/test_gpu_*.py) -- A/B knobs of past rounds are gone, their measurements are in HISTORY.md.
struct Env {
  bool log;             // RT_MI355X_LOG=1: recorded failures and launch-shape changes go to stderr
  bool no_split;        // RT_MI355X_NO_SPLIT=1: launches as one kernel on one stream
  bool no_pretest;      // RT_MI355X_NO_PRETEST=1: no per-sample forms in the dense-scene kernels
  bool no_sure_table;   // RT_MI355X_NO_SURE_TABLE=1: certain-winner tiles add their samples' colours per pixel
  int row_interleave;   // RT_MI355X_ROW_INTERLEAVE=0|1: pins the halves of a split small-scene launch (-1: by the builder's counts)
  long macro_cap;       // RT_MI355X_MACRO_CAP=n: capacity of the macro lists (tests: forces the overflow fallback); 0 = default
  static bool on(const char* name) { const char* e = getenv(name); return e && e[0] == '1'; }
  Env() {
    log = getenv("RT_MI355X_LOG") != nullptr;
    no_split = on("RT_MI355X_NO_SPLIT");
    no_pretest = on("RT_MI355X_NO_PRETEST");
    no_sure_table = on("RT_MI355X_NO_SURE_TABLE");
    const char* ri = getenv("RT_MI355X_ROW_INTERLEAVE");
    row_interleave = (ri && (ri[0] == '0' || ri[0] == '1') && ri[1] == 0) ? ri[0] - '0' : -1;
    const char* mc = getenv("RT_MI355X_MACRO_CAP");
    macro_cap = mc ? strtol(mc, nullptr, 10) : 0;
  }
};
// (tests flip switches between tracers of one process: the snapshot is taken per tracer, at rt_tracer_create)
inline Env read_env() { return Env(); }

void set_global_error(const std::string& s) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  g_last_error = s;
  static const bool log = getenv("RT_MI355X_LOG") != nullptr;
  if (log) fprintf(stderr, "[rt_mi355x] %s\n", s.c_str());
}

std::string fmt(const char* f, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf, sizeof buf, f, ap);
  va_end(ap);
  return buf;
}

struct HipFail { std::string what; };
#define HIP_CHECK(expr)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) throw HipFail{fmt("%s failed: %s", #expr, hipGetErrorString(e_))}; \
  } while (0)

// jump table: built once per process, uploaded once per device
std::mutex g_jump_mu;
std::vector<uint32_t> g_jump_host;
std::map<int, uint32_t*> g_jump_dev;    // per device: the jump table followed by the window tables

const std::vector<uint32_t>& jump_host() {
  std::lock_guard<std::mutex> lk(g_jump_mu);
  if (g_jump_host.empty()) g_jump_host = rth::build_jump_table();
  return g_jump_host;
}

constexpr size_t kJumpWords = 32u * 160u * 8u;

uint32_t* jump_device(int device) {
  const std::vector<uint32_t>& h = jump_host();
  std::lock_guard<std::mutex> lk(g_jump_mu);
  auto it = g_jump_dev.find(device);
  if (it != g_jump_dev.end()) return it->second;
  static const std::vector<uint32_t> win = rth::build_window_tables(h);
  uint32_t* d = nullptr;
  HIP_CHECK(hipMalloc(&d, (h.size() + win.size()) * sizeof(uint32_t)));
  HIP_CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(d + h.size(), win.data(), win.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  g_jump_dev[device] = d;
  return d;
}

// ThinLensCamera host side, ThinLensCamera.cuh:16-28,79-108,132-141 (host code: no fusing)
struct Camera {
  float position[3];   // mPosition: stored, never used (reference quirk Q1)
  float angles[2];     // mRotationAngles, radians
  float fov;           // mFov, radians
  float focal, aperture;
  float M[16];         // column-major mCameraTransformation

  static float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

  void transform() {                                                     // :132-141
    float sx, cx, sy, cy;
    rtd::sincos_spec(angles[0] * 0.5f, sx, cx);                          // glm::angleAxis
    rtd::sincos_spec(angles[1] * 0.5f, sy, cy);
    const float Xw = cx, Xx = 1.0f * sx, Xy = 0.0f * sx, Xz = 0.0f * sx;  // qX
    const float Yw = cy, Yx = 0.0f * sy, Yy = 1.0f * sy, Yz = 0.0f * sy;  // qY
    const float w = Yw * Xw - Yx * Xx - Yy * Xy - Yz * Xz;                // qY * qX
    const float x = Yw * Xx + Yx * Xw + Yy * Xz - Yz * Xy;
    const float y = Yw * Xy + Yy * Xw + Yz * Xx - Yx * Xz;
    const float z = Yw * Xz + Yz * Xw + Yx * Xy - Yy * Xx;
    const float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z;
    const float qwx = w * x, qwy = w * y, qwz = w * z;
    memset(M, 0, sizeof M);                                              // glm::mat4_cast
    M[0] = 1.0f - 2.0f * (qyy + qzz); M[1] = 2.0f * (qxy + qwz);        M[2] = 2.0f * (qxz - qwy);
    M[4] = 2.0f * (qxy - qwz);        M[5] = 1.0f - 2.0f * (qxx + qzz); M[6] = 2.0f * (qyz + qwx);
    M[8] = 2.0f * (qxz + qwy);        M[9] = 2.0f * (qyz - qwx);        M[10] = 1.0f - 2.0f * (qxx + qyy);
    M[15] = 1.0f;
  }
  float tan_half_fov() const {                                           // :114, hoisted per launch
    float s, c;
    rtd::sincos_spec(fov / 2.0f, s, c);
    return s / c;
  }
};

#ifndef RT_EVENT_STRIDE
#define RT_EVENT_STRIDE 16        // every 16th launch carries timing events (an event record costs its stream 1.4 us: stride 4 -> 16 bought 1.8 % of a C3 step)
#endif
#ifndef RT_PRETEST_LIST
#define RT_PRETEST_LIST 84u      // per-wave list capacity of the dense-scene kernels with forms: 4 x 84 x 116 bytes = 38 KiB of LDS per block (C4's fullest tile: 48)
#endif
struct EventPair { hipEvent_t a, b, c; uint32_t launches; bool split; uint64_t seq; };   // c: end of the lower half on stream_b

// The render thread of a tracer.  The reference starts a std::thread per Trace and joins the previous one first
// (RayTracerImpl.cu:69-87); its only caller re-traces on every mouse-move event (OpenGLView/MainFrame.cpp:394-444), so the
// thread's start-up is part of every frame's latency.  Here ONE thread per tracer, created by the first Trace, runs the
// Traces one after the other: between two of them it polls for the next job for a short while (a drag loop's next Trace
// arrives within microseconds of the finished callback) and then parks on a condition variable.  What a caller can observe
// is unchanged: run() returns at once, the job and its callbacks run on a thread that is not the caller's, wait_idle() is
// the join.
class RenderThread {
 public:
  ~RenderThread() { shutdown(); }
  bool busy() const { return busy_.load(std::memory_order_acquire); }
  // hands `job` to the render thread; the previous job has finished (callers cancel + wait_idle() first)
  void run(std::function<void()> job) {
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return !busy_.load(); });
    if (!th_.joinable()) th_ = std::thread([this] { loop(); });
    job_ = std::move(job);
    busy_.store(true, std::memory_order_release);
    posted_.store(true, std::memory_order_release);
    lk.unlock();
    cv_.notify_one();
  }
  void wait_idle() {
    if (!busy()) return;
    for (int i = 0; i < 2000 && busy(); ++i) spin_pause();              // a short Trace ends within microseconds
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return !busy_.load(); });
  }
  void shutdown() {
    {
      std::unique_lock<std::mutex> lk(mu_);
      done_cv_.wait(lk, [&] { return !busy_.load(); });
      quit_ = true;
    }
    cv_.notify_one();
    if (th_.joinable()) th_.join();
  }

 private:
  static void spin_pause() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
  }
  void loop() {
    for (;;) {
      // poll ~50 us for the next job before parking (no lock taken while polling)
      const auto t0 = std::chrono::steady_clock::now();
      while (!posted_.load(std::memory_order_acquire) &&
             std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(50)) spin_pause();
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return posted_.load() || quit_; });
        if (!posted_.load()) return;                                   // quit
        posted_.store(false);
        job = std::move(job_);
        job_ = nullptr;
      }
      job();
      {
        std::lock_guard<std::mutex> lk(mu_);
        busy_.store(false, std::memory_order_release);
      }
      done_cv_.notify_all();
    }
  }
  std::thread th_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  std::function<void()> job_;
  std::atomic<bool> busy_{false}, posted_{false};
  bool quit_ = false;
};

struct Group;        // rt_multi.hpp: the tile gather of a frame sharded over several GPUs
struct MultiState;   // rt_multi.hpp: the bands of a multi-device tracer

}  // namespace

struct rt_tracer {
  // A handle is one of: a plain tracer (one device, the whole frame or one row band), a band tracer that
  // joined a multi-process group (grp != null), or a multi-device tracer (mg != null: the fields below then
  // describe the whole frame and hold camera, callbacks, render thread and error text; the device buffers
  // live in the band tracers mg owns).
  Group* grp = nullptr;
  MultiState* mg = nullptr;
  // configuration
  Env env;                          // the environment switches as they were at rt_tracer_create
  int device = 0;
  uint32_t W = 0, H = 0;            // full image
  uint32_t row0 = 0, rows = 0;      // owned band
  bool band_mode = false;
  uint64_t seed = 1;
  bool fma = true, filter = true, bin = true, nearest_hit = false;
  bool smooth_normals = false;        // RT_FLAG_SMOOTH_NORMALS; takes effect for edge-format scenes
  float4* d_tri_n = nullptr;          // 3 unpacked vertex normals per triangle, edge-format scenes only
  uint32_t k_req = 0, chunk_req = 0, bin_list_req = 0;

  // device state
  hipStream_t stream = nullptr;       // primary stream: everything that is not the lower half of a split launch
  // Trace launches of tall frames are split into two half-frame kernels on two streams: consecutive
  // launches then overlap one half's drain (falling occupancy at the end of a kernel) with the other half's
  // bulk -- 146 -> 131 us per back-to-back C3 step (tools/two_stream.py); a pixel's launches stay ordered
  // because its half always uses the same stream.  main_stream() is the ordering point for everything else.
  hipStream_t stream_b = nullptr;
  hipEvent_t join_event = nullptr, fork_event = nullptr;
  bool b_dirty = false;               // work on stream_b the primary stream has not waited for yet
  bool a_dirty = false;               // non-launch work on the primary stream that stream_b has not waited for yet
  bool split_launches = true;         // RT_MI355X_NO_SPLIT=1 turns it off

  // The ordering state is shared by the render thread and by entry points that do not join it
  // (rt_tracer_sync, rt_tracer_read_buffer, the device copies): order_mu serialises the dirty flags and the
  // re-recording of the two shared events.
  std::mutex order_mu;
  hipStream_t main_stream() {         // primary stream, made to wait for everything enqueued on stream_b
    std::lock_guard<std::mutex> lk(order_mu);
    if (b_dirty) {
      HIP_CHECK(hipEventRecord(join_event, stream_b));
      HIP_CHECK(hipStreamWaitEvent(stream, join_event, 0));
      b_dirty = false;
    }
    a_dirty = true;
    return stream;
  }
  void fork_b() {                     // stream_b waits for the non-launch work enqueued on the primary stream
    std::lock_guard<std::mutex> lk(order_mu);
    if (!a_dirty) return;
    HIP_CHECK(hipEventRecord(fork_event, stream));
    HIP_CHECK(hipStreamWaitEvent(stream_b, fork_event, 0));
    a_dirty = false;
  }
  void mark_b_dirty() { std::lock_guard<std::mutex> lk(order_mu); b_dirty = true; }
  float4* d_render = nullptr;
  uint32_t* d_counts = nullptr;
  uint32_t* d_image = nullptr;
  uint32_t* d_rng = nullptr;
  uint32_t* h_image = nullptr;      // pinned, handed to callbacks
  uint32_t* h_image_alt = nullptr;  // second pinned image: update i+1 is produced while the callback reads update i
  uint32_t* image_mirror = nullptr; // rt_tracer_set_image_mirror: second target of emitting rt_tracer_launch* / trace_enqueue launches
  hipEvent_t handoff_event = nullptr;
  int handoff_next = 0;             // which of the two host images the next emitting launch of a Trace writes
  float4* d_tri = nullptr;          // (e2.xyz,e1.x),(e1.yz,v0.xy) records
  float* d_tri_b = nullptr;         // v0.z
  float4* d_tri_color = nullptr;
  uint32_t n_tris = 0;
  float4* d_spheres = nullptr;
  uint32_t n_spheres = 0;

  // camera + callbacks (guarded by state_mu; snapshotted per launch like the by-value kernel argument)
  std::mutex state_mu;
  Camera cam;
  rt_callback_fn update_cb = nullptr; void* update_user = nullptr;
  rt_callback_fn finished_cb = nullptr; void* finished_user = nullptr;

  // render thread
  std::mutex api_mu;
  RenderThread render;
  std::atomic<bool> stopped{false};
  std::atomic<bool> completed{false};

  // timing
  std::mutex time_mu;
  std::vector<EventPair> pending;
  std::vector<EventPair> free_events;
  uint64_t next_event_seq = 0;
  double kernel_ms = 0.0;
  double span_ms = 0.0;             // same sampled launches, each counted to the end of the later of its halves
  uint64_t kernel_launches = 0;

  std::mutex err_mu;
  std::string last_error;
  uint32_t last_k = 0, last_chunk = 0, last_lds = 0;

  void set_error(const std::string& s) {
    { std::lock_guard<std::mutex> lk(err_mu); last_error = s; }
    set_global_error(s);
  }
  uint32_t npix() const { return W * rows; }
  void use_device() { HIP_CHECK(hipSetDevice(device)); }

  void cancel_and_join() {                                               // RayTracerImpl.cu:72-77
    if (render.busy()) {
      stopped = true;
      render.wait_idle();
      stopped = false;
    }
  }

  void release_buffers() {                                               // :317-342
    if (d_render) (void)hipFree(d_render);
    if (d_counts) (void)hipFree(d_counts);
    if (d_image) (void)hipFree(d_image);
    if (d_rng) (void)hipFree(d_rng);
    if (h_image) (void)hipHostFree(h_image);
    if (h_image_alt) (void)hipHostFree(h_image_alt);
    h_image_alt = nullptr;
    d_render = nullptr; d_counts = nullptr; d_image = nullptr; d_rng = nullptr; h_image = nullptr;
  }

  void create_states() {                                                 // random::CreateStates, Random.cu:32-52
    uint32_t seeded[6];
    rth::seed_state(seed, seeded);
    const uint32_t p0 = row0 * W;                                        // subsequence of the band's first pixel
    if (static_cast<uint64_t>(W) * H > 0xFFFFFFFFull) throw HipFail{"frames above 2^32 pixels are not supported (32-bit pixel index, Kernels.cuh:128)"};
    uint32_t* const tables = jump_device(device);
    HIP_CHECK(rtk::launch_rng_init(d_rng, npix(), p0, seeded, tables, tables + kJumpWords, main_stream()));
  }

  void create_buffers() {                                                // ctor :33-40, Resize :96-102
    const size_t n = npix();
    if (n == 0) throw HipFail{"image has no pixels"};
    HIP_CHECK(hipMalloc(&d_rng, n * 6 * sizeof(uint32_t)));
    HIP_CHECK(hipMalloc(&d_render, n * sizeof(float4)));
    HIP_CHECK(hipMalloc(&d_counts, n * sizeof(uint32_t)));
    HIP_CHECK(hipMalloc(&d_image, n * sizeof(uint32_t)));
    HIP_CHECK(hipHostMalloc(&h_image, n * sizeof(uint32_t), hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc(&h_image_alt, n * sizeof(uint32_t), hipHostMallocDefault));
    memset(h_image, 0, n * sizeof(uint32_t));
    memset(h_image_alt, 0, n * sizeof(uint32_t));
    // the reference leaves new buffers uninitialised until the first Trace clears them; we
    // zero them so that reading before a Trace is defined
    HIP_CHECK(hipMemsetAsync(d_render, 0, n * sizeof(float4), main_stream()));
    HIP_CHECK(hipMemsetAsync(d_counts, 0, n * sizeof(uint32_t), main_stream()));
    HIP_CHECK(hipMemsetAsync(d_image, 0, n * sizeof(uint32_t), main_stream()));
    create_states();
    HIP_CHECK(hipStreamSynchronize(main_stream()));
  }

  // multi-device Resize: new frame size AND new band of it (scene, camera and options stay)
  void reshape(uint32_t w, uint32_t h, uint32_t r0, uint32_t n) {
    HIP_CHECK(hipStreamSynchronize(main_stream()));
    sync_list_stream();
    release_buffers();
    W = w; H = h; row0 = r0; rows = n;
    list_key_valid = false;
    create_buffers();
  }

  // the band's BGRA8 image to a gather buffer on the same device (paths that converted without a trace launch)
  void copy_image_to(uint32_t* target) {
    HIP_CHECK(hipMemcpyAsync(target, d_image, static_cast<size_t>(npix()) * sizeof(uint32_t), hipMemcpyDeviceToDevice, main_stream()));
  }

  // Focal points of a full 8x8 tile from its four corner pixels.  F = pos + focal * d, d = M q / |M q|, q = (cx, cy, -1)
  // affine in the pixel (ThinLensCamera.cuh:116-128).  Along an axis direction h the second derivative of x -> M x / |M x|
  // at q is ((3 c^2 - 1) u - 2 c h') |h'|^2 / |M q|^2 (u = M q / |M q|, h' = M h / |M h|, c = u.h'), of norm
  // <= 4 smax^2 / (smin^2 |q|^2) <= 4 lmax / lmin with lmax, lmin bounds of the eigenvalues of M^T M (Gershgorin; 1 for the
  // rotation the camera builds) and |q| >= 1.  A bilinear interpolant over a rectangle of sides a x b is off by at most
  // (a^2 sup|f_xx| + b^2 sup|f_yy|) / 8 in every direction, and its extremes are at the corners; cx, cy are monotone in the
  // pixel index (rounded operations are monotone), so the corner pixels bound the rectangle.  Evaluated in double, rounded up.
  void tile_corner_bound(rtk::TraceParams& p) const {
    p.tile_curv = -1.0f; p.tile_round = 0.0f;
    double G[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        G[i][j] = 0.0;
        for (int r = 0; r < 3; ++r) G[i][j] += static_cast<double>(p.cam[i * 3 + r]) * static_cast<double>(p.cam[j * 3 + r]);
      }
    double lmax = 0.0, lmin = 1e300;
    for (int i = 0; i < 3; ++i) {
      const double off = std::fabs(G[i][(i + 1) % 3]) + std::fabs(G[i][(i + 2) % 3]);
      lmax = std::max(lmax, G[i][i] + off);
      lmin = std::min(lmin, G[i][i] - off);
    }
    if (!(lmin > 0.25) || !(lmax < 4.0)) return;                       // not a (near-)rotation: every lane bounds
    const double hh = std::fabs(static_cast<double>(p.half_height)), asp = std::fabs(static_cast<double>(p.aspect));
    const double a = 7.0 * 2.0 * hh * asp / static_cast<double>(W), b = 7.0 * 2.0 * hh / static_cast<double>(H);
    // worth it only where the curvature term is small against the tile itself (<= 10 % of its smaller side: 1080p at
    // 70 degrees is 0.9 %); coarse wide-angle frames keep the exact range of their lanes
    if (!(0.5 * (a * a + b * b) * (lmax / lmin) <= 0.1 * std::min(a, b))) return;
    const double foc = std::fabs(static_cast<double>(p.focal));
#ifndef RT_TILE_CURV_SCALE          // teeth test of the adversarial campaign only (profiles/r02_boundary_campaign.txt)
#define RT_TILE_CURV_SCALE 1.0
#endif
    const double curv = foc * 0.5 * (a * a + b * b) * (lmax / lmin) * 1.001 * RT_TILE_CURV_SCALE + 1e-30;
    const double pos = std::max(std::fabs(p.cam[9]), std::max(std::fabs(p.cam[10]), std::fabs(p.cam[11])));
    const double round = foc * (1.0 + hh * asp + hh) + pos;
    if (!(curv <= 1e30) || !(round <= 1e30)) return;                   // NaN / inf lens: every lane bounds
    p.tile_curv = std::nextafter(static_cast<float>(curv), 3.0e38f);
    p.tile_round = std::nextafter(static_cast<float>(round), 3.0e38f);
  }

  rtk::TraceParams params(uint32_t samples) {
    rtk::TraceParams p;
    memset(&p, 0, sizeof p);
    Camera c;
    { std::lock_guard<std::mutex> lk(state_mu); c = cam; }               // *mCamera by value, :221
    p.render = d_render; p.counts = d_counts; p.rng = d_rng;
    p.W = W; p.H = H; p.row0 = row0; p.rows = rows; p.npix = npix(); p.samples = samples;
    for (int col = 0; col < 4; ++col)
      for (int r = 0; r < 3; ++r) p.cam[col * 3 + r] = c.M[col * 4 + r];
    p.half_height = c.tan_half_fov();
    p.aspect = static_cast<float>(W) / static_cast<float>(H);            // ThinLensCamera.cuh:113
    p.focal = c.focal; p.aperture = c.aperture;
    tile_corner_bound(p);
    p.tri_a = d_tri; p.tri_b = d_tri_b; p.tri_color = d_tri_color; p.n_tris = n_tris;
    p.tri_n = smooth_normals ? d_tri_n : nullptr;
    p.stats = nullptr;
    p.spheres = d_spheres; p.n_spheres = n_spheres;
    p.chunk = chunk_req ? chunk_req : 1024u;
    if (p.chunk > 4096u) p.chunk = 4096u;                                // 144 KiB of the CU's 160 KiB LDS
    // per-wave candidate list: whole (small) scene if it fits, else 256 records = 40 KiB per
    // block -> 4 blocks per CU; 64 records = 10 KiB per block lets 8 blocks (32 waves) share a CU
    // (a trace block holds its LDS until its slowest wave is done -- in a frame of mostly certain-winner tiles most resident
    // blocks are down to one or two live waves, and at 10 KiB per block the CU's LDS, not its wave slots, capped the waves in
    // flight: 32-record granularity, 5 KiB per block for scenes of up to 32 triangles)
    uint32_t want = bin_list_req ? bin_list_req : ((n_tris + 31u) / 32u) * 32u;
    want = ((want + 31u) / 32u) * 32u;
    p.bin_list = want < 32u ? 32u : want > (bin_list_req ? 960u : 256u) ? (bin_list_req ? 960u : 256u) : want;
    // scenes that do not fit the per-wave list: 192 records per wave + a 1024-entry block-level
    // pre-cull list keep the block at 34.9 KiB of LDS (4 blocks per CU)
    p.block_list = n_tris > p.bin_list ? 1024u : 0u;
    if (p.block_list != 0u && !bin_list_req) p.bin_list = 192u;
    // Per-sample conservative forms (TRACE_PRETEST) for the large-scene kernels: 76 instead of 40 bytes per
    // candidate in LDS, so 128 candidates per wave and a 448-entry block list keep the block at 40 KiB
    // (4 blocks per CU, as before).  Scenes dense enough to overflow 128-entry lists regularly lose more
    // by the extra classification rounds than the forms save (100 k triangles at 4K: 4.98 -> 5.43 ms),
    // hence the size limit; C4 (10 k): 5.93 -> 5.66 ms.
    if (pretest && filter && bin && n_tris >= kPretestMinTris && n_tris <= 50000u) {
      p.pretest_on = 1u;
      if (!bin_list_req) p.bin_list = RT_PRETEST_LIST;
      else p.bin_list = (p.bin_list + 1u) & ~1u;
      if (p.block_list != 0u) p.block_list = 448u;
    }
    return p;
  }

  // the launch-independent TRACE_* flags of this tracer
  uint32_t mode_flags(const rtk::TraceParams& p) const {
    return (nearest_hit ? rtk::TRACE_NEAREST_HIT : 0u) | (p.pretest_on ? rtk::TRACE_PRETEST : 0u) |
           (sure_hit ? 0u : rtk::TRACE_NO_SURE_HIT);
  }

  int pick_k(uint32_t samples) const {
    if (k_req == 1 || k_req == 2 || k_req == 4) return static_cast<int>(k_req);
    // K samples of a pixel in registers per pass.  4 amortises the LDS record reads on long
    // candidate lists (C4: 27 ms vs 35 ms at K = 1); scenes with a handful of triangles are
    // ray-generation bound and run ~5 % faster at 2 (fewer VGPRs, C3: 191 vs 197 us).
    const uint32_t want = (n_tris <= 128u) ? 2u : 4u;
    return samples >= want ? static_cast<int>(want) : samples >= 2 ? 2 : 1;
  }

  EventPair take_events() {
    std::lock_guard<std::mutex> lk(time_mu);
    if (!free_events.empty()) { EventPair e = free_events.back(); free_events.pop_back(); return e; }
    EventPair e{};
    HIP_CHECK(hipEventCreate(&e.a));
    HIP_CHECK(hipEventCreate(&e.b));
    HIP_CHECK(hipEventCreate(&e.c));         // timed as well: the cost of a split launch is the later of its two halves (span_ms)
    return e;
  }

  // RunTraceKernel, RayTracerImpl.cu:204-234, without the blocking wait.  `flags` are the
  // TRACE_* fusions: the first launch after the clear treats the accumulators as zero (no
  // memset, no accumulator read), a launch whose result is handed out also writes BGRA8.
  // sync_after: 0 = none, 1 = wait for this launch (the reference's behaviour, :228),
  // N > 1 = keep at most N launches in flight (wait for the launch N-1 back).
  void enqueue_trace_launch(uint32_t samples, uint32_t flags, int sync_after, uint32_t iters = 1,
                            uint32_t* host_image = nullptr, bool allow_split = true) {
    const int K = pick_k(samples);
    rtk::TraceParams p = params(samples);
    p.iters = iters;
    p.image_host = host_image;
    p.flags = flags | mode_flags(p);
    p.image = d_image;
    bool have_lists = false;
    const bool build_lists = prepare_tile_lists(p, (flags & rtk::TRACE_ZERO_ACC) != 0u, have_lists);
    attach_sure_table(p, have_lists);
    last_k = K; last_chunk = p.chunk;
    last_lds = rtk::trace_lds_bytes(p, bin);
    // Event pairs bracket every `event_stride`-th launch (and every launch the caller waits for):
    // an event record is a packet of its own that the next kernel has to wait behind -- measured
    // 5.6 us per C3 step (157.9 -> 152.3 us) and 2.3x on the 38x21 interactive loop (13.9 -> 6.0 us
    // per iteration) with both events on every launch.  The mean of the sampled launches is what
    // rt_tracer_kernel_time reports; the first launch after a reset is always sampled.
    const bool timed = sync_after == 1 || (launch_counter++ % kEventStride) == 0u;
    // Tall frames: upper half on the primary stream, lower half on stream_b (see the fields' comment).
    // The split row is a multiple of 8, each half is a row band of its own (own tile / macro lists).
    const uint32_t r0 = allow_split ? split_row(p.rows) : 0u;
    EventPair e{};
    if (timed) { e = take_events(); e.launches = 1; e.split = r0 != 0u; }
    if (r0 == 0u) {
      (void)main_stream();                                               // a launch on one stream orders behind both
      if (build_lists) build_tile_lists_ahead(p);
      attach_tile_lists(p, have_lists);
      if (timed) HIP_CHECK(hipEventRecord(e.a, stream));
      if (have_lists) wait_for_lists(stream, list_waited_a);
      attach_macro_lists(p, 0, stream, (flags & rtk::TRACE_ZERO_ACC) != 0u);   // part of the launch: timed with it
      HIP_CHECK(rtk::launch_trace(p, fma, filter, bin, K, stream));
      if (timed) HIP_CHECK(hipEventRecord(e.b, stream));
    } else {
      fork_b();
      if (build_lists) build_tile_lists_ahead(p);
      // The two halves overlap best in ANTI-phase (one half's drain under the other's bulk); started together -- both
      // released by the same event, or from an idle device -- they can lock IN phase and stay there for a whole run
      // (measured at C3: 93 instead of 80 us per step, profiles/r03_phase_regimes.txt).  The first split launch after
      // the tracer was idle therefore lets its second kernel start about half a kernel behind its first (a delay wave, or
      // -- before any kernel has been sampled -- behind the first kernel's end).  Later launches free-run.
      const bool stagger = stagger_next.exchange(false);
      // Small scenes: the halves are the band's upper and lower rows or its even and odd block rows (want_interleave()).
      // Dense scenes keep row halves (their macro lists are per half, in macro tiles of 8 block rows).
      const bool interleave = have_lists && want_interleave();
      if (have_lists && interleave != rows_interleaved) {                  // pixels change streams: everything before goes first
        (void)main_stream();
        fork_b();
        rows_interleaved = interleave;
        if (env.log) fprintf(stderr, "[rt_mi355x] split launches: halves by %s\n", interleave ? "even / odd block rows" : "rows");
      }
      rtk::TraceParams half[2] = {sub_band(p, 0u, r0), sub_band(p, r0, p.rows - r0)};
      if (interleave) { half[0] = p; half[1] = p; half[0].row_il = half[1].row_il = 1u; half[1].row_phase = 1u; }
      hipStream_t st[2] = {stream, stream_b};
      if (timed) HIP_CHECK(hipEventRecord(e.a, stream));                 // the sampled duration is the upper half-frame kernel's
      for (int h = 0; h < 2; ++h) {
        attach_tile_lists(half[h], have_lists);
        if (have_lists) wait_for_lists(st[h], h == 0 ? list_waited_a : list_waited_b);
        if (h == 1 && stagger) {
          // half a kernel behind the upper half: by the clock when the tracer knows how long its half-frame kernels take
          // (0.45 of the last sampled one), else behind the upper half's end
          const uint32_t us = static_cast<uint32_t>(last_half_ms.load() * 450.0f);
          if (us >= 5u) HIP_CHECK(rtk::launch_delay(us, stream_b));
          else HIP_CHECK(hipStreamWaitEvent(stream_b, stagger_event, 0));
        }
        attach_macro_lists(half[h], h, st[h], (flags & rtk::TRACE_ZERO_ACC) != 0u);
        HIP_CHECK(rtk::launch_trace(half[h], fma, filter, bin, K, st[h]));
        if (h == 0 && stagger) HIP_CHECK(hipEventRecord(stagger_event, stream));
      }
      if (timed) { HIP_CHECK(hipEventRecord(e.b, stream)); HIP_CHECK(hipEventRecord(e.c, stream_b)); }
      mark_b_dirty();
    }
    if (!timed) return;
    const EventPair* wait_for = nullptr;
    EventPair waited{};
    size_t back = 2;
    {
      std::lock_guard<std::mutex> lk(time_mu);
      // enqueue-only callers (rt_tracer_trace_enqueue / rt_tracer_launch*) never wait here: recycle what has
      // finished meanwhile, so that a long enqueue loop without rt_tracer_sync does not grow `pending`
      if (sync_after == 0) reap_finished_locked();
      e.seq = next_event_seq++;
      pending.push_back(e);
      // flow control in units of sampled launches: with stride s the launch waited for is
      // max(s, sync_after) launches back, i.e. fewer than sync_after + s launches are in flight
      const size_t stride = kEventStride;
      back = (static_cast<size_t>(sync_after > 1 ? sync_after : 2) + stride - 1u) / stride;
      if (back < 2u) back = 2u;
      if (sync_after == 1) { waited = e; wait_for = &waited; }
      else if (sync_after > 1 && pending.size() >= back) { waited = pending[pending.size() - back]; wait_for = &waited; }
    }
    if (wait_for) {
      HIP_CHECK(hipEventSynchronize(wait_for->b));                        // :228
      if (wait_for->split) HIP_CHECK(hipEventSynchronize(wait_for->c));
      if (sync_after > 1) drain_events(back - 1u);                        // everything older has finished: recycle
    }
  }

  // a row band [off, off + n) of a launch as a launch of its own
  static rtk::TraceParams sub_band(const rtk::TraceParams& p, uint32_t off, uint32_t n) {
    rtk::TraceParams q = p;
    const size_t px = static_cast<size_t>(off) * p.W;
    q.row0 = p.row0 + off; q.rows = n;
    q.render = p.render + px; q.counts = p.counts + px; q.rng = p.rng + px;    // npix stays the RNG planes' stride
    q.image = p.image + px;
    if (p.image_host != nullptr) q.image_host = p.image_host + px;
    return q;
  }

  // account and recycle the oldest n_done pairs (their launches have finished); time_mu held
  void recycle_locked(size_t n_done) {
    for (size_t i = 0; i < n_done; ++i) {
      EventPair& e = pending[i];
      float ms = 0.0f;
      if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
        kernel_ms += ms; kernel_launches += e.launches;
        if (e.split && e.launches == 1) last_half_ms = ms;
        // what the launch COST: for a split launch from the start of the upper half to the end of the later half (the
        // lower half runs on stream_b; a band whose expensive rows sit there must not look cheap to the load balancer)
        float lower = 0.0f;
        if (e.split && hipEventElapsedTime(&lower, e.a, e.c) == hipSuccess && lower > ms) ms = lower;
        span_ms += ms;
      }
      free_events.push_back(e);
    }
    pending.erase(pending.begin(), pending.begin() + static_cast<std::ptrdiff_t>(n_done));
  }
  // the event pairs of finished launches, keeping the newest `keep_last` (the caller knows they finished)
  void drain_events(size_t keep_last = 0) {
    std::lock_guard<std::mutex> lk(time_mu);
    if (pending.size() > keep_last) recycle_locked(pending.size() - keep_last);
  }
  // only the pairs pushed before `seq_end` (a caller that synchronised the streams at that point: pairs the
  // render thread has pushed since may still be in flight)
  void drain_events_before(uint64_t seq_end) {
    std::lock_guard<std::mutex> lk(time_mu);
    size_t n = 0;
    while (n < pending.size() && pending[n].seq < seq_end) ++n;
    recycle_locked(n);
  }
  void reap_finished_locked() {        // oldest first, no waiting
    size_t n = 0;
    while (n < pending.size() && hipEventQuery(pending[n].b) == hipSuccess &&
           (!pending[n].split || hipEventQuery(pending[n].c) == hipSuccess)) ++n;
    (void)hipGetLastError();           // hipErrorNotReady is an answer, not a failure of the next launch
    recycle_locked(n);
  }
  uint64_t event_seq_now() { std::lock_guard<std::mutex> lk(time_mu); return next_event_seq; }

  void clear_accumulators() {                                            // :242-243
    HIP_CHECK(hipMemsetAsync(d_render, 0, static_cast<size_t>(npix()) * sizeof(float4), main_stream()));
    HIP_CHECK(hipMemsetAsync(d_counts, 0, static_cast<size_t>(npix()) * sizeof(uint32_t), main_stream()));
  }

  void convert() {                                                       // RunConverterKernel :189-202
    HIP_CHECK(rtk::launch_convert(d_render, d_counts, d_image, npix(), main_stream()));
  }

  // Waits for a stream with the host polling: the end of a Trace is latency, not throughput (the reference's caller re-traces
  // on every mouse-move event), and the runtime's blocking wait adds its wake-up to every frame.  Long waits block.
  static void sync_polling(hipStream_t st) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t e = hipStreamQuery(st);
      if (e == hipSuccess) return;
      if (e != hipErrorNotReady) HIP_CHECK(e);
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    (void)hipGetLastError();                                             // hipErrorNotReady is an answer, not a failure
    HIP_CHECK(hipStreamSynchronize(st));
  }

  void fetch_image() {                                                   // device image -> pinned host copy
    HIP_CHECK(hipMemcpyAsync(h_image, d_image, static_cast<size_t>(npix()) * sizeof(uint32_t),
                             hipMemcpyDeviceToHost, main_stream()));
    HIP_CHECK(hipStreamSynchronize(main_stream()));                             // :259,:287
  }

  // Candidate lists (ONEPASS scenes) are kept across launches: the launch that clears the
  // accumulators (the first of a Trace) classifies as usual and stores nothing, so one-launch
  // passes -- bench.py's step -- neither pay for nor profit from the cache; the first
  // accumulating launch classifies and stores its tiles' lists, later ones load them as long as
  // camera snapshot, scene, frame, list length and arithmetic mode are unchanged (they do not
  // depend on the samples).
  struct ListKey {
    float cam[12], half_height, aspect, focal, aperture;
    uint32_t W, H, row0, rows, bin_list, n_tris, scene_generation;
    bool fma;
  };
  // The lists live in a small ring of buffers and are built on a stream of their own (stream_l, high priority): a build is
  // enqueued when the launch that needs it is enqueued, so it runs UNDER the trace kernels of the previous launch instead of
  // in front of its own (measured in-stream: each half-frame build took 35-45 us competing for wave slots with the other
  // half's trace kernel and stalled its own stream meanwhile, profiles/r03_lists_inline_timeline.txt).  Ordering: the trace
  // streams wait for list_ready[slot] (recorded on stream_l behind the build); a build into a slot waits until the slot's last
  // readers are done.  An event record is a packet the next kernel of its stream queues behind (two of them per step cost
  // 4 us of a 66 us C3 step), so the trace streams record a "free" event only every kFreeStride-th build -- it covers every
  // kernel enqueued before it -- and the ring is long enough that a build always finds such an event that is at least
  // kFreeStride builds old and still covers the readers of the slot it overwrites (build m: the oldest recorded at a build
  // e >= m - kListRing + 1; then m - kListRing < e <= m - kFreeStride).
  static constexpr int kListRing = 8, kFreeEvents = 10;
  int ring_n = 8, kFreeStride = 4;    // (4 : 2 when one slot exceeds 128 MiB; measured alternatives: HISTORY.md "List ring")
  uint32_t* d_list_ring[kListRing] = {};
  hipEvent_t list_ready[kListRing] = {};
  hipEvent_t list_free_a[kFreeEvents] = {}, list_free_b[kFreeEvents] = {};
  uint64_t list_free_build[kFreeEvents] = {};   // the build index each pair was recorded at (0 = never)
  uint64_t list_alloc_build = 0;      // builds before this one wrote buffers that no longer exist
  uint64_t list_free_waited = 0;      // the build index of the free-event pair stream_l waited for last
  int list_cur = 0;                   // slot of the current lists
  uint64_t list_builds = 0;           // builds so far; the trace streams remember which one they have waited for
  uint64_t list_waited_a = 0, list_waited_b = 0;
  hipStream_t stream_l = nullptr;
  std::atomic<bool> stagger_next{true};   // the next split launch starts from an idle tracer: stagger its halves (enqueue_trace_launch)
  hipEvent_t stagger_event = nullptr;
  std::atomic<float> last_half_ms{0.0f};  // duration of the last sampled upper half-frame kernel of a split launch
  size_t tile_lists_words = 0;
  uint32_t* tile_lists_now() const { return d_list_ring[list_cur]; }
  // Which halves a split small-scene launch uses: the band's upper and lower rows (best when the two cost the same: C3 61.0
  // against 62.2 us per step) or its even and odd block rows (the same cost whatever the picture: a tilted camera with 57 % /
  // 22 % ray-generating tiles above / below the split 62.6 against 70.5 us).  The two-level list builder counts the
  // ray-generating tiles per half and publishes the pair to pinned host memory behind every build; the enqueueing thread
  // reads the latest pair (a few launches old: the picture does not jump) and switches with hysteresis -- a switch moves
  // pixels from one stream to the other, so both streams are joined first.
  uint32_t* d_half_cost = nullptr;                // two device counters
  unsigned long long* h_half_cost = nullptr;      // pinned: upper | lower << 32 of the latest finished build
  bool rows_interleaved = false;
  bool want_interleave() {
    if (env.row_interleave >= 0) return env.row_interleave != 0;
    if (h_half_cost == nullptr) return false;
    const unsigned long long w = *reinterpret_cast<volatile unsigned long long*>(h_half_cost);
    const double u = static_cast<double>(w & 0xFFFFFFFFull), l = static_cast<double>(w >> 32);
    if (u + l < 16.0) return rows_interleaved;    // nothing (yet) to go by
    const double ratio = (u > l ? u : l) / ((u > l ? l : u) + 1.0);
    return rows_interleaved ? ratio > 1.15 : ratio > 1.25;
  }
  uint32_t split_row(uint32_t band_rows) const {   // first row of the lower half of a split launch (a multiple of 8); 0 = not split
    return (split_launches && band_rows >= 128u) ? ((band_rows / 2u + 7u) / 8u) * 8u : 0u;   // (40 / 45 / 55 / 60 % measured: the halves have to cost the same)
  }
  void release_tile_lists() {         // callers have synchronised every stream
    for (int r = 0; r < kListRing; ++r) { if (d_list_ring[r]) (void)hipFree(d_list_ring[r]); d_list_ring[r] = nullptr; }
    tile_lists_words = 0; list_key_valid = false; list_cur = 0; list_alloc_build = list_builds;
  }
  void sync_list_stream() { if (stream_l) HIP_CHECK(hipStreamSynchronize(stream_l)); }
  ListKey list_key{};
  bool list_key_valid = false;
  uint32_t scene_generation = 0;

  // Small scenes (no more triangles than the per-wave list holds): the tiles' candidate lists + certain-winner verdicts are
  // built by tile_lists_kernel ahead of the trace launch that needs them.  Decides once per launch whether the lists have to
  // be (re)built first -- camera snapshot, scene, frame, list length or arithmetic mode changed; or this is the first launch
  // of a Trace and the lists are not kept across Traces (rt_tracer_set_list_reuse(t, 0): bench.py's headline, every step
  // builds its own) -- and makes sure the buffer holds the whole band's lists.  have = the scene uses lists at all.
  bool prepare_tile_lists(const rtk::TraceParams& p, bool first_launch_of_trace, bool& have) {
    have = false;
    if (!bin || p.n_tris == 0u || p.n_tris > p.bin_list) return false;
    have = true;
    const size_t tiles = static_cast<size_t>((W + 31u) / 32u) * ((rows + 7u) / 8u + 1u) * 4u;   // (+1: a split adds a partial block row)
    const size_t words = tiles * (1u + p.bin_list);
    if (words > tile_lists_words) {
      HIP_CHECK(hipStreamSynchronize(main_stream()));
      sync_list_stream();
      release_tile_lists();
      {                                                                  // long lists on large frames: a shorter ring (<= 1 GiB of lists)
        const bool big = words * sizeof(uint32_t) > (size_t(128) << 20);
        ring_n = big ? 4 : 8; kFreeStride = big ? 2 : 4;
      }
      for (int r = 0; r < ring_n; ++r) {
        HIP_CHECK(hipMalloc(&d_list_ring[r], words * sizeof(uint32_t)));
        // count 0 everywhere until a launch builds; on the stream the builds run on (a hipMemset on the null stream is not
        // ordered with the non-blocking streams and may land AFTER the first build)
        HIP_CHECK(hipMemsetAsync(d_list_ring[r], 0, words * sizeof(uint32_t), stream_l));
      }
      tile_lists_words = words;
    }
    ListKey k;
    memset(&k, 0, sizeof k);                       // padding too: the key is compared bytewise
    memcpy(k.cam, p.cam, sizeof k.cam);
    k.half_height = p.half_height; k.aspect = p.aspect; k.focal = p.focal; k.aperture = p.aperture;
    k.W = p.W; k.H = p.H; k.row0 = p.row0; k.rows = p.rows; k.bin_list = p.bin_list; k.n_tris = p.n_tris;
    k.scene_generation = scene_generation; k.fma = fma;
    const bool same = list_key_valid && memcmp(&k, &list_key, sizeof k) == 0 && !(first_launch_of_trace && !reuse_across_traces);
    if (same) return false;
    list_key = k;
    list_key_valid = true;
    return true;
  }

  // Points one (half-)launch at its slots of the list buffer: a lower half starts behind the upper half's
  // block rows (the split row is a multiple of 8).
  void attach_tile_lists(rtk::TraceParams& p, bool have) {
    p.tile_lists = nullptr;
    if (!have) return;
    const size_t slot_base = static_cast<size_t>((W + 31u) / 32u) * ((p.row0 - row0) / 8u) * 4u;
    p.tile_lists = tile_lists_now() + slot_base * (1u + p.bin_list);
  }

  // Small scenes: the per-triangle table of what a certain-winner pixel accumulates in a launch of `samples` samples
  // (rtk::sure_table_kernel), rebuilt when the sample count or the scene changed.  Built on the stream that orders behind
  // both trace streams: earlier launches may still read the previous table.
  float4* d_sure_table = nullptr;
  uint32_t sure_table_cap = 0, sure_table_samples = 0;
  uint64_t sure_table_scene = ~0ull;
  void attach_sure_table(rtk::TraceParams& p, bool have) {
    p.sure_table = nullptr;
    if (!have || env.no_sure_table || p.n_tris == 0u) return;
    if (sure_table_samples != p.samples || sure_table_scene != scene_generation || sure_table_cap < p.n_tris) {
      hipStream_t st = main_stream();
      if (sure_table_cap < p.n_tris) {
        HIP_CHECK(hipStreamSynchronize(st));
        if (d_sure_table) { HIP_CHECK(hipFree(d_sure_table)); d_sure_table = nullptr; }
        HIP_CHECK(hipMalloc(&d_sure_table, static_cast<size_t>(p.n_tris) * sizeof(float4)));
        sure_table_cap = p.n_tris;
      }
      HIP_CHECK(rtk::launch_sure_table(p.tri_color, p.n_tris, p.samples, d_sure_table, st));
      sure_table_samples = p.samples; sure_table_scene = scene_generation;
    }
    p.sure_table = d_sure_table;
  }

  // The lists of the whole band, built on stream_l into the next slot of the ring (see the fields' comment).
  void build_tile_lists_ahead(const rtk::TraceParams& p_band) {
    const uint64_t m = list_builds;                                    // this build's index
    if (m > 0 && m % kFreeStride == 0) {                               // everything the trace streams hold now: the readers of every earlier build
      const int i = static_cast<int>((m / kFreeStride) % kFreeEvents);
      HIP_CHECK(hipEventRecord(list_free_a[i], stream));
      HIP_CHECK(hipEventRecord(list_free_b[i], stream_b));
      list_free_build[i] = m;
    }
    const int r = static_cast<int>(m % ring_n);
    if (m >= list_alloc_build + ring_n) {                              // the slot has readers: builds since the buffers exist wrap around
      const uint64_t e = ((m - ring_n + 1 + kFreeStride - 1) / kFreeStride) * kFreeStride;   // oldest record that covers build m - ring_n
      const int i = static_cast<int>((e / kFreeStride) % kFreeEvents);
      if (list_free_build[i] != e) throw HipFail{"list ring: the free event of the slot's readers is missing"};
      if (e != list_free_waited) {                                     // (kFreeStride builds in a row need the same pair: stream_l has it behind it already)
        HIP_CHECK(hipStreamWaitEvent(stream_l, list_free_a[i], 0));
        HIP_CHECK(hipStreamWaitEvent(stream_l, list_free_b[i], 0));
        list_free_waited = e;
      }
    }
    list_cur = r;
    rtk::TraceParams q = p_band;
    attach_tile_lists(q, true);
    const uint32_t sr = split_row(rows);
    // (counted by every 32nd build only: the atomics and the publishing kernel cost 3.5 us per step when every build has
    //  them -- and nothing to decide when the mode is pinned)
    if (sr != 0u && d_half_cost != nullptr && env.row_interleave < 0 && (m % 32u) == 0u) { q.half_cost = d_half_cost; q.cost_split_brow = sr / 8u; }
    HIP_CHECK(rtk::launch_tile_lists(q, fma, stream_l));
    HIP_CHECK(hipEventRecord(list_ready[r], stream_l));
    if (q.half_cost != nullptr) HIP_CHECK(rtk::launch_publish_half_cost(d_half_cost, h_half_cost, stream_l));   // (behind list_ready: nobody waits for it)
    ++list_builds;
  }
  // stream `st` (the primary stream or stream_b) is about to run a trace kernel that reads the current lists
  void wait_for_lists(hipStream_t st, uint64_t& waited) {
    if (waited == list_builds) return;
    HIP_CHECK(hipStreamWaitEvent(st, list_ready[list_cur], 0));
    waited = list_builds;
  }

  // Macro level of the classification (scenes that do not fit the per-wave list): sizes the
  // lists, points the launch at them and runs macro_bin_kernel on the stream ahead of the trace
  // launch.  Every launch re-bins (the camera may have changed; the pass costs N x macro tiles tests).
  static constexpr uint32_t kEventStride = RT_EVENT_STRIDE;     // every n-th launch carries timing events (an event record is a packet the next kernel queues behind)
  std::atomic<uint32_t> launch_counter{0};
  uint32_t* d_macro_lists[2] = {nullptr, nullptr};   // one per half of a split launch
  size_t macro_lists_words[2] = {0, 0};
  bool macro = true;                  // RT_FLAG_NO_MACRO_BINS turns it off
  bool pretest = true;                // RT_MI355X_NO_PRETEST=1 turns the per-sample forms off
  bool sure_hit = true;               // RT_FLAG_NO_SURE_HIT: tiles of one certainly-hit triangle run the tests anyway
  // Stored tile candidate lists (small scenes) survive from one Trace to the next while camera, lens, scene,
  // frame and arithmetic mode are unchanged -- like any acceleration structure that is rebuilt only when its
  // inputs change.  rt_tracer_set_list_reuse(t, 0) restricts the reuse to the launches of one Trace.
  bool reuse_across_traces = true;
  // the forms pay for themselves on dense scenes only (break-even ~3000 triangles at 1080p; C4: -13 %)
  static constexpr uint32_t kPretestMinTris = 4096;
  static constexpr uint32_t kMacroW = 128, kMacroH = 64, kMacroCapMax = 65536;
  // Dense scenes: a level above the macro tiles (super tiles of kSuperF x kSuperF of them, super_bin_kernel) so that a macro
  // tile tests its super tile's lists instead of the scene (rt_lists.hpp): C4 10.2 M -> ~1.5 M triangle tests per rebuild.
#ifndef RT_SUPER_F
#define RT_SUPER_F 4
#endif
  static constexpr uint32_t kSuperF = RT_SUPER_F, kSuperMinTris = 2048;
  bool super_level = true;            // RT_FLAG_NO_SUPER_BINS turns it off
  uint32_t* d_super_lists[2] = {nullptr, nullptr};
  size_t super_lists_words[2] = {0, 0};

  // Like the small scenes' tile lists the macro lists depend on camera, scene and frame only: a launch re-bins when one of
  // them changed since the lists of this half were built (key below) -- or when it is the first launch of a Trace and the
  // lists are not kept across Traces (bench.py's headline: every step bins afresh) -- and reads the kept lists otherwise
  // (accumulating launches of a progressive Trace: macro_bin_kernel is 0.15 ms per half at C4, 7 % of a launch).
  ListKey macro_key[2] = {};
  bool macro_key_valid[2] = {false, false};
  void attach_macro_lists(rtk::TraceParams& p, int half, hipStream_t st, bool first_launch_of_trace = true) {
    p.macro_lists = nullptr;
    if (!bin || !macro || p.n_tris <= p.bin_list) return;
    p.macro_w = kMacroW; p.macro_h = kMacroH;
    p.macro_nx = (p.W + p.macro_w - 1u) / p.macro_w;
    const uint32_t ny = (p.rows + p.macro_h - 1u) / p.macro_h;
    p.macro_cap = p.n_tris < kMacroCapMax ? p.n_tris : kMacroCapMax;
    if (env.macro_cap > 0 && static_cast<uint32_t>(env.macro_cap) < p.macro_cap) p.macro_cap = static_cast<uint32_t>(env.macro_cap);   // tests: force the overflow fallback
    const size_t words = static_cast<size_t>(p.macro_nx) * ny * (p.macro_cap + 1u);
    if (words > macro_lists_words[half]) {                              // (hipFree waits for the device: safe while the other half runs)
      if (d_macro_lists[half]) (void)hipFree(d_macro_lists[half]);
      d_macro_lists[half] = nullptr; macro_lists_words[half] = 0; macro_key_valid[half] = false;
      HIP_CHECK(hipMalloc(&d_macro_lists[half], words * sizeof(uint32_t)));
      macro_lists_words[half] = words;
    }
    p.macro_lists = d_macro_lists[half];
    p.super_lists = nullptr; p.macro_bounds = nullptr; p.super_f = 0u; p.super_chunks = 0u; p.super_nx = 0u;
    const uint32_t chunks = (p.n_tris + rtk::kSuperChunk - 1u) / rtk::kSuperChunk;
    if (super_level && p.n_tris >= kSuperMinTris && chunks <= rtk::kSuperMaxChunks && p.macro_nx * ny > kSuperF * kSuperF) {
      p.super_f = kSuperF; p.super_chunks = chunks;
      p.super_nx = (p.macro_nx + kSuperF - 1u) / kSuperF;
      const size_t sw = static_cast<size_t>(p.super_nx) * ((ny + kSuperF - 1u) / kSuperF) * chunks * (rtk::kSuperChunk + 1u) +
                        static_cast<size_t>(p.macro_nx) * ny * 8u;      // + the macro tiles' focal boxes behind the lists
      if (sw > super_lists_words[half]) {
        if (d_super_lists[half]) (void)hipFree(d_super_lists[half]);
        d_super_lists[half] = nullptr; super_lists_words[half] = 0; macro_key_valid[half] = false;
        HIP_CHECK(hipMalloc(&d_super_lists[half], sw * sizeof(uint32_t)));
        super_lists_words[half] = sw;
      }
      p.super_lists = d_super_lists[half];
      p.macro_bounds = reinterpret_cast<float*>(d_super_lists[half] + (sw - static_cast<size_t>(p.macro_nx) * ny * 8u));
    }
    ListKey k;
    memset(&k, 0, sizeof k);                       // padding too: the key is compared bytewise
    memcpy(k.cam, p.cam, sizeof k.cam);
    k.half_height = p.half_height; k.aspect = p.aspect; k.focal = p.focal; k.aperture = p.aperture;
    k.W = p.W; k.H = p.H; k.row0 = p.row0; k.rows = p.rows; k.bin_list = p.macro_cap * 65536u + p.macro_w * 256u + p.macro_h; k.n_tris = p.n_tris;
    k.scene_generation = scene_generation; k.fma = fma;
    const bool same = macro_key_valid[half] && memcmp(&k, &macro_key[half], sizeof k) == 0 && !(first_launch_of_trace && !reuse_across_traces);
    if (!same) {
      macro_key[half] = k;
      macro_key_valid[half] = true;
      if (p.super_lists != nullptr) HIP_CHECK(rtk::launch_super_bin(p, fma, st));
      HIP_CHECK(rtk::launch_macro_bin(p, fma, st));
    }
    attach_wave_lists(p, half, st, !same);
  }

  // Dense scenes with the per-sample forms: the tiles' candidate lists (forms + triangle index, 64 bytes per candidate) live in HBM,
  // built by wave_lists_kernel behind the macro lists -- same key, same reuse rule -- and read by dense_trace_kernel through the
  // scalar cache (rt_dense.hpp).  Sized for the list capacity, (1 + cap) x 64 bytes per tile: 0.7 GB for a 4K frame at cap 84
  // (what a launch touches is the survivors: ~75 MB at C4); frames whose lists would exceed kWaveListsMaxBytes per half
  // and instrumented launches keep the classification inside the trace kernel.
  static constexpr size_t kWaveListsMaxBytes = size_t(6) << 30;
  uint32_t* d_wave_lists[2] = {nullptr, nullptr};
  size_t wave_lists_words[2] = {0, 0};
  bool wave_lists_valid[2] = {false, false};
  void attach_wave_lists(rtk::TraceParams& p, int half, hipStream_t st, bool macro_rebuilt) {
    p.wave_lists = nullptr; p.wave_cap = 0u;
    if (macro_rebuilt) wave_lists_valid[half] = false;                  // (also when this launch does not use them: they follow the macro lists' key)
    if (!p.pretest_on || p.stats != nullptr || p.macro_lists == nullptr) return;
    const size_t tiles = static_cast<size_t>((p.W + 31u) / 32u) * ((p.rows + 7u) / 8u) * 4u;
    const size_t words = tiles * (1u + p.bin_list) * 16u;
    if (words * sizeof(uint32_t) > kWaveListsMaxBytes) return;
    if (words > wave_lists_words[half]) {                                // (hipFree waits for the device: safe while the other half runs)
      if (d_wave_lists[half]) (void)hipFree(d_wave_lists[half]);
      d_wave_lists[half] = nullptr; wave_lists_words[half] = 0; wave_lists_valid[half] = false;
      HIP_CHECK(hipMalloc(&d_wave_lists[half], words * sizeof(uint32_t)));
      wave_lists_words[half] = words;
    }
    p.wave_lists = d_wave_lists[half]; p.wave_cap = p.bin_list;
    if (!wave_lists_valid[half]) {
      HIP_CHECK(rtk::launch_wave_lists(p, fma, st));
      wave_lists_valid[half] = true;
    }
  }

  static constexpr int kWindow = 4;

  // Device-resident form of one Trace (rt_tracer_trace_enqueue): clear + iterationCount launches + conversion,
  // all enqueued, no callbacks, no host synchronisation.  `target`: second BGRA8 destination of the emitting
  // launch (the caller's mirror or a gather buffer), or null.
  void trace_enqueue_body(uint32_t iterationCount, uint32_t samplesPerIteration, uint32_t* target) {
    use_device();
    if (iterationCount == 0) {
      clear_accumulators();
      convert();
      if (target) copy_image_to(target);
      return;
    }
    const uint32_t group = fused_iterations(samplesPerIteration);
    for (uint32_t i = 0; i < iterationCount;) {
      const uint32_t n = iterationCount - i < group ? iterationCount - i : group;
      const bool last = i + n == iterationCount;
      enqueue_trace_launch(samplesPerIteration, (i == 0 ? rtk::TRACE_ZERO_ACC : 0u) | (last ? rtk::TRACE_EMIT_IMAGE : 0u),
                           0, n, last ? target : nullptr);
      i += n;
    }
  }

  // RayTracerImpl::TraceFunct, RayTracerImpl.cu:236-315 (runs on the render thread)
  // How many consecutive iterations one launch may run (1 = no fusing): bounded so that a launch
  // stays short (<= 64 samples per pixel) and a Stop() takes effect within a few launches.
  uint32_t fused_iterations(uint32_t samplesPerIteration) const {
    if (!rtk::trace_can_fuse(filter, bin) || samplesPerIteration == 0u) return 1u;
    const uint32_t n = 64u / samplesPerIteration;
    return n < 1u ? 1u : n;
  }

  void trace_funct(uint32_t iterationCount, uint32_t samplesPerIteration, uint32_t updateInterval) {
    try {
      use_device();
      bool cleared = false;                                              // :242-243, fused into launch 0
      // Update hand-off, pipelined: the launch that ends at an update point writes the BGRA8 image
      // into one of two pinned host images itself; its callback runs after the NEXT launch has been
      // enqueued, i.e. while the GPU is already tracing again (the reference converts, copies and
      // calls back with the GPU idle, :259-272).  An update whose iteration ran is always delivered,
      // also when Stop() arrives meanwhile, as in the reference's loop order.
      struct { bool due = false; uint32_t* image = nullptr; rt_callback_fn cb = nullptr; void* user = nullptr; } pend;
      auto deliver = [&] {
        if (!pend.due) return;
        HIP_CHECK(hipEventSynchronize(handoff_event));                   // :259
        pend.cb(pend.image, static_cast<size_t>(npix()) * sizeof(uint32_t), pend.user);   // :272
        pend.due = false;
      };
      uint32_t* final_image = h_image;
      uint32_t i = 0;
      while (!stopped && i < iterationCount) {                           // :246
        rt_callback_fn cb; void* user;
        { std::lock_guard<std::mutex> lk(state_mu); cb = update_cb; user = update_user; }
        auto is_update = [&](uint32_t k) { return cb != nullptr && k > 0 && updateInterval > 0 && k % updateInterval == 0; };   // :256
        // Iterations nobody observes in between -- up to the next update point or the end of the
        // Trace -- run as ONE launch (fused_iterations(): bit-identical to separate launches).
        const uint32_t last_allowed = iterationCount - 1u - i < fused_iterations(samplesPerIteration) - 1u
                                          ? iterationCount - 1u : i + fused_iterations(samplesPerIteration) - 1u;
        uint32_t e = i;                                                  // last iteration of this launch
        while (e < last_allowed && !is_update(e)) ++e;
        const bool update = is_update(e);
        const bool emit = update || e + 1 == iterationCount;
        const uint32_t flags = (cleared ? 0u : rtk::TRACE_ZERO_ACC) | (emit ? rtk::TRACE_EMIT_IMAGE : 0u);
        uint32_t* const target = emit ? (handoff_next ? h_image_alt : h_image) : nullptr;
        if (emit && pend.due && pend.image == target) deliver();         // never overwrite an image still to be handed out
        // The reference blocks on every launch (:228), which makes a stop take effect after one
        // kernel.  Here up to `kWindow` sampled launches are in flight: the host never starves the
        // GPU on short launches, and a stop still takes effect within a few launches.
        // (not split over two streams: the update hand-off is an ordering point for both halves anyway,
        //  and fused launches have no drain between their iterations: measured 18.6 vs 20.3 us per iteration)
        enqueue_trace_launch(samplesPerIteration, flags, kWindow, e - i + 1u, target, false);   // :249
        cleared = true;
        deliver();                                                       // the previous update, while this launch runs
        if (emit) { final_image = target; handoff_next ^= 1; }
        if (update) {
          HIP_CHECK(hipEventRecord(handoff_event, main_stream()));
          pend.due = true; pend.image = target; pend.cb = cb; pend.user = user;
        }
        i = e + 1u;
      }
      deliver();
      drain_events();
      if (!cleared) {                                                    // no launch ran: plain clear (+ convert below)
        clear_accumulators();
        if (!stopped) convert();
      }
      if (stopped) { HIP_CHECK(hipStreamSynchronize(main_stream())); return; }   // :280-284, no callback
      if (cleared && i == iterationCount) {
        sync_polling(main_stream());                                     // the last launch wrote final_image itself
      } else {
        fetch_image();                                                   // :287-295 (no launch ran)
        final_image = h_image;
      }
      completed = true;
      rt_callback_fn cb; void* user;
      { std::lock_guard<std::mutex> lk(state_mu); cb = finished_cb; user = finished_user; }
      if (cb != nullptr) cb(final_image, static_cast<size_t>(npix()) * sizeof(uint32_t), user);   // :302-305
    } catch (const HipFail& f) {                                         // :307-314 swallowed, but recorded
      set_error(f.what);
    } catch (...) {
      set_error("unknown failure in the render thread");
    }
  }
};

#include "rt_multi.hpp"

namespace {

size_t buffer_bytes(rt_tracer* t, int which) {
  const size_t n = t->npix();          // (a multi-device tracer: W x H of the whole frame)
  switch (which) {
    case RT_BUF_RENDER: return n * sizeof(float4);
    case RT_BUF_COUNTS: return n * sizeof(uint32_t);
    case RT_BUF_IMAGE: return n * sizeof(uint32_t);
    case RT_BUF_RNG: return n * 6 * sizeof(uint32_t);
    case RT_BUF_FRAME: {               // the gathered frame: on the root of a group only
      const Group* g = t->mg ? &t->mg->group : t->grp;
      return (g && g->has_root) ? g->frame_bytes() : 0;
    }
    default: return 0;
  }
}

void* buffer_ptr(rt_tracer* t, int which) {
  if (which == RT_BUF_FRAME || (t->mg && which == RT_BUF_IMAGE)) {
    Group* g = t->mg ? &t->mg->group : t->grp;
    return (g && g->has_root) ? g->d_frame[g->last_b < 0 ? 0 : g->last_b] : nullptr;
  }
  if (t->mg) return nullptr;           // per-band buffers of a multi-device tracer: rt_tracer_read_buffer assembles them
  switch (which) {
    case RT_BUF_RENDER: return t->d_render;
    case RT_BUF_COUNTS: return t->d_counts;
    case RT_BUF_IMAGE: return t->d_image;
    case RT_BUF_RNG: return t->d_rng;
    default: return nullptr;
  }
}

template <class F>
int guarded(rt_tracer* t, F&& f) {
  try {
    f();
    return RT_OK;
  } catch (const HipFail& e) {
    if (t) t->set_error(e.what); else set_global_error(e.what);
    return RT_ERR_HIP;
  } catch (const std::exception& e) {
    if (t) t->set_error(e.what()); else set_global_error(e.what());
    return RT_ERR_STATE;
  } catch (...) {
    if (t) t->set_error("unknown failure"); else set_global_error("unknown failure");
    return RT_ERR_STATE;
  }
}

struct DevBuf {
  void* p = nullptr;
  explicit DevBuf(size_t bytes) { HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1)); }
  ~DevBuf() { if (p) (void)hipFree(p); }
  template <class T> T* as() { return static_cast<T*>(p); }
};

int require_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_global_error(fmt("no HIP device available (%s); librt_mi355x has no CPU fallback",
                         e == hipSuccess ? "device count 0" : hipGetErrorString(e)));
    return RT_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    set_global_error(fmt("device %d out of range (%d devices)", device, n));
    return RT_ERR_INVALID;
  }
  return RT_OK;
}

}  // namespace

#include "rt_multi_api.hpp"

extern "C" {

#ifndef RT_KERNEL_SOURCE_HASH
#define RT_KERNEL_SOURCE_HASH "unknown"
#endif
const char* rt_version(void) { return "rt_mi355x 0.2 (gfx950, kernels=" RT_KERNEL_SOURCE_HASH ")"; }

int rt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* rt_last_error(void) {
  static thread_local std::string copy;
  std::lock_guard<std::mutex> lk(g_err_mu);
  copy = g_last_error;
  return copy.c_str();
}

const char* rt_tracer_last_error(rt_tracer* t) {
  static thread_local std::string copy;
  if (!t) return rt_last_error();
  std::lock_guard<std::mutex> lk(t->err_mu);
  copy = t->last_error;
  return copy.c_str();
}

int rt_tracer_create_ex(const uint32_t imageSize[2], const float cameraPosition[3],
                        const float cameraAngles[2], float fov, float focalLength, float aperture,
                        const rt_options* options, rt_tracer** out) {
  if (!out) return RT_ERR_INVALID;
  *out = nullptr;
  if (!imageSize || !cameraAngles || imageSize[0] == 0 || imageSize[1] == 0) {
    set_global_error("rt_tracer_create: invalid image size or camera angles");
    return RT_ERR_INVALID;
  }
  rt_options opt;
  memset(&opt, 0, sizeof opt);
  opt.use_time_seed = 1;                                                 // Random.cu:45 by default
  if (options) {
    const size_t n = options->struct_size < sizeof(opt) ? options->struct_size : sizeof(opt);
    if (n < 8) { set_global_error("rt_options.struct_size not set"); return RT_ERR_INVALID; }
    opt.use_time_seed = 0;
    memcpy(&opt, options, n);
  }
  int rc = require_device(opt.device);
  if (rc != RT_OK) return rc;

  rt_tracer* t = new rt_tracer();
  t->device = opt.device;
  t->W = imageSize[0];
  if (opt.full_height) {
    t->band_mode = true;
    t->H = opt.full_height; t->row0 = opt.row_begin; t->rows = imageSize[1];
    if (static_cast<uint64_t>(t->row0) + t->rows > t->H) {
      delete t;
      set_global_error("row band exceeds full_height");
      return RT_ERR_INVALID;
    }
  } else {
    t->H = imageSize[1]; t->row0 = 0; t->rows = imageSize[1];
  }
  t->seed = opt.use_time_seed ? static_cast<uint64_t>(static_cast<uint32_t>(time(nullptr))) : opt.seed;
  t->fma = opt.math_mode != RT_MATH_STRICT;
  t->filter = (opt.flags & RT_FLAG_NO_FILTER) == 0;
  t->bin = (opt.flags & RT_FLAG_NO_BINNING) == 0;
  t->nearest_hit = (opt.flags & RT_FLAG_NEAREST_HIT) != 0;
  t->smooth_normals = (opt.flags & RT_FLAG_SMOOTH_NORMALS) != 0;
  t->env = read_env();
  t->pretest = !t->env.no_pretest;
  t->sure_hit = (opt.flags & RT_FLAG_NO_SURE_HIT) == 0;
  t->split_launches = !t->env.no_split;
  t->macro = (opt.flags & RT_FLAG_NO_MACRO_BINS) == 0;
  t->super_level = (opt.flags & RT_FLAG_NO_SUPER_BINS) == 0;
  t->k_req = opt.samples_in_flight;
  t->chunk_req = opt.lds_chunk;
  t->bin_list_req = opt.bin_list;
  Camera& c = t->cam;
  for (int i = 0; i < 3; ++i) c.position[i] = cameraPosition ? cameraPosition[i] : 0.0f;
  c.angles[0] = cameraAngles[0]; c.angles[1] = cameraAngles[1];
  c.fov = Camera::radians(fov);                                          // ThinLensCamera.cuh:23
  c.focal = focalLength; c.aperture = aperture;
  c.transform();                                                         // :27
  rc = guarded(t, [&] {
    t->use_device();
    HIP_CHECK(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    // The runtime maps streams onto a few hardware queues.  The two streams of the first tracer of a process get queues
    // of their own; a tracer created while another one is alive on the device was measured 20 % slower (its two
    // half-frame kernels serialise on one queue; tools/placement_probe.py); GPU_MAX_HW_QUEUES=8 in the environment cures it.
    HIP_CHECK(hipStreamCreateWithFlags(&t->stream_b, hipStreamNonBlocking));
    {
      int lo = 0, hi = 0;                                                // (numerically lower = higher priority)
      if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo)
        HIP_CHECK(hipStreamCreateWithPriority(&t->stream_l, hipStreamNonBlocking, hi));
      else
        HIP_CHECK(hipStreamCreateWithFlags(&t->stream_l, hipStreamNonBlocking));
      HIP_CHECK(hipEventCreateWithFlags(&t->stagger_event, hipEventDisableTiming));
      HIP_CHECK(hipMalloc(&t->d_half_cost, 2 * sizeof(uint32_t)));
      HIP_CHECK(hipMemset(t->d_half_cost, 0, 2 * sizeof(uint32_t)));
      HIP_CHECK(hipDeviceSynchronize());                                  // (a null-stream memset is not ordered with the list stream)
      HIP_CHECK(hipHostMalloc(&t->h_half_cost, sizeof(unsigned long long), hipHostMallocDefault));
      *t->h_half_cost = 0ull;
      for (int r = 0; r < rt_tracer::kListRing; ++r) HIP_CHECK(hipEventCreateWithFlags(&t->list_ready[r], hipEventDisableTiming));
      for (int r = 0; r < rt_tracer::kFreeEvents; ++r) {
        HIP_CHECK(hipEventCreateWithFlags(&t->list_free_a[r], hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&t->list_free_b[r], hipEventDisableTiming));
      }
    }
    HIP_CHECK(hipEventCreateWithFlags(&t->handoff_event, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&t->join_event, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&t->fork_event, hipEventDisableTiming));
    t->create_buffers();
  });
  if (rc != RT_OK) {
    // the reference's ctor logs and carries on with null buffers (RayTracerImpl.cu:42-45);
    // a C ABI can do better: report, release, hand back nothing.
    std::string why = t->last_error;
    t->release_buffers();
    if (t->handoff_event) (void)hipEventDestroy(t->handoff_event);
    if (t->join_event) (void)hipEventDestroy(t->join_event);
    if (t->fork_event) (void)hipEventDestroy(t->fork_event);
    if (t->stream_b) (void)hipStreamDestroy(t->stream_b);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
    set_global_error("rt_tracer_create: " + why);
    return rc;
  }
  *out = t;
  return RT_OK;
}

int rt_tracer_create(const uint32_t imageSize[2], const float cameraPosition[3],
                     const float cameraAngles[2], float fov, float focalLength, float aperture,
                     rt_tracer** out) {
  return rt_tracer_create_ex(imageSize, cameraPosition, cameraAngles, fov, focalLength, aperture,
                             nullptr, out);
}

void rt_tracer_destroy(rt_tracer* t) {                                   // RayTracerImpl.cu:48-67
  if (!t) return;
  t->stopped = true;
  t->render.shutdown();
  if (t->mg) {                                                           // multi-device: the bands own the device state
    multi_destroy(t);
    delete t;
    return;
  }
  member_leave(t);
  (void)hipSetDevice(t->device);
  if (t->stream_b) (void)hipStreamSynchronize(t->stream_b);
  if (t->stream) (void)hipStreamSynchronize(t->stream);
  t->drain_events();
  for (EventPair& e : t->free_events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); (void)hipEventDestroy(e.c); }
  if (t->stream_l) (void)hipStreamSynchronize(t->stream_l);
  t->release_tile_lists();
  for (int r = 0; r < rt_tracer::kListRing; ++r)
    if (t->list_ready[r]) (void)hipEventDestroy(t->list_ready[r]);
  for (int r = 0; r < rt_tracer::kFreeEvents; ++r) {
    if (t->list_free_a[r]) (void)hipEventDestroy(t->list_free_a[r]);
    if (t->list_free_b[r]) (void)hipEventDestroy(t->list_free_b[r]);
  }
  if (t->stream_l) (void)hipStreamDestroy(t->stream_l);
  if (t->stagger_event) (void)hipEventDestroy(t->stagger_event);
  if (t->d_half_cost) (void)hipFree(t->d_half_cost);
  if (t->h_half_cost) (void)hipHostFree(t->h_half_cost);
  for (int h = 0; h < 2; ++h) if (t->d_macro_lists[h]) (void)hipFree(t->d_macro_lists[h]);
  for (int h = 0; h < 2; ++h) if (t->d_super_lists[h]) (void)hipFree(t->d_super_lists[h]);
  for (int h = 0; h < 2; ++h) if (t->d_wave_lists[h]) (void)hipFree(t->d_wave_lists[h]);
  if (t->d_tri_n) (void)hipFree(t->d_tri_n);
  if (t->d_tri) (void)hipFree(t->d_tri);
  if (t->d_tri_b) (void)hipFree(t->d_tri_b);
  if (t->d_tri_color) (void)hipFree(t->d_tri_color);
  if (t->d_sure_table) (void)hipFree(t->d_sure_table);
  if (t->d_spheres) (void)hipFree(t->d_spheres);
  t->release_buffers();
  if (t->handoff_event) (void)hipEventDestroy(t->handoff_event);
  if (t->join_event) (void)hipEventDestroy(t->join_event);
  if (t->fork_event) (void)hipEventDestroy(t->fork_event);
  if (t->stream_b) (void)hipStreamDestroy(t->stream_b);
  if (t->stream) (void)hipStreamDestroy(t->stream);
  delete t;
}

int rt_tracer_trace(rt_tracer* t, uint32_t iterationCount, uint32_t samplesPerIteration,
                    uint32_t updateInterval) {
  if (!t) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] {
    t->cancel_and_join();                                                // :72-77
    t->completed = false;
    if (t->mg) t->render.run([=] { multi_trace_funct(t, iterationCount, samplesPerIteration, updateInterval); });
    else t->render.run([=] { t->trace_funct(iterationCount, samplesPerIteration, updateInterval); });   // :80-85
  });
}

void rt_tracer_stop(rt_tracer* t) {                                      // :89-92
  if (t) t->stopped = true;
}

int rt_tracer_wait(rt_tracer* t) {
  if (!t) return 0;
  std::lock_guard<std::mutex> lk(t->api_mu);
  t->render.wait_idle();
  t->stopped = false;
  return t->completed ? 1 : 0;
}

int rt_tracer_resize(rt_tracer* t, const uint32_t size[2]) {             // :94-103
  if (!t || !size || size[0] == 0 || size[1] == 0) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] {
    t->cancel_and_join();
    if (t->mg) { multi_resize(t, size[0], size[1]); return; }
    if (t->grp) throw HipFail{"Resize: the tracer is a member of a multi-process group (leave and re-join with the new bands)"};
    t->use_device();
    HIP_CHECK(hipStreamSynchronize(t->main_stream()));
    t->release_buffers();
    t->W = size[0];
    if (t->band_mode) {
      if (static_cast<uint64_t>(t->row0) + size[1] > t->H) throw HipFail{"row band exceeds full_height"};
      t->rows = size[1];
    } else {
      t->H = size[1]; t->rows = size[1];
    }
    t->create_buffers();
  });
}

void rt_tracer_set_camera_parameters(rt_tracer* t, float fov, float focalLength, float aperture) {
  if (!t) return;                                                        // :105-112
  std::lock_guard<std::mutex> lk(t->state_mu);
  t->cam.fov = Camera::radians(fov);
  t->cam.focal = focalLength;
  t->cam.aperture = aperture;
}

void rt_tracer_rotate_camera(rt_tracer* t, const float angles[2]) {      // :114-117
  if (!t || !angles) return;
  std::lock_guard<std::mutex> lk(t->state_mu);
  t->cam.angles[0] += angles[0];                                         // ThinLensCamera.cuh:104-108
  t->cam.angles[1] += angles[1];
  t->cam.transform();
}

static int upload_scene_impl(rt_tracer* t, const rt_float4* hostData, size_t count, bool edges) {
  if (!t) return RT_ERR_INVALID;
  if (!hostData || count < 3 || count % 3 != 0) {                        // :121-125
    t->set_error(fmt("UploadScene got invalid triangle list. Size = %zu", count));
    return RT_ERR_INVALID;
  }
  std::lock_guard<std::mutex> lk(t->api_mu);
  if (t->mg) {                                                           // the scene is replicated on every device
    int rc = guarded(t, [&] { t->cancel_and_join(); multi_sync_all(t); });
    for (rt_tracer* b : t->mg->bands) {
      if (rc != RT_OK) break;
      rc = upload_scene_impl(b, hostData, count, edges);
      if (rc != RT_OK) t->set_error(b->last_error);
    }
    if (rc == RT_OK) { t->n_tris = static_cast<uint32_t>(count / 3); t->scene_generation++; }
    return rc;
  }
  return guarded(t, [&] {
    t->cancel_and_join();
    t->use_device();
    HIP_CHECK(hipStreamSynchronize(t->main_stream()));
    t->sync_list_stream();                                               // (a list build may still be reading the old records)
    if (t->d_tri) { (void)hipFree(t->d_tri); t->d_tri = nullptr; }       // :128-137
    if (t->d_tri_b) { (void)hipFree(t->d_tri_b); t->d_tri_b = nullptr; }
    if (t->d_tri_color) { (void)hipFree(t->d_tri_color); t->d_tri_color = nullptr; }
    if (t->d_tri_n) { (void)hipFree(t->d_tri_n); t->d_tri_n = nullptr; }
    t->n_tris = 0;
    const uint32_t n = static_cast<uint32_t>(count / 3);                 // :139
    DevBuf verts(count * sizeof(float4));
    HIP_CHECK(hipMalloc(&t->d_tri, static_cast<size_t>(n) * 2 * sizeof(float4)));
    HIP_CHECK(hipMalloc(&t->d_tri_b, static_cast<size_t>(n) * sizeof(float)));
    HIP_CHECK(hipMalloc(&t->d_tri_color, static_cast<size_t>(n) * sizeof(float4)));
    if (edges) HIP_CHECK(hipMalloc(&t->d_tri_n, static_cast<size_t>(n) * 3 * sizeof(float4)));
    HIP_CHECK(hipMemcpyAsync(verts.p, hostData, count * sizeof(float4), hipMemcpyHostToDevice, t->main_stream()));
    HIP_CHECK(rtk::launch_prep_triangles(t->fma, edges, verts.as<float4>(), n, t->d_tri, t->d_tri_b,
                                         t->d_tri_color, t->d_tri_n, t->main_stream()));
    HIP_CHECK(hipStreamSynchronize(t->main_stream()));
    t->n_tris = n;
    t->scene_generation++;
  });
}

int rt_tracer_upload_scene(rt_tracer* t, const rt_float4* hostData, size_t count) {
  return upload_scene_impl(t, hostData, count, false);
}

int rt_tracer_upload_scene_edges(rt_tracer* t, const rt_float4* hostData, size_t count) {
  return upload_scene_impl(t, hostData, count, true);
}

// Corrected form of the reference's experimental normal packing (UnitTests/NormalPackingTest.cpp:10-23,
// Documentation/gpu.meshes.txt:20-33): three 8-bit components in the 24-bit fraction of one float.
float rt_pack_normal(const float n[3]) {
  return floorf(n[0] * 127.0f + 127.5f) / 256.0f + floorf(n[1] * 127.0f + 127.5f) / 65536.0f +
         floorf(n[2] * 127.0f + 127.5f) / 16777216.0f;
}

void rt_unpack_normal(float packed, float n[3]) {
  // byte k sits at bits 2^-8(k+1): shift by 1, 256, 65536 (the reference's test multiplies
  // by 1, 65536, 16777216, which does not invert pack)
  const float m[3] = {1.0f, 256.0f, 65536.0f};
  for (int i = 0; i < 3; ++i) {
    const float s = packed * m[i];
    const float frac = s - floorf(s);
    n[i] = floorf(frac * 256.0f) / 127.0f - 1.0f;
  }
}

int rt_tracer_upload_spheres(rt_tracer* t, const rt_float4* spheres, size_t count) {
  if (!t || (count && !spheres)) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  if (t->mg) {
    int rc = guarded(t, [&] { t->cancel_and_join(); multi_sync_all(t); });
    for (rt_tracer* b : t->mg->bands) {
      if (rc != RT_OK) break;
      rc = rt_tracer_upload_spheres(b, spheres, count);
      if (rc != RT_OK) t->set_error(b->last_error);
    }
    if (rc == RT_OK) t->n_spheres = static_cast<uint32_t>(count);
    return rc;
  }
  return guarded(t, [&] {
    t->cancel_and_join();
    t->use_device();
    HIP_CHECK(hipStreamSynchronize(t->main_stream()));
    if (t->d_spheres) { (void)hipFree(t->d_spheres); t->d_spheres = nullptr; }
    t->n_spheres = 0;
    if (count == 0) return;
    HIP_CHECK(hipMalloc(&t->d_spheres, count * sizeof(float4)));
    HIP_CHECK(hipMemcpy(t->d_spheres, spheres, count * sizeof(float4), hipMemcpyHostToDevice));
    t->n_spheres = static_cast<uint32_t>(count);
  });
}

void rt_tracer_set_update_callback(rt_tracer* t, rt_callback_fn fn, void* user) {     // :179-182
  if (!t) return;
  std::lock_guard<std::mutex> lk(t->state_mu);
  t->update_cb = fn; t->update_user = user;
}

void rt_tracer_set_finished_callback(rt_tracer* t, rt_callback_fn fn, void* user) {   // :184-187
  if (!t) return;
  std::lock_guard<std::mutex> lk(t->state_mu);
  t->finished_cb = fn; t->finished_user = user;
}

int rt_tracer_set_seed(rt_tracer* t, uint64_t seed) {
  if (!t) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] {
    t->cancel_and_join();
    t->seed = seed;
    if (t->mg) {
      t->mg->opt.seed = seed;
      for (rt_tracer* b : t->mg->bands)
        if (rt_tracer_set_seed(b, seed) != RT_OK) throw HipFail{b->last_error};
      return;
    }
    t->use_device();
    t->create_states();
    HIP_CHECK(hipStreamSynchronize(t->main_stream()));
  });
}

// one device-resident Trace pass on any kind of handle (api_mu held, render thread idle)
static void trace_enqueue_once(rt_tracer* t, uint32_t iterationCount, uint32_t samplesPerIteration) {
  if (t->mg) { multi_trace_enqueue(t, iterationCount, samplesPerIteration); return; }
  if (t->grp) {                                                          // member of a multi-process group: trace, then the gather
    const size_t k = member_band_index(t);
    const int b = t->grp->begin_frame();
    t->trace_enqueue_body(iterationCount, samplesPerIteration, t->grp->tile_target(k, b));
    t->grp->tile_written(k);
    t->grp->gather(b);
    return;
  }
  t->trace_enqueue_body(iterationCount, samplesPerIteration, t->image_mirror);
}

int rt_tracer_trace_enqueue(rt_tracer* t, uint32_t iterationCount, uint32_t samplesPerIteration) {
  if (!t) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] {
    t->cancel_and_join();
    trace_enqueue_once(t, iterationCount, samplesPerIteration);
  });
}

int rt_tracer_trace_enqueue_n(rt_tracer* t, uint32_t iterationCount, uint32_t samplesPerIteration, uint32_t n_steps) {
  if (!t) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] {
    t->cancel_and_join();
    for (uint32_t s = 0; s < n_steps; ++s) trace_enqueue_once(t, iterationCount, samplesPerIteration);
  });
}

int rt_tracer_set_list_reuse(rt_tracer* t, int across_traces) {
  if (!t) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  t->cancel_and_join();
  if (t->mg) for (rt_tracer* b : t->mg->bands) (void)rt_tracer_set_list_reuse(b, across_traces);
  t->reuse_across_traces = across_traces != 0;
  t->list_key_valid = false;
  return RT_OK;
}

int rt_tracer_set_image_mirror(rt_tracer* t, void* device_visible_image) {
  if (!t) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  if (t->mg || t->grp) { t->set_error("rt_tracer_set_image_mirror: the gather owns the second image target of a sharded frame"); return RT_ERR_STATE; }
  t->image_mirror = static_cast<uint32_t*>(device_visible_image);
  return RT_OK;
}

int rt_tracer_fused_iterations(rt_tracer* t, uint32_t samples) {
  if (t && t->mg) return static_cast<int>(t->mg->bands[0]->fused_iterations(samples));
  return t ? static_cast<int>(t->fused_iterations(samples)) : 0;
}

// one launch of rt_tracer_launch / rt_tracer_launch_iterations on any kind of handle
static void launch_impl(rt_tracer* t, uint32_t samples, uint32_t iterations, bool clear_first, bool emit) {
  t->cancel_and_join();
  if (t->mg) { multi_launch(t, samples, iterations, clear_first, emit); return; }
  t->use_device();
  const uint32_t flags = (clear_first ? rtk::TRACE_ZERO_ACC : 0u) | (emit ? rtk::TRACE_EMIT_IMAGE : 0u);
  if (t->grp && emit) {                                                  // member of a multi-process group: the tile travels
    const size_t k = member_band_index(t);
    const int b = t->grp->begin_frame();
    t->enqueue_trace_launch(samples, flags, 0, iterations, t->grp->tile_target(k, b));
    t->grp->tile_written(k);
    t->grp->gather(b);
    return;
  }
  t->enqueue_trace_launch(samples, flags, 0, iterations, emit ? t->image_mirror : nullptr);
}

int rt_tracer_launch_iterations(rt_tracer* t, uint32_t samples, uint32_t iterations, int clear_first, int emit_image) {
  if (!t || iterations == 0u) return RT_ERR_INVALID;
  if (iterations > static_cast<uint32_t>(rt_tracer_fused_iterations(t, samples))) {
    t->set_error(fmt("rt_tracer_launch_iterations: %u iterations of %u samples exceed rt_tracer_fused_iterations", iterations, samples));
    return RT_ERR_INVALID;
  }
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] { launch_impl(t, samples, iterations, clear_first != 0, emit_image != 0); });
}

int rt_tracer_launch(rt_tracer* t, uint32_t samples, int clear_first, int emit_image) {
  if (!t) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] { launch_impl(t, samples, 1u, clear_first != 0, emit_image != 0); });
}

int rt_tracer_trace_stats(rt_tracer* t, uint32_t samples, uint64_t out[16]) {
  if (!t || !out) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  if (t->mg) {                                                           // sums over the bands
    memset(out, 0, 16 * sizeof(uint64_t));
    t->cancel_and_join();
    multi_push_camera(t);
    for (rt_tracer* b : t->mg->bands) {
      uint64_t part[16];
      const int rc = rt_tracer_trace_stats(b, samples, part);
      if (rc != RT_OK) { t->set_error(b->last_error); return rc; }
      for (int i = 0; i < 16; ++i) out[i] += part[i];
    }
    return RT_OK;
  }
  return guarded(t, [&] {
    t->cancel_and_join();
    t->use_device();
    DevBuf counters(16 * sizeof(unsigned long long));
    HIP_CHECK(hipMemsetAsync(counters.p, 0, 16 * sizeof(unsigned long long), t->main_stream()));
    t->clear_accumulators();
    rtk::TraceParams p = t->params(samples);
    p.stats = counters.as<unsigned long long>();
    p.flags = t->mode_flags(p);
    bool have_lists = false;
    const bool build_lists = t->prepare_tile_lists(p, true, have_lists);
    t->sync_list_stream();                                               // (a synchronous path: build in-stream into the current slot)
    t->attach_tile_lists(p, have_lists);
    if (have_lists) HIP_CHECK(rtk::launch_tile_lists(p, t->fma, t->main_stream()));
    (void)build_lists;
    t->attach_macro_lists(p, 0, t->main_stream());
    HIP_CHECK(rtk::launch_trace(p, t->fma, t->filter, t->bin, t->pick_k(samples), t->main_stream()));
    HIP_CHECK(hipStreamSynchronize(t->main_stream()));
    HIP_CHECK(hipMemcpy(out, counters.p, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  });
}

int rt_tracer_sync(rt_tracer* t) {
  if (!t) return RT_ERR_INVALID;
  if (t->mg) return guarded(t, [&] { multi_sync_all(t); });
  return guarded(t, [&] {
    t->use_device();
    const uint64_t seen = t->event_seq_now();        // launches enqueued so far; a running render thread may add more
    HIP_CHECK(hipStreamSynchronize(t->main_stream()));
    t->sync_list_stream();                           // (nothing the caller could read depends on it; a Sync leaves the device idle)
    t->stagger_next = true;
    t->drain_events_before(seen);
    if (t->grp) t->grp->sync();
  });
}

int rt_tracer_kernel_time(rt_tracer* t, double* total_ms, uint64_t* launches, int reset_after) {
  if (!t) return RT_ERR_INVALID;
  if (t->mg) {                                                           // of the first band; reset applies to every band
    const int rc = rt_tracer_kernel_time(t->mg->bands[0], total_ms, launches, reset_after);
    if (reset_after) for (size_t k = 1; k < t->mg->bands.size(); ++k) (void)rt_tracer_kernel_time(t->mg->bands[k], nullptr, nullptr, 1);
    return rc;
  }
  std::lock_guard<std::mutex> lk(t->time_mu);
  if (total_ms) *total_ms = t->kernel_ms;
  if (launches) *launches = t->kernel_launches;
  if (reset_after) { t->kernel_ms = 0.0; t->span_ms = 0.0; t->kernel_launches = 0; t->launch_counter = 0; }   // next launch is sampled
  return RT_OK;
}

int rt_tracer_launch_time(rt_tracer* t, double* total_ms, uint64_t* launches, int reset_after) {
  if (!t) return RT_ERR_INVALID;
  if (t->mg) {
    const int rc = rt_tracer_launch_time(t->mg->bands[0], total_ms, launches, reset_after);
    if (reset_after) for (size_t k = 1; k < t->mg->bands.size(); ++k) (void)rt_tracer_launch_time(t->mg->bands[k], nullptr, nullptr, 1);
    return rc;
  }
  std::lock_guard<std::mutex> lk(t->time_mu);
  if (total_ms) *total_ms = t->span_ms;
  if (launches) *launches = t->kernel_launches;
  if (reset_after) { t->kernel_ms = 0.0; t->span_ms = 0.0; t->kernel_launches = 0; t->launch_counter = 0; }
  return RT_OK;
}

size_t rt_tracer_buffer_bytes(rt_tracer* t, int which) { return t ? buffer_bytes(t, which) : 0; }
void* rt_tracer_device_pointer(rt_tracer* t, int which) { return t ? buffer_ptr(t, which) : nullptr; }

int rt_tracer_read_buffer(rt_tracer* t, int which, void* dst, size_t bytes) {
  if (!t || !dst) return RT_ERR_INVALID;
  if (t->mg) {
    if (bytes > buffer_bytes(t, which) || buffer_bytes(t, which) == 0) return RT_ERR_INVALID;
    return guarded(t, [&] { multi_read_buffer(t, which, dst, bytes); });
  }
  if (buffer_ptr(t, which) == nullptr || bytes > buffer_bytes(t, which)) return RT_ERR_INVALID;
  return guarded(t, [&] {
    t->use_device();
    HIP_CHECK(hipStreamSynchronize(t->main_stream()));
    if (t->grp) t->grp->sync();
    HIP_CHECK(hipMemcpy(dst, buffer_ptr(t, which), bytes, hipMemcpyDeviceToHost));
  });
}

int rt_tracer_copy_buffer_to_device(rt_tracer* t, int which, void* dst_device, size_t bytes) {
  if (!t || !dst_device || buffer_ptr(t, which) == nullptr || bytes > buffer_bytes(t, which)) return RT_ERR_INVALID;
  if (t->mg) return guarded(t, [&] {                                      // the gathered frame, from the root device
    multi_sync_all(t);
    HIP_CHECK(hipSetDevice(t->mg->group.local[0].device));
    HIP_CHECK(hipMemcpy(dst_device, buffer_ptr(t, which), bytes, hipMemcpyDeviceToDevice));
  });
  return guarded(t, [&] {
    t->use_device();
    if (t->grp && which == RT_BUF_FRAME) t->grp->sync();
    HIP_CHECK(hipMemcpyAsync(dst_device, buffer_ptr(t, which), bytes, hipMemcpyDeviceToDevice, t->main_stream()));
    HIP_CHECK(hipStreamSynchronize(t->main_stream()));
  });
}

int rt_tracer_copy_buffer_to_device_async(rt_tracer* t, int which, void* dst_device, size_t bytes) {
  if (!t || !dst_device || buffer_ptr(t, which) == nullptr || bytes > buffer_bytes(t, which)) return RT_ERR_INVALID;
  if (t->mg || which == RT_BUF_FRAME) { t->set_error("rt_tracer_copy_buffer_to_device_async: per-stream copies are for plain tracers' own buffers"); return RT_ERR_STATE; }
  return guarded(t, [&] {
    t->use_device();
    HIP_CHECK(hipMemcpyAsync(dst_device, buffer_ptr(t, which), bytes, hipMemcpyDeviceToDevice, t->main_stream()));
  });
}

void* rt_tracer_stream(rt_tracer* t) { return (t && !t->mg) ? static_cast<void*>(t->stream) : nullptr; }
void* rt_tracer_stream_b(rt_tracer* t) { return (t && !t->mg) ? static_cast<void*>(t->stream_b) : nullptr; }

int rt_tracer_info(rt_tracer* t, uint32_t out[8]) {
  if (!t || !out) return RT_ERR_INVALID;
  if (t->mg) {                                                           // launch geometry of the first band, sizes of the frame
    (void)rt_tracer_info(t->mg->bands[0], out);
    out[3] = (t->W + 31) / 32; out[4] = (t->H + 7) / 8;
    out[7] = static_cast<uint32_t>(t->device);
    return RT_OK;
  }
  out[0] = t->last_k; out[1] = t->last_chunk; out[2] = t->last_lds;
  out[3] = (t->W + 31) / 32; out[4] = (t->rows + 7) / 8;
  out[5] = t->n_tris; out[6] = t->n_spheres; out[7] = static_cast<uint32_t>(t->device);
  return RT_OK;
}

// ---- a frame sharded over several GPUs ------------------------------------------------------

int rt_tracer_create_multi(const uint32_t imageSize[2], const float cameraPosition[3], const float cameraAngles[2],
                           float fov, float focalLength, float aperture, const rt_options* options,
                           const int32_t* devices, uint32_t n_bands, rt_tracer** out) {
  return multi_create(imageSize, cameraPosition, cameraAngles, fov, focalLength, aperture, options, devices, n_bands, out);
}

int rt_group_unique_id(uint8_t id[RT_GROUP_ID_BYTES]) {
  if (!id) return RT_ERR_INVALID;
  return guarded(nullptr, [&] {
    ncclUniqueId uid;
    RCCL_CHECK(need_rccl().GetUniqueId(&uid));
    memcpy(id, uid.internal, RT_GROUP_ID_BYTES);
  });
}

int rt_tracer_join_group(rt_tracer* t, uint32_t n_ranks, uint32_t rank, const uint8_t id[RT_GROUP_ID_BYTES]) {
  if (!t || t->mg || n_ranks == 0u || rank >= n_ranks || (!id && n_ranks > 1u)) return RT_ERR_INVALID;
  static const uint8_t zero_id[RT_GROUP_ID_BYTES] = {0};
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] {
    t->cancel_and_join();
    member_join(t, n_ranks, rank, id ? id : zero_id, nullptr);
  });
}

int rt_tracer_join_group_bands(rt_tracer* t, uint32_t n_ranks, uint32_t rank, const uint8_t id[RT_GROUP_ID_BYTES],
                               const uint32_t* row_begin) {
  if (!t || t->mg || n_ranks == 0u || rank >= n_ranks || (!id && n_ranks > 1u) || !row_begin) return RT_ERR_INVALID;
  static const uint8_t zero_id[RT_GROUP_ID_BYTES] = {0};
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] {
    t->cancel_and_join();
    member_join(t, n_ranks, rank, id ? id : zero_id, row_begin);
  });
}

int rt_balance_rows(uint32_t n_bands, const uint32_t* row_begin, const double* cost, uint32_t granule, uint32_t* new_row_begin) {
  if (n_bands == 0u || !row_begin || !cost || !new_row_begin) return RT_ERR_INVALID;
  for (uint32_t k = 0; k < n_bands; ++k) if (row_begin[k + 1] <= row_begin[k]) return RT_ERR_INVALID;
  balance_rows(n_bands, row_begin, cost, granule, new_row_begin);
  return RT_OK;
}

int rt_tracer_set_band(rt_tracer* t, uint32_t row_begin, uint32_t rows) {
  if (!t || t->mg || rows == 0u) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] {
    t->cancel_and_join();
    if (t->grp) throw HipFail{"rt_tracer_set_band: leave the group first"};
    if (!t->band_mode) throw HipFail{"rt_tracer_set_band: the tracer owns a whole image, not a band (rt_options.full_height)"};
    if (static_cast<uint64_t>(row_begin) + rows > t->H) throw HipFail{"row band exceeds full_height"};
    t->use_device();
    t->reshape(t->W, t->H, row_begin, rows);
  });
}

int rt_tracer_rebalance(rt_tracer* t) {
  if (!t || !t->mg) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] {
    t->cancel_and_join();
    multi_sync_all(t);
    MultiState& m = *t->mg;
    const uint32_t n = static_cast<uint32_t>(m.bands.size());
    std::vector<uint32_t> begins(n + 1u), fresh(n + 1u);
    std::vector<double> cost(n);
    for (uint32_t k = 0; k < n; ++k) {
      begins[k] = m.group.bands[k].row0;
      std::lock_guard<std::mutex> tl(m.bands[k]->time_mu);
      cost[k] = m.bands[k]->kernel_launches ? m.bands[k]->span_ms / static_cast<double>(m.bands[k]->kernel_launches) : 0.0;
    }
    begins[n] = t->H;
    for (uint32_t k = 0; k < n; ++k) if (!(cost[k] > 0.0)) throw HipFail{"rt_tracer_rebalance: no timed launch on every band yet"};
    balance_rows(n, begins.data(), cost.data(), 8u, fresh.data());
    if (fresh == begins) return;
    multi_resize(t, t->W, t->H, fresh.data());
  });
}

int rt_tracer_leave_group(rt_tracer* t) {
  if (!t || t->mg) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] {
    t->cancel_and_join();
    member_leave(t);
  });
}

int rt_tracer_gather_time(rt_tracer* t, double* total_ms, uint64_t* gathers, int reset_after) {
  if (!t) return RT_ERR_INVALID;
  Group* g = t->mg ? &t->mg->group : t->grp;
  if (total_ms) *total_ms = 0.0;
  if (gathers) *gathers = 0u;
  if (g) g->read_time(total_ms, gathers, reset_after != 0);
  return RT_OK;
}

int rt_tracer_gather_only(rt_tracer* t) {
  if (!t || (!t->mg && !t->grp)) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] {
    t->cancel_and_join();
    group_gather_only(t);
  });
}

int rt_tracer_group_info(rt_tracer* t, char* json, size_t capacity) {
  if (!t || !json || capacity == 0) return RT_ERR_INVALID;
  const std::string s = group_info_json(t);
  if (s.size() + 1 > capacity) return RT_ERR_INVALID;
  memcpy(json, s.c_str(), s.size() + 1);
  return RT_OK;
}

int rt_tracer_band_count(rt_tracer* t) {
  if (!t) return 0;
  return t->mg ? static_cast<int>(t->mg->bands.size()) : 1;
}

int rt_tracer_band_info(rt_tracer* t, uint32_t band, uint32_t out[4]) {
  if (!t || !out) return RT_ERR_INVALID;
  if (t->mg) {
    if (band >= t->mg->bands.size()) return RT_ERR_INVALID;
    const GroupBand& b = t->mg->group.bands[band];
    out[0] = static_cast<uint32_t>(t->mg->band_device[band]); out[1] = b.row0; out[2] = b.rows; out[3] = static_cast<uint32_t>(b.rank);
    return RT_OK;
  }
  if (band != 0u) return RT_ERR_INVALID;
  out[0] = static_cast<uint32_t>(t->device); out[1] = t->row0; out[2] = t->rows;
  out[3] = t->grp ? static_cast<uint32_t>(t->grp->local[0].rank) : 0u;
  return RT_OK;
}

// ---- single-function device harnesses -----------------------------------------------------

int rt_dbg_hit_triangle(int device, uint32_t math_mode, uint32_t n, const float* rays, const float* tris,
                        int eps_mode, int32_t* hit, float* tuv, float* normal, float* point) {
  int rc = require_device(device);
  if (rc != RT_OK) return rc;
  return guarded(nullptr, [&] {
    HIP_CHECK(hipSetDevice(device));
    DevBuf dr(n * 6 * sizeof(float)), dt(n * 9 * sizeof(float)), dh(n * sizeof(int)),
        duv(n * 3 * sizeof(float)), dn(n * 3 * sizeof(float)), dp(n * 3 * sizeof(float));
    HIP_CHECK(hipMemcpy(dr.p, rays, n * 6 * sizeof(float), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dt.p, tris, n * 9 * sizeof(float), hipMemcpyHostToDevice));
    HIP_CHECK(rtk::launch_dbg_hit_triangle(math_mode != RT_MATH_STRICT, n, dr.as<float>(), dt.as<float>(),
                                           eps_mode, dh.as<int>(), duv.as<float>(), dn.as<float>(),
                                           dp.as<float>(), nullptr));
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(hit, dh.p, n * sizeof(int), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(tuv, duv.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(normal, dn.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(point, dp.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost));
  });
}

int rt_dbg_check_midrange(int device, uint64_t out[4]) {
  if (!out) return RT_ERR_INVALID;
  int rc = require_device(device);
  if (rc != RT_OK) return rc;
  return guarded(nullptr, [&] {
    HIP_CHECK(hipSetDevice(device));
    DevBuf d(4 * sizeof(unsigned long long));
    HIP_CHECK(hipMemset(d.p, 0, 4 * sizeof(unsigned long long)));
    HIP_CHECK(rtk::launch_dbg_check_midrange(d.as<unsigned long long>(), nullptr));
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(out, d.p, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  });
}

int rt_dbg_valu_peak(int device, double* lane_fma_per_s, double* clock_ghz) {
  int rc = require_device(device);
  if (rc != RT_OK) return rc;
  return guarded(nullptr, [&] {
    HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    const uint32_t blocks = static_cast<uint32_t>(prop.multiProcessorCount) * 8u;   // 8 waves per SIMD
    const int iters = 20000;
    DevBuf out(static_cast<size_t>(blocks) * 256 * sizeof(float)), clk(2 * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(rtk::launch_dbg_valu_peak(blocks, iters, out.as<float>(), clk.as<unsigned long long>(), nullptr));
    HIP_CHECK(hipEventRecord(e0, nullptr));
    HIP_CHECK(rtk::launch_dbg_valu_peak(blocks, iters, out.as<float>(), clk.as<unsigned long long>(), nullptr));
    HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c[2];
    HIP_CHECK(hipMemcpy(c, clk.p, sizeof c, hipMemcpyDeviceToHost));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (lane_fma_per_s) *lane_fma_per_s = static_cast<double>(blocks) * 256.0 * iters * 8.0 / (ms * 1e-3);
    if (clock_ghz) *clock_ghz = c[1] ? static_cast<double>(c[0]) / static_cast<double>(c[1]) * 0.1 : 0.0;
  });
}

#ifdef RT_TIMELINE
// experiment builds only (make EXTRA=-DRT_TIMELINE): one launch with per-wave timestamps
extern "C" int rt_dbg_trace_timeline(rt_tracer* t, uint32_t samples, unsigned long long* out, size_t capacity_words) {
  if (!t || !out) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] {
    t->cancel_and_join();
    t->use_device();
    const size_t words = static_cast<size_t>((t->W + 31u) / 32u) * ((t->rows + 7u) / 8u) * 4u * 16u;
    if (words > capacity_words) throw HipFail{fmt("timeline needs %zu words", words)};
    DevBuf buf(words * sizeof(unsigned long long));
    HIP_CHECK(hipMemsetAsync(buf.p, 0, words * sizeof(unsigned long long), t->main_stream()));
    rtk::TraceParams p = t->params(samples);
    p.flags = rtk::TRACE_ZERO_ACC | rtk::TRACE_EMIT_IMAGE | t->mode_flags(p);
    p.image = t->d_image;
    p.timeline = buf.as<unsigned long long>();
    bool have_lists = false;
    const bool build_lists = t->prepare_tile_lists(p, true, have_lists);
    t->sync_list_stream();
    t->attach_tile_lists(p, have_lists);
    if (have_lists) HIP_CHECK(rtk::launch_tile_lists(p, t->fma, t->main_stream()));
    (void)build_lists;
    t->attach_macro_lists(p, 0, t->main_stream());
    HIP_CHECK(rtk::launch_trace(p, t->fma, t->filter, t->bin, t->pick_k(samples), t->main_stream()));
    HIP_CHECK(hipStreamSynchronize(t->main_stream()));
    HIP_CHECK(hipMemcpy(out, buf.p, words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  });
}
#endif

// the stored tile candidate lists of a small-scene tracer, as the last storing launch wrote them: per wave tile
// (grid order, 4 per 32x8 block) 1 + bin_list words; word 0 = count | winner << 10 | certain << 31.  Measurement aid.
int rt_dbg_read_tile_lists(rt_tracer* t, uint32_t* dst, size_t capacity_words, uint32_t* words_per_tile) {
  if (!t || t->mg || !dst) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] {
    t->cancel_and_join();
    t->use_device();
    HIP_CHECK(hipStreamSynchronize(t->main_stream()));
    t->sync_list_stream();
    if (!t->tile_lists_now() || !t->list_key_valid) throw HipFail{"no tile lists (small scenes build them ahead of their first trace launch)"};
    const size_t n = t->tile_lists_words < capacity_words ? t->tile_lists_words : capacity_words;
    HIP_CHECK(hipMemcpy(dst, t->tile_lists_now(), n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (words_per_tile) *words_per_tile = 1u + t->list_key.bin_list;
  });
}

// dense scenes: the header words (candidate count, 0xFFFFFFFF = overflow) of the tiles' lists in HBM as the last launch left
// them -- half 0 = the unsplit launch or the upper half of a split one, half 1 = the lower half.  Measurement aid / tests.
int rt_dbg_wave_list_counts(rt_tracer* t, int half, uint32_t* dst, size_t capacity_tiles, uint32_t* n_tiles, uint32_t* capacity_per_tile) {
  if (!t || t->mg || !dst || half < 0 || half > 1) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] {
    t->cancel_and_join();
    t->use_device();
    HIP_CHECK(hipStreamSynchronize(t->main_stream()));
    if (!t->d_wave_lists[half] || !t->wave_lists_valid[half]) throw HipFail{"no wave lists (dense scenes build them ahead of their first trace launch)"};
    const rt_tracer::ListKey& k = t->macro_key[half];
    const size_t tiles = static_cast<size_t>((k.W + 31u) / 32u) * ((k.rows + 7u) / 8u) * 4u;
    const uint32_t cap = t->params(1).bin_list;
    const size_t n = tiles < capacity_tiles ? tiles : capacity_tiles;
    HIP_CHECK(hipMemcpy2D(dst, sizeof(uint32_t), t->d_wave_lists[half], static_cast<size_t>(1u + cap) * 64u, sizeof(uint32_t), n, hipMemcpyDeviceToHost));
    if (n_tiles) *n_tiles = static_cast<uint32_t>(tiles);
    if (capacity_per_tile) *capacity_per_tile = cap;
  });
}

int rt_dbg_trace_occupancy(int device, int samples_in_flight, uint32_t lds_bytes) {
  if (require_device(device) != RT_OK) return -1;
  if (hipSetDevice(device) != hipSuccess) return -1;
  return rtk::trace_occupancy(samples_in_flight, lds_bytes);
}

int rt_dbg_sincos(int device, uint32_t n, const float* x, float* s, float* c) {
  int rc = require_device(device);
  if (rc != RT_OK) return rc;
  return guarded(nullptr, [&] {
    HIP_CHECK(hipSetDevice(device));
    DevBuf dx(n * sizeof(float)), ds(n * sizeof(float)), dc(n * sizeof(float));
    HIP_CHECK(hipMemcpy(dx.p, x, n * sizeof(float), hipMemcpyHostToDevice));
    HIP_CHECK(rtk::launch_dbg_sincos(n, dx.as<float>(), ds.as<float>(), dc.as<float>(), nullptr));
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(s, ds.p, n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(c, dc.p, n * sizeof(float), hipMemcpyDeviceToHost));
  });
}

int rt_dbg_uniform(int device, uint32_t n, uint32_t m, uint32_t* states, float* out) {
  int rc = require_device(device);
  if (rc != RT_OK) return rc;
  return guarded(nullptr, [&] {
    HIP_CHECK(hipSetDevice(device));
    DevBuf ds(n * 6 * sizeof(uint32_t)), dout(static_cast<size_t>(n) * m * sizeof(float));
    HIP_CHECK(hipMemcpy(ds.p, states, n * 6 * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_CHECK(rtk::launch_dbg_uniform(n, m, ds.as<uint32_t>(), dout.as<float>(), nullptr));
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(states, ds.p, n * 6 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out, dout.p, static_cast<size_t>(n) * m * sizeof(float), hipMemcpyDeviceToHost));
  });
}

int rt_dbg_get_ray(rt_tracer* t, uint32_t n, const uint32_t* pixels, uint32_t* states, float* rays) {
  if (!t) return RT_ERR_INVALID;
  if (t->mg) { multi_push_camera(t); return rt_dbg_get_ray(t->mg->bands[0], n, pixels, states, rays); }   // the camera of the whole frame
  return guarded(t, [&] {
    t->use_device();
    rtk::TraceParams p = t->params(1);
    DevBuf dpix(n * 2 * sizeof(uint32_t)), ds(n * 6 * sizeof(uint32_t)), dr(n * 6 * sizeof(float));
    HIP_CHECK(hipMemcpy(dpix.p, pixels, n * 2 * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(ds.p, states, n * 6 * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_CHECK(rtk::launch_dbg_get_ray(t->fma, p, n, dpix.as<uint32_t>(), ds.as<uint32_t>(), dr.as<float>(),
                                      t->main_stream()));
    HIP_CHECK(hipStreamSynchronize(t->main_stream()));
    HIP_CHECK(hipMemcpy(states, ds.p, n * 6 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(rays, dr.p, n * 6 * sizeof(float), hipMemcpyDeviceToHost));
  });
}

int rt_dbg_focal_boxes(rt_tracer* t, float curv_scale, float* boxes, size_t boxes_capacity, float* focal, size_t focal_capacity) {
  if (!t || t->mg || !boxes || !focal) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] {
    t->cancel_and_join();
    t->use_device();
    rtk::TraceParams p = t->params(1);
    if (p.tile_curv > 0.0f) p.tile_curv = std::max(p.tile_curv * curv_scale, 1e-30f);   // this launch only; the corner path stays on (the test's teeth: 0 must fail)
    const size_t tiles = static_cast<size_t>((t->W + 31u) / 32u) * ((t->rows + 7u) / 8u) * 4u;
    const size_t nb = tiles * 8u, nf = static_cast<size_t>(t->npix()) * 3u;
    if (nb > boxes_capacity || nf > focal_capacity) throw HipFail{fmt("focal boxes need %zu + %zu floats", nb, nf)};
    DevBuf db(nb * sizeof(float)), df(nf * sizeof(float));
    HIP_CHECK(hipMemsetAsync(db.p, 0, nb * sizeof(float), t->main_stream()));
    HIP_CHECK(rtk::launch_dbg_focal_boxes(t->fma, p, db.as<float>(), df.as<float>(), t->main_stream()));
    HIP_CHECK(hipStreamSynchronize(t->main_stream()));
    HIP_CHECK(hipMemcpy(boxes, db.p, nb * sizeof(float), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(focal, df.p, nf * sizeof(float), hipMemcpyDeviceToHost));
  });
}

int rt_dbg_classify(rt_tracer* t, uint32_t level, uint32_t forms, uint32_t slack_milli, const uint32_t* regions, uint32_t n_regions,
                    float* out, size_t capacity_floats) {
  if (!t || t->mg || !regions || !out || level > 4u) return RT_ERR_INVALID;
  std::lock_guard<std::mutex> lk(t->api_mu);
  return guarded(t, [&] {
    t->cancel_and_join();
    t->use_device();
    rtk::TraceParams p = t->params(1);
    p.macro_w = rt_tracer::kMacroW; p.macro_h = rt_tracer::kMacroH;      // level 2: the macro tile of attach_macro_lists
    p.super_f = rt_tracer::kSuperF;                                       // level 4: the super tile above it
    const uint32_t rw = level == 0u ? 8u : level == 2u ? p.macro_w : level == 4u ? p.macro_w * p.super_f : 32u;
    const uint32_t rh = level == 2u ? p.macro_h : level == 4u ? p.macro_h * p.super_f : level == 3u ? 16u : 8u;
    for (uint32_t i = 0; i < n_regions; ++i)                             // the kernel's pixel <-> lane mapping assumes the trace grid
      if (regions[2u * i] % rw != 0u || regions[2u * i + 1u] % rh != 0u || regions[2u * i] >= t->W || regions[2u * i + 1u] >= t->rows)
        throw HipFail{fmt("region %u (%u, %u) is not a level-%u region of the %ux%u band", i, regions[2u * i], regions[2u * i + 1u], level, t->W, t->rows)};
    const size_t per = 16u + static_cast<size_t>(t->n_tris) * (forms ? 32u : 12u);
    if (per * n_regions > capacity_floats) throw HipFail{fmt("rt_dbg_classify needs %zu floats", per * n_regions)};
    DevBuf dr(static_cast<size_t>(n_regions) * 2u * sizeof(uint32_t)), dout(per * n_regions * sizeof(float));
    HIP_CHECK(hipMemcpyAsync(dr.p, regions, static_cast<size_t>(n_regions) * 2u * sizeof(uint32_t), hipMemcpyHostToDevice, t->main_stream()));
    HIP_CHECK(rtk::launch_dbg_classify(t->fma, forms != 0u, slack_milli, p, level, n_regions, dr.as<uint32_t>(), dout.as<float>(), t->main_stream()));
    HIP_CHECK(hipStreamSynchronize(t->main_stream()));
    HIP_CHECK(hipMemcpy(out, dout.p, per * n_regions * sizeof(float), hipMemcpyDeviceToHost));
  });
}

void rt_dbg_rng_init_host(uint64_t seed, uint64_t subsequence, uint32_t state[6]) {
  rth::init_state(jump_host(), seed, subsequence & 0xffffffffull, state);
}

}  // extern "C"
