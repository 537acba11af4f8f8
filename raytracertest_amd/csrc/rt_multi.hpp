// rt_multi.hpp -- one frame sharded in contiguous row bands over several GPUs.
// Included by rt_tracer.hip behind the definition of struct rt_tracer (same translation unit).
//
// The reference drives ONE rt::RayTracer on device 0 (OpenGLView/MainFrame.cpp:44-45,
// OpenGLView/GLCanvas.cpp:259-260).  Pixels are independent and a pixel's RNG stream is keyed by its
// GLOBAL index (Random.cu:21-27), so a frame splits into row bands with no exchange while tracing; the
// only data movement is the hand-off the reference does through its callback (RayTracerImpl.cu:287-305):
// the finished BGRA8 tiles travel to the root device -- an RCCL gather over xGMI (grouped ncclSend /
// ncclRecv: bands may be ragged and several bands may share a device) -- and from there to the host.
//
//   Group       who owns which band, the communicator(s), the gather itself, double-buffered:
//               a band's emitting launch writes its tile straight into the gather's buffer
//               (TraceParams::image_host): on the root device that IS the gathered frame (no copy at
//               all), elsewhere the send buffer.  Frame i is gathered on per-device gather streams
//               while the trace streams already work on frame i+1.
//   MultiState  the in-process form (rt_tracer_create_multi): one band tracer per band, one host
//               thread per device for the launches, the render thread + callbacks of
//               RayTracerImpl::TraceFunct (:236-315) over the whole frame.
//   a band tracer that joined a multi-PROCESS group (rt_tracer_join_group: one process per GPU, the
//   unique id travels through the launcher's own rendezvous) uses the same Group with one local rank.
#pragma once
#include <condition_variable>
#include <functional>

#include "rt_rccl.hpp"

namespace {

#define RCCL_CHECK(expr)                                                                              \
  do {                                                                                                \
    ncclResult_t r_ = (expr);                                                                         \
    if (r_ != ncclSuccess) throw HipFail{fmt("%s failed: %s", #expr, rtc::Rccl::get().GetErrorString(r_))}; \
  } while (0)

rtc::Rccl& need_rccl() {
  rtc::Rccl& r = rtc::Rccl::get();
  if (!r.ok()) throw HipFail{"RCCL is needed for a frame sharded over several devices and could not be loaded: " + r.why};
  return r;
}

// rows [begin, begin + count) of band k of n over `height` rows: contiguous, balanced to one row
inline void band_rows(uint32_t height, uint32_t n, uint32_t k, uint32_t& begin, uint32_t& count) {
  begin = static_cast<uint32_t>(static_cast<uint64_t>(height) * k / n);
  count = static_cast<uint32_t>(static_cast<uint64_t>(height) * (k + 1u) / n) - begin;
}

struct GatherRank {                 // one rank of the communicator that lives in this process
  int rank = 0, device = 0;
  ncclComm_t comm = nullptr;        // null: the group has a single rank and nothing to exchange
  hipStream_t gstream = nullptr;    // this rank's gather stream
};

struct GroupBand {                  // one row band of the frame
  uint32_t row0 = 0, rows = 0;
  int rank = 0;                     // communicator rank that owns it
  rt_tracer* tracer = nullptr;      // local bands only
  uint32_t* send[2] = {nullptr, nullptr};      // local bands that travel: BGRA8 send buffers on the owner's device
  hipEvent_t ready = nullptr;                  // local: the tile of the current frame is written
  hipEvent_t sent[2] = {nullptr, nullptr};     // local travelling bands: the send out of buffer b has finished
  bool sent_valid[2] = {false, false};
};

struct Group {
  uint32_t W = 0, H = 0;
  int n_ranks = 1;
  std::vector<GroupBand> bands;     // every band of the frame, ascending rows (geometry of remote ones included)
  std::vector<GatherRank> local;    // ranks of this process; the root (rank 0), when local, is local[0]
  bool has_root = false;
  bool self_rccl = false;           // RT_MI355X_GATHER_SELF=1: root-local bands travel through RCCL too (1-GPU rehearsal of the call sequence)
  // Peer transport (rt_options.transport = RT_TRANSPORT_PEER, one process only): every band's emitting launch writes its
  // BGRA8 tile straight into its rows of the ROOT's frame buffer through a peer mapping (hipDeviceEnablePeerAccess) --
  // posted xGMI writes spread over the kernel's lifetime, no send buffers, no collective, no receive kernels on the root's
  // CUs; the "gather" is the root's gather stream waiting for every band's event.  peer_note: why it was not granted.
  bool peer = false;
  std::string peer_note;
  uint32_t* d_frame[2] = {nullptr, nullptr};   // root: the gathered frames
  hipEvent_t frame_done[2] = {nullptr, nullptr};   // root: frame b is complete (gather stream)
  hipEvent_t frame_free[2] = {nullptr, nullptr};   // root: the consumer of frame b (host copy) has finished
  bool frame_free_valid[2] = {false, false};
  int next_b = 0, last_b = -1;
  // gather() runs on whoever drives the frames (render thread, or an API thread under the handle's api_mu);
  // sync() and the timing read-out may come from any thread meanwhile: mu serialises them
  std::mutex mu;
  // gather timing (root): event pairs on the root's gather stream around the exchange
  struct Timed { hipEvent_t a, b; };
  std::vector<Timed> timed_pending, timed_free;
  double gather_ms = 0.0;
  uint64_t gathers = 0;

  GatherRank* rank_local(int rank) {
    for (GatherRank& r : local) if (r.rank == rank) return &r;
    return nullptr;
  }
  bool travels(const GroupBand& b) const { return !peer && (b.rank != 0 || self_rccl); }
  size_t frame_bytes() const { return static_cast<size_t>(W) * H * sizeof(uint32_t); }

  // Device-side resources for the current geometry (bands[] filled in, local[] with rank/device/comm).
  void allocate() {
    for (GatherRank& r : local) {
      HIP_CHECK(hipSetDevice(r.device));
      if (!r.gstream) HIP_CHECK(hipStreamCreateWithFlags(&r.gstream, hipStreamNonBlocking));
    }
    for (GroupBand& b : bands) {
      if (!b.tracer) continue;
      GatherRank* r = rank_local(b.rank);
      HIP_CHECK(hipSetDevice(r->device));
      HIP_CHECK(hipEventCreateWithFlags(&b.ready, hipEventDisableTiming));
      if (travels(b)) {
        for (int i = 0; i < 2; ++i) {
          HIP_CHECK(hipMalloc(&b.send[i], static_cast<size_t>(b.rows) * W * sizeof(uint32_t)));
          HIP_CHECK(hipMemset(b.send[i], 0, static_cast<size_t>(b.rows) * W * sizeof(uint32_t)));
          HIP_CHECK(hipEventCreateWithFlags(&b.sent[i], hipEventDisableTiming));
          b.sent_valid[i] = false;
        }
      }
    }
    if (has_root) {
      HIP_CHECK(hipSetDevice(local[0].device));
      for (int i = 0; i < 2; ++i) {
        HIP_CHECK(hipMalloc(&d_frame[i], frame_bytes()));
        HIP_CHECK(hipMemset(d_frame[i], 0, frame_bytes()));
        HIP_CHECK(hipEventCreateWithFlags(&frame_done[i], hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&frame_free[i], hipEventDisableTiming));
        frame_free_valid[i] = false;
      }
    }
    // The memsets above ran on the null stream, which the tracers' non-blocking streams are not ordered with: make sure they
    // have landed before any trace kernel writes a tile into these buffers.
    for (GatherRank& r : local) {
      HIP_CHECK(hipSetDevice(r.device));
      HIP_CHECK(hipDeviceSynchronize());
    }
    next_b = 0; last_b = -1;
  }

  void release_buffers() {          // everything allocate() made except the gather streams (callers have synchronised)
    for (GroupBand& b : bands) {
      if (!b.tracer) continue;
      if (GatherRank* r = rank_local(b.rank)) (void)hipSetDevice(r->device);
      if (b.ready) { (void)hipEventDestroy(b.ready); b.ready = nullptr; }
      for (int i = 0; i < 2; ++i) {
        if (b.send[i]) { (void)hipFree(b.send[i]); b.send[i] = nullptr; }
        if (b.sent[i]) { (void)hipEventDestroy(b.sent[i]); b.sent[i] = nullptr; }
      }
    }
    if (has_root && !local.empty()) {
      (void)hipSetDevice(local[0].device);
      for (int i = 0; i < 2; ++i) {
        if (d_frame[i]) { (void)hipFree(d_frame[i]); d_frame[i] = nullptr; }
        if (frame_done[i]) { (void)hipEventDestroy(frame_done[i]); frame_done[i] = nullptr; }
        if (frame_free[i]) { (void)hipEventDestroy(frame_free[i]); frame_free[i] = nullptr; }
      }
      for (Timed& t : timed_pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
      for (Timed& t : timed_free) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
      timed_pending.clear(); timed_free.clear();
    }
  }

  void destroy() {
    release_buffers();
    for (GatherRank& r : local) {
      (void)hipSetDevice(r.device);
      if (r.comm) { (void)rtc::Rccl::get().CommDestroy(r.comm); r.comm = nullptr; }
      if (r.gstream) { (void)hipStreamDestroy(r.gstream); r.gstream = nullptr; }
    }
  }

  // Where band k's emitting launch of the next frame writes its tile; orders the band's streams behind the
  // last reader of that buffer.  Call once per local band before its launch; `b` from begin_frame().
  int begin_frame() { return next_b; }
  uint32_t* tile_target(size_t k, int b) {
    GroupBand& band = bands[k];
    rt_tracer* t = band.tracer;
    HIP_CHECK(hipSetDevice(t->device));
    if (travels(band)) {
      if (band.sent_valid[b]) HIP_CHECK(hipStreamWaitEvent(t->main_stream(), band.sent[b], 0));
      return band.send[b];
    }
    if (frame_free_valid[b]) HIP_CHECK(hipStreamWaitEvent(t->main_stream(), frame_free[b], 0));
    return d_frame[b] + static_cast<size_t>(band.row0) * W;
  }
  // band k's launch of this frame is enqueued: its tile is complete behind both of its streams
  void tile_written(size_t k) {
    GroupBand& band = bands[k];
    HIP_CHECK(hipSetDevice(band.tracer->device));
    HIP_CHECK(hipEventRecord(band.ready, band.tracer->main_stream()));
  }

  // The exchange of frame b, enqueued on the gather streams behind every local tile: grouped
  // ncclSend (owners) / ncclRecv (root) of the travelling bands, one group for all local ranks.
  void gather(int b) {
    std::lock_guard<std::mutex> lk(mu);
    for (GatherRank& r : local) {
      HIP_CHECK(hipSetDevice(r.device));
      for (GroupBand& band : bands)
        if (band.tracer && (band.rank == r.rank || (peer && r.rank == 0)))   // peer: the root waits for every band's stores
          HIP_CHECK(hipStreamWaitEvent(r.gstream, band.ready, 0));
    }
    bool any = false;
    for (const GroupBand& band : bands) any = any || travels(band);
    Timed tm{};
    const bool timed = has_root && any;
    if (timed) {
      HIP_CHECK(hipSetDevice(local[0].device));
      reap_timed();
      if (!timed_free.empty()) { tm = timed_free.back(); timed_free.pop_back(); }
      else { HIP_CHECK(hipEventCreate(&tm.a)); HIP_CHECK(hipEventCreate(&tm.b)); }
      HIP_CHECK(hipEventRecord(tm.a, local[0].gstream));
    }
    if (any) {
      rtc::Rccl& nccl = need_rccl();
      RCCL_CHECK(nccl.GroupStart());
      for (GroupBand& band : bands) {
        if (!travels(band)) continue;
        const size_t count = static_cast<size_t>(band.rows) * W;
        if (band.tracer) {                                            // owner: send to the root
          GatherRank* r = rank_local(band.rank);
          RCCL_CHECK(nccl.Send(band.send[b], count, ncclUint32, 0, r->comm, r->gstream));
        }
        if (has_root)                                                 // root: receive into the band's rows of the frame
          RCCL_CHECK(nccl.Recv(d_frame[b] + static_cast<size_t>(band.row0) * W, count, ncclUint32, band.rank,
                               local[0].comm, local[0].gstream));
      }
      RCCL_CHECK(nccl.GroupEnd());
    }
    for (GroupBand& band : bands) {
      if (!band.tracer || !travels(band)) continue;
      GatherRank* r = rank_local(band.rank);
      HIP_CHECK(hipSetDevice(r->device));
      HIP_CHECK(hipEventRecord(band.sent[b], r->gstream));
      band.sent_valid[b] = true;
    }
    if (has_root) {
      HIP_CHECK(hipSetDevice(local[0].device));
      if (timed) { HIP_CHECK(hipEventRecord(tm.b, local[0].gstream)); timed_pending.push_back(tm); }
      HIP_CHECK(hipEventRecord(frame_done[b], local[0].gstream));
    }
    last_b = b;
    next_b = b ^ 1;
  }

  void reap_timed(bool all = false) {
    size_t n = 0;
    while (n < timed_pending.size() && (all || hipEventQuery(timed_pending[n].b) == hipSuccess)) {
      float ms = 0.0f;
      if (hipEventElapsedTime(&ms, timed_pending[n].a, timed_pending[n].b) == hipSuccess) { gather_ms += ms; gathers += 1; }
      timed_free.push_back(timed_pending[n]);
      ++n;
    }
    (void)hipGetLastError();
    timed_pending.erase(timed_pending.begin(), timed_pending.begin() + static_cast<std::ptrdiff_t>(n));
  }

  void sync() {                     // every gather stream of this process
    std::lock_guard<std::mutex> lk(mu);
    for (GatherRank& r : local) {
      HIP_CHECK(hipSetDevice(r.device));
      HIP_CHECK(hipStreamSynchronize(r.gstream));
    }
    if (has_root) { HIP_CHECK(hipSetDevice(local[0].device)); reap_timed(true); }
  }
  void read_time(double* total_ms, uint64_t* n, bool reset) {
    std::lock_guard<std::mutex> lk(mu);
    if (total_ms) *total_ms = gather_ms;
    if (n) *n = gathers;
    if (reset) { gather_ms = 0.0; gathers = 0; }
  }
};

// One host thread per device: the launches of a frame's bands are enqueued concurrently (a launch costs
// ~15 us of host time per band; eight bands issued by one thread would starve a 150 us step).
class WorkerPool {
 public:
  explicit WorkerPool(size_t n) : n_(n) {
    for (size_t i = 0; i < n; ++i) threads_.emplace_back([this, i] { loop(i); });
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> lk(mu_); quit_ = true; }
    cv_.notify_all();
    for (std::thread& t : threads_) t.join();
  }
  // f(i) on worker i for every i; returns when all have finished; the first failure is rethrown here
  void run(const std::function<void(size_t)>& f) {
    std::unique_lock<std::mutex> lk(mu_);
    fn_ = &f; remaining_ = n_; failed_ = false; ++generation_;
    cv_.notify_all();
    done_cv_.wait(lk, [&] { return remaining_ == 0; });
    fn_ = nullptr;
    if (failed_) throw HipFail{error_};
  }

 private:
  void loop(size_t i) {
    uint64_t seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return quit_ || generation_ != seen; });
      if (quit_) return;
      seen = generation_;
      const std::function<void(size_t)>* f = fn_;
      lk.unlock();
      std::string err;
      try { (*f)(i); }
      catch (const HipFail& e) { err = e.what; }
      catch (const std::exception& e) { err = e.what(); }
      catch (...) { err = "unknown failure in a device worker"; }
      lk.lock();
      if (!err.empty() && !failed_) { failed_ = true; error_ = err; }
      if (--remaining_ == 0) done_cv_.notify_one();
    }
  }
  size_t n_;
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(size_t)>* fn_ = nullptr;
  size_t remaining_ = 0;
  uint64_t generation_ = 0;
  bool quit_ = false, failed_ = false;
  std::string error_;
};

// State of a multi-device tracer (the handle's own fields hold the whole frame's W/H, the camera, the
// callbacks, the render thread and the error text).
struct MultiState {
  Group group;
  std::vector<rt_tracer*> bands;            // owned band tracers, group.bands[k].tracer == bands[k]
  std::vector<int> band_device;             // device ordinal per band
  std::vector<int> devices;                 // distinct devices, devices[0] = root
  std::vector<std::vector<size_t>> bands_of_device;
  std::unique_ptr<WorkerPool> pool;         // null with a single device
  uint32_t* h_image[2] = {nullptr, nullptr};   // pinned whole-frame host images handed to the callbacks
  hipEvent_t handoff[2] = {nullptr, nullptr};  // host image b is complete
  rt_options opt{};                         // creation options (band tracers are re-created with them on Resize)

  // f(band index) for every band, bands of one device in order on that device's thread
  void for_bands(const std::function<void(size_t)>& f) {
    if (!pool) { for (size_t k = 0; k < bands.size(); ++k) f(k); return; }
    pool->run([&](size_t d) { for (size_t k : bands_of_device[d]) f(k); });
  }
};

}  // namespace
