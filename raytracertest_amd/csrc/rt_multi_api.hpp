// rt_multi_api.hpp -- the entry points of include/rt_mi355x.h for a multi-device tracer (t->mg) and for a
// band tracer inside a multi-process group (t->grp).  Included by rt_tracer.hip in front of its extern "C"
// block; the C functions there dispatch here when the handle is one of the two.
#pragma once

namespace {

bool env_on(const char* name) { const char* e = getenv(name); return e && e[0] == '1'; }

// ---- helpers over all bands ---------------------------------------------------------------------
void multi_push_camera(rt_tracer* t) {          // the frame's camera, snapshotted once, to every band
  Camera c;
  { std::lock_guard<std::mutex> lk(t->state_mu); c = t->cam; }
  for (rt_tracer* b : t->mg->bands) { std::lock_guard<std::mutex> lk(b->state_mu); b->cam = c; }
}

void multi_sync_all(rt_tracer* t) {
  MultiState& m = *t->mg;
  for (rt_tracer* b : m.bands) {
    b->use_device();
    const uint64_t seen = b->event_seq_now();        // a running render thread may enqueue more meanwhile
    HIP_CHECK(hipStreamSynchronize(b->main_stream()));
    b->sync_list_stream();
    b->stagger_next = true;
    b->drain_events_before(seen);
  }
  m.group.sync();
}

// frame b is gathered: copy it to the pinned host image b for the callbacks, then the frame buffer is free
void multi_frame_to_host(rt_tracer* t, int b) {
  MultiState& m = *t->mg;
  Group& g = m.group;
  HIP_CHECK(hipSetDevice(g.local[0].device));
  HIP_CHECK(hipMemcpyAsync(m.h_image[b], g.d_frame[b], g.frame_bytes(), hipMemcpyDeviceToHost, g.local[0].gstream));
  HIP_CHECK(hipEventRecord(m.handoff[b], g.local[0].gstream));
  HIP_CHECK(hipEventRecord(g.frame_free[b], g.local[0].gstream));
  g.frame_free_valid[b] = true;
}

// one Trace pass, device resident, of every band + the gather of the finished frame (rt_tracer_trace_enqueue)
void multi_trace_enqueue(rt_tracer* t, uint32_t iterationCount, uint32_t samplesPerIteration) {
  MultiState& m = *t->mg;
  Group& g = m.group;
  multi_push_camera(t);
  const int b = g.begin_frame();
  m.for_bands([&](size_t k) {
    rt_tracer* band = m.bands[k];
    band->use_device();
    band->trace_enqueue_body(iterationCount, samplesPerIteration, g.tile_target(k, b));
    g.tile_written(k);
  });
  g.gather(b);
}

void multi_launch(rt_tracer* t, uint32_t samples, uint32_t iterations, bool clear_first, bool emit) {
  MultiState& m = *t->mg;
  Group& g = m.group;
  multi_push_camera(t);
  const int b = emit ? g.begin_frame() : -1;
  m.for_bands([&](size_t k) {
    rt_tracer* band = m.bands[k];
    band->use_device();
    band->enqueue_trace_launch(samples, (clear_first ? rtk::TRACE_ZERO_ACC : 0u) | (emit ? rtk::TRACE_EMIT_IMAGE : 0u), 0,
                               iterations, emit ? g.tile_target(k, b) : nullptr);
    if (emit) g.tile_written(k);
  });
  if (emit) g.gather(b);
}

// RayTracerImpl::TraceFunct (RayTracerImpl.cu:236-315) over a frame in bands: the same loop, launch fusion and
// pipelined update hand-off as rt_tracer::trace_funct; an update or the end of the Trace gathers the tiles to
// the root device, copies the frame to pinned host memory and fires ONE callback with the whole frame.
void multi_trace_funct(rt_tracer* t, uint32_t iterationCount, uint32_t samplesPerIteration, uint32_t updateInterval) {
  MultiState& m = *t->mg;
  Group& g = m.group;
  try {
    const size_t bytes = g.frame_bytes();
    bool cleared = false;                                               // :242-243, fused into launch 0
    struct { bool due = false; int b = 0; rt_callback_fn cb = nullptr; void* user = nullptr; } pend;
    auto deliver = [&] {
      if (!pend.due) return;
      HIP_CHECK(hipSetDevice(g.local[0].device));
      HIP_CHECK(hipEventSynchronize(m.handoff[pend.b]));               // :259
      pend.cb(m.h_image[pend.b], bytes, pend.user);                    // :272
      pend.due = false;
    };
    int final_b = -1;
    uint32_t i = 0;
    const uint32_t fuse = m.bands[0]->fused_iterations(samplesPerIteration);
    while (!t->stopped && i < iterationCount) {                        // :246
      rt_callback_fn cb; void* user;
      { std::lock_guard<std::mutex> lk(t->state_mu); cb = t->update_cb; user = t->update_user; }
      auto is_update = [&](uint32_t k) { return cb != nullptr && k > 0 && updateInterval > 0 && k % updateInterval == 0; };   // :256
      const uint32_t last_allowed = iterationCount - 1u - i < fuse - 1u ? iterationCount - 1u : i + fuse - 1u;
      uint32_t e = i;                                                  // last iteration of this launch
      while (e < last_allowed && !is_update(e)) ++e;
      const bool update = is_update(e);
      const bool emit = update || e + 1 == iterationCount;
      const uint32_t flags = (cleared ? 0u : rtk::TRACE_ZERO_ACC) | (emit ? rtk::TRACE_EMIT_IMAGE : 0u);
      const int b = emit ? g.begin_frame() : -1;
      if (emit && pend.due && pend.b == b) deliver();                  // never overwrite a host image still to be handed out
      multi_push_camera(t);                                            // *mCamera by value, once per launch, :221
      m.for_bands([&](size_t k) {
        rt_tracer* band = m.bands[k];
        band->use_device();
        band->enqueue_trace_launch(samplesPerIteration, flags, rt_tracer::kWindow, e - i + 1u,
                                   emit ? g.tile_target(k, b) : nullptr, false);   // :249
        if (emit) g.tile_written(k);
      });
      cleared = true;
      if (emit) {
        g.gather(b);
        multi_frame_to_host(t, b);
        final_b = b;
      }
      deliver();                                                       // the previous update, while these launches run
      if (update) { pend.due = true; pend.b = b; pend.cb = cb; pend.user = user; }
      i = e + 1u;
    }
    deliver();
    if (!cleared) {                                                    // no launch ran: plain clear (+ convert)
      const bool conv = !t->stopped;
      const int b = g.begin_frame();
      m.for_bands([&](size_t k) {
        rt_tracer* band = m.bands[k];
        band->use_device();
        band->clear_accumulators();
        if (conv) { band->convert(); band->copy_image_to(g.tile_target(k, b)); g.tile_written(k); }
      });
      if (conv) { g.gather(b); multi_frame_to_host(t, b); final_b = b; }
    }
    multi_sync_all(t);                                                 // :287 (and :280-284: a stopped run ends here)
    if (t->stopped) return;                                            // no callback
    t->completed = true;
    rt_callback_fn cb; void* user;
    { std::lock_guard<std::mutex> lk(t->state_mu); cb = t->finished_cb; user = t->finished_user; }
    if (cb != nullptr && final_b >= 0) cb(m.h_image[final_b], bytes, user);   // :302-305
  } catch (const HipFail& f) {                                         // :307-314 swallowed, but recorded
    t->set_error(f.what);
  } catch (...) {
    t->set_error("unknown failure in the render thread");
  }
}

// ---- geometry --------------------------------------------------------------------------------------
void multi_free_host_images(MultiState& m) {
  if (!m.group.local.empty()) (void)hipSetDevice(m.group.local[0].device);
  for (int i = 0; i < 2; ++i) {
    if (m.h_image[i]) { (void)hipHostFree(m.h_image[i]); m.h_image[i] = nullptr; }
  }
}

void multi_alloc_host_images(rt_tracer* t) {
  MultiState& m = *t->mg;
  HIP_CHECK(hipSetDevice(m.group.local[0].device));
  for (int i = 0; i < 2; ++i) {
    HIP_CHECK(hipHostMalloc(&m.h_image[i], m.group.frame_bytes(), hipHostMallocDefault));
    memset(m.h_image[i], 0, m.group.frame_bytes());
    if (!m.handoff[i]) HIP_CHECK(hipEventCreateWithFlags(&m.handoff[i], hipEventDisableTiming));
  }
}

void multi_destroy(rt_tracer* t) {
  MultiState* m = t->mg;
  if (!m) return;
  for (rt_tracer* b : m->bands) {                                      // quiesce before the gather buffers go
    if (!b) continue;
    (void)hipSetDevice(b->device);
    if (b->stream_b) (void)hipStreamSynchronize(b->stream_b);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
  }
  for (GatherRank& r : m->group.local) { (void)hipSetDevice(r.device); if (r.gstream) (void)hipStreamSynchronize(r.gstream); }
  m->pool.reset();
  multi_free_host_images(*m);
  if (!m->group.local.empty()) (void)hipSetDevice(m->group.local[0].device);
  for (int i = 0; i < 2; ++i) if (m->handoff[i]) (void)hipEventDestroy(m->handoff[i]);
  m->group.destroy();
  for (rt_tracer* b : m->bands) if (b) rt_tracer_destroy(b);
  delete m;
  t->mg = nullptr;
}

int multi_create(const uint32_t imageSize[2], const float cameraPosition[3], const float cameraAngles[2], float fov,
                 float focalLength, float aperture, const rt_options* options, const int32_t* devices, uint32_t n_bands,
                 rt_tracer** out) {
  if (!out) return RT_ERR_INVALID;
  *out = nullptr;
  if (!imageSize || !cameraAngles || imageSize[0] == 0 || imageSize[1] == 0 || !devices || n_bands == 0 ||
      n_bands > imageSize[1]) {
    set_global_error("rt_tracer_create_multi: invalid image size, camera angles, device list or band count (1..height)");
    return RT_ERR_INVALID;
  }
  rt_options opt;
  memset(&opt, 0, sizeof opt);
  opt.use_time_seed = 1;
  if (options) {
    const size_t n = options->struct_size < sizeof(opt) ? options->struct_size : sizeof(opt);
    if (n < 8) { set_global_error("rt_options.struct_size not set"); return RT_ERR_INVALID; }
    opt.use_time_seed = 0;
    memcpy(&opt, options, n);
  }
  if (opt.full_height != 0u) { set_global_error("rt_tracer_create_multi owns the whole frame: rt_options.full_height must be 0"); return RT_ERR_INVALID; }
  for (uint32_t k = 0; k < n_bands; ++k) {
    const int rc = require_device(devices[k]);
    if (rc != RT_OK) return rc;
  }
  opt.struct_size = sizeof(opt);
  opt.seed = opt.use_time_seed ? static_cast<uint64_t>(static_cast<uint32_t>(time(nullptr))) : opt.seed;   // one seed for every band
  opt.use_time_seed = 0;

  rt_tracer* t = new rt_tracer();
  MultiState* m = new MultiState();
  t->mg = m;
  m->opt = opt;
  t->W = imageSize[0]; t->H = imageSize[1]; t->row0 = 0; t->rows = t->H;
  t->seed = opt.seed;
  t->fma = opt.math_mode != RT_MATH_STRICT;
  Camera& c = t->cam;
  for (int i = 0; i < 3; ++i) c.position[i] = cameraPosition ? cameraPosition[i] : 0.0f;
  c.angles[0] = cameraAngles[0]; c.angles[1] = cameraAngles[1];
  c.fov = Camera::radians(fov); c.focal = focalLength; c.aperture = aperture;
  c.transform();

  Group& g = m->group;
  g.W = t->W; g.H = t->H;
  g.self_rccl = env_on("RT_MI355X_GATHER_SELF");
  for (uint32_t k = 0; k < n_bands; ++k) {                             // distinct devices in order of appearance: rank = index
    size_t d = 0;
    while (d < m->devices.size() && m->devices[d] != devices[k]) ++d;
    if (d == m->devices.size()) { m->devices.push_back(devices[k]); m->bands_of_device.emplace_back(); }
    m->bands_of_device[d].push_back(k);
    m->band_device.push_back(devices[k]);
  }
  t->device = m->devices[0];
  g.n_ranks = static_cast<int>(m->devices.size());
  g.has_root = true;
  const int rc = guarded(t, [&] {
    for (uint32_t k = 0; k < n_bands; ++k) {
      GroupBand gb;
      band_rows(t->H, n_bands, k, gb.row0, gb.rows);
      rt_options bo = opt;
      bo.device = devices[k]; bo.full_height = t->H; bo.row_begin = gb.row0;
      const uint32_t size[2] = {t->W, gb.rows};
      rt_tracer* band = nullptr;
      const int brc = rt_tracer_create_ex(size, cameraPosition, cameraAngles, fov, focalLength, aperture, &bo, &band);
      if (brc != RT_OK) throw HipFail{fmt("band %u on device %d: %s", k, devices[k], rt_last_error())};
      m->bands.push_back(band);
      size_t d = 0;
      while (m->devices[d] != devices[k]) ++d;
      gb.rank = static_cast<int>(d);
      gb.tracer = band;
      g.bands.push_back(gb);
    }
    for (size_t d = 0; d < m->devices.size(); ++d) {
      GatherRank r;
      r.rank = static_cast<int>(d); r.device = m->devices[d];
      g.local.push_back(r);
    }
    // transport: peer stores into the root's frame when asked for and every device may map the root's memory
    const bool want_peer = opt.transport == RT_TRANSPORT_PEER;
    if (want_peer && !g.self_rccl) {
      g.peer = true;
      for (size_t d = 1; d < m->devices.size() && g.peer; ++d) {
        int can = 0;
        const hipError_t e = hipDeviceCanAccessPeer(&can, m->devices[d], m->devices[0]);
        if (env_on("RT_MI355X_PEER_DENY")) can = 0;                  // (tests: the fallback path on any box)
        if (e != hipSuccess || !can) {
          g.peer = false;
          g.peer_note = fmt("device %d cannot map the memory of root device %d (%s): falling back to the RCCL gather", m->devices[d],
                            m->devices[0], e != hipSuccess ? hipGetErrorString(e) : "hipDeviceCanAccessPeer = 0");
        }
      }
      if (m->devices.size() == 1 && env_on("RT_MI355X_PEER_DENY")) { g.peer = false; g.peer_note = "peer access denied (RT_MI355X_PEER_DENY): falling back to the RCCL gather"; }
      for (size_t d = 1; d < m->devices.size() && g.peer; ++d) {
        HIP_CHECK(hipSetDevice(m->devices[d]));
        const hipError_t e = hipDeviceEnablePeerAccess(m->devices[0], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
          g.peer = false;
          g.peer_note = fmt("hipDeviceEnablePeerAccess(%d -> %d): %s: falling back to the RCCL gather", m->devices[d], m->devices[0], hipGetErrorString(e));
        }
        (void)hipGetLastError();
      }
    }
    if ((g.n_ranks > 1 && !g.peer) || g.self_rccl) {                   // one communicator per device (single-process clique)
      rtc::Rccl& nccl = need_rccl();
      std::vector<ncclComm_t> comms(m->devices.size(), nullptr);
      RCCL_CHECK(nccl.CommInitAll(comms.data(), static_cast<int>(m->devices.size()), m->devices.data()));
      for (size_t d = 0; d < comms.size(); ++d) g.local[d].comm = comms[d];
    }
    g.allocate();
    multi_alloc_host_images(t);
    if (m->devices.size() > 1) m->pool.reset(new WorkerPool(m->devices.size()));
  });
  if (rc != RT_OK) {
    const std::string why = t->last_error;
    multi_destroy(t);
    delete t;
    set_global_error("rt_tracer_create_multi: " + why);
    return rc;
  }
  *out = t;
  return RT_OK;
}

// Resize of the whole frame: every band gets its new rows of the new frame; scenes, camera, options stay.
// begins (n + 1 ascending row indices, begins[0] = 0, begins[n] = h) gives an explicit partition
// (rt_tracer_rebalance); null = equal bands.
void multi_resize(rt_tracer* t, uint32_t w, uint32_t h, const uint32_t* begins = nullptr) {
  MultiState& m = *t->mg;
  Group& g = m.group;
  const uint32_t n = static_cast<uint32_t>(m.bands.size());
  if (n > h) throw HipFail{fmt("Resize: %u bands do not fit %u rows", n, h)};
  multi_sync_all(t);
  g.release_buffers();
  multi_free_host_images(m);
  t->W = w; t->H = h; t->rows = h;
  g.W = w; g.H = h;
  for (uint32_t k = 0; k < n; ++k) {
    if (begins) { g.bands[k].row0 = begins[k]; g.bands[k].rows = begins[k + 1] - begins[k]; }
    else band_rows(h, n, k, g.bands[k].row0, g.bands[k].rows);
  }
  m.for_bands([&](size_t k) {
    m.bands[k]->use_device();
    m.bands[k]->reshape(w, h, g.bands[k].row0, g.bands[k].rows);
  });
  g.allocate();
  multi_alloc_host_images(t);
}

// Rows per band such that every band costs the same, from what each band cost so far (piecewise-constant cost
// per row inside a band).  Boundaries are multiples of `granule` rows (the trace kernel's tiles are 8 rows high),
// every band keeps at least one granule.  begins / out: n + 1 ascending row indices.
void balance_rows(uint32_t n, const uint32_t* begins, const double* cost, uint32_t granule, uint32_t* out) {
  const uint32_t H = begins[n];
  if (granule == 0u) granule = 1u;
  double total = 0.0;
  for (uint32_t k = 0; k < n; ++k) total += cost[k] > 0.0 ? cost[k] : 0.0;
  out[0] = 0u; out[n] = H;
  if (!(total > 0.0) || static_cast<uint64_t>(n) * granule > H) { for (uint32_t k = 1; k < n; ++k) out[k] = begins[k]; return; }
  uint32_t src = 0;                        // band of the old partition the sweep is in
  double before = 0.0;                     // cost of the old bands in front of src
  for (uint32_t k = 1; k < n; ++k) {
    const double want = total * k / n;
    while (src + 1u < n && before + (cost[src] > 0.0 ? cost[src] : 0.0) < want) { before += cost[src] > 0.0 ? cost[src] : 0.0; ++src; }
    const double c = cost[src] > 0.0 ? cost[src] : 0.0;
    const double rows_src = static_cast<double>(begins[src + 1] - begins[src]);
    double row = begins[src] + (c > 0.0 ? (want - before) / c * rows_src : 0.0);
    uint32_t r = static_cast<uint32_t>(row / granule + 0.5) * granule;
    const uint32_t lo = out[k - 1] + granule;                          // at least one granule per band ...
    const uint32_t hi = H - (n - k) * granule;                         // ... also for the bands behind
    out[k] = r < lo ? lo : r > hi ? hi : r;
  }
}

size_t member_band_index(rt_tracer* t);

// The tiles as they are, gathered once more (no tracing): what one exchange costs on its own -- rt_tracer_gather_time reads
// the device time.  Collective in a multi-process group.
void group_gather_only(rt_tracer* t) {
  if (t->mg) {
    MultiState& m = *t->mg;
    Group& g = m.group;
    const int b = g.last_b < 0 ? 0 : g.last_b;                         // the frame (and send buffers) the last launch wrote: tiles of
    m.for_bands([&](size_t k) {                                        // root-local bands were stored in place, into that frame only
      m.bands[k]->use_device();
      (void)g.tile_target(k, b);                                       // (orders the band's streams behind the buffer's last reader)
      g.tile_written(k);
    });
    g.gather(b);
    return;
  }
  Group& g = *t->grp;
  const int b = g.last_b < 0 ? 0 : g.last_b;
  const size_t k = member_band_index(t);
  t->use_device();
  (void)g.tile_target(k, b);
  g.tile_written(k);
  g.gather(b);
}

// what the group is made of, as JSON text (reporting: bench.py, tests)
std::string group_info_json(rt_tracer* t) {
  Group* g = t->mg ? &t->mg->group : t->grp;
  if (!g) return "{\"transport\": \"none\", \"ranks\": 1, \"note\": \"a single tracer: nothing to gather\"}";
  rtc::Rccl& nccl = rtc::Rccl::get();
  bool any_travel = false;
  for (const GroupBand& b : g->bands) any_travel = any_travel || g->travels(b);
  const char* transport = g->peer ? "peer" : (any_travel ? "rccl" : "local");
  std::string s = fmt("{\"transport\": \"%s\", \"ranks\": %d, \"bands\": %zu", transport, g->n_ranks, g->bands.size());
  s += ", \"note\": \"";
  s += g->peer ? "every band stores its BGRA8 tile into the root's frame through a peer mapping; no collective"
               : (any_travel ? "grouped ncclSend / ncclRecv of the BGRA8 tiles to rank 0" : "every band lives on the root device: tiles are written in place");
  if (!g->peer_note.empty()) { s += "; "; s += g->peer_note; }
  s += "\"";
  s += ", \"band_ranks\": [";
  for (size_t k = 0; k < g->bands.size(); ++k) s += fmt("%s%d", k ? ", " : "", g->bands[k].rank);
  s += "], \"local_devices\": [";
  for (size_t k = 0; k < g->local.size(); ++k) s += fmt("%s%d", k ? ", " : "", g->local[k].device);
  s += "]";
  bool have_comm = false;
  for (const GatherRank& r : g->local) have_comm = have_comm || r.comm != nullptr;
  if (have_comm && nccl.ok()) {
    int ver = 0;
    if (nccl.GetVersion && nccl.GetVersion(&ver) == ncclSuccess) s += fmt(", \"rccl_version\": %d", ver);
    s += fmt(", \"rccl_library\": \"%s\", \"communicators\": [", nccl.path.c_str());
    bool first = true;
    for (const GatherRank& r : g->local) {
      if (!r.comm) continue;
      int count = -1, dev = -1, rank = -1;
      if (nccl.CommCount) (void)nccl.CommCount(r.comm, &count);
      if (nccl.CommCuDevice) (void)nccl.CommCuDevice(r.comm, &dev);
      if (nccl.CommUserRank) (void)nccl.CommUserRank(r.comm, &rank);
      s += fmt("%s{\"rank\": %d, \"ranks_in_communicator\": %d, \"device\": %d}", first ? "" : ", ", rank, count, dev);
      first = false;
    }
    s += "]";
  }
  s += "}";
  return s;
}

// whole-frame view of the per-band buffers (parity tests, host read-back)
void multi_read_buffer(rt_tracer* t, int which, void* dst, size_t bytes) {
  MultiState& m = *t->mg;
  Group& g = m.group;
  multi_sync_all(t);
  if (which == RT_BUF_IMAGE || which == RT_BUF_FRAME) {
    HIP_CHECK(hipSetDevice(g.local[0].device));
    HIP_CHECK(hipMemcpy(dst, g.d_frame[g.last_b < 0 ? 0 : g.last_b], bytes, hipMemcpyDeviceToHost));
    return;
  }
  const size_t per_px = which == RT_BUF_RENDER ? sizeof(float4) : sizeof(uint32_t);
  const size_t planes = which == RT_BUF_RNG ? 6u : 1u;
  const size_t frame_px = static_cast<size_t>(t->W) * t->H;
  if (bytes < frame_px * per_px * planes) throw HipFail{"rt_tracer_read_buffer: a multi-device tracer reads whole buffers only"};
  for (size_t k = 0; k < m.bands.size(); ++k) {
    rt_tracer* band = m.bands[k];
    band->use_device();
    const size_t band_px = band->npix(), off_px = static_cast<size_t>(g.bands[k].row0) * t->W;
    for (size_t p = 0; p < planes; ++p)
      HIP_CHECK(hipMemcpy(static_cast<char*>(dst) + (p * frame_px + off_px) * per_px,
                          static_cast<const char*>(buffer_ptr(band, which)) + p * band_px * per_px, band_px * per_px,
                          hipMemcpyDeviceToHost));
  }
}

// ---- a band tracer as a member of a multi-process group ----------------------------------------------
void member_join(rt_tracer* t, uint32_t n_ranks, uint32_t rank, const uint8_t id[RT_GROUP_ID_BYTES], const uint32_t* begins) {
  if (t->grp) throw HipFail{"rt_tracer_join_group: already a member of a group"};
  uint32_t r0 = 0, rn = 0;
  if (begins) {
    if (begins[0] != 0u || begins[n_ranks] != t->H) throw HipFail{"rt_tracer_join_group_bands: the bands must cover rows 0 .. full_height"};
    for (uint32_t k = 0; k < n_ranks; ++k) if (begins[k + 1] <= begins[k]) throw HipFail{"rt_tracer_join_group_bands: empty or descending band"};
    r0 = begins[rank]; rn = begins[rank + 1] - begins[rank];
  } else {
    band_rows(t->H, n_ranks, rank, r0, rn);
  }
  if (!t->band_mode && n_ranks > 1u) throw HipFail{"rt_tracer_join_group: the tracer must own a row band (rt_options.full_height)"};
  if (r0 != t->row0 || rn != t->rows)
    throw HipFail{fmt("rt_tracer_join_group: rank %u of %u owns rows [%u, %u) of %u, the tracer has [%u, %u)", rank, n_ranks, r0,
                      r0 + rn, t->H, t->row0, t->row0 + t->rows)};
  Group* g = new Group();
  g->W = t->W; g->H = t->H;
  g->n_ranks = static_cast<int>(n_ranks);
  g->has_root = rank == 0u;
  g->self_rccl = env_on("RT_MI355X_GATHER_SELF");
  for (uint32_t k = 0; k < n_ranks; ++k) {
    GroupBand gb;
    if (begins) { gb.row0 = begins[k]; gb.rows = begins[k + 1] - begins[k]; }
    else band_rows(t->H, n_ranks, k, gb.row0, gb.rows);
    gb.rank = static_cast<int>(k);
    gb.tracer = k == rank ? t : nullptr;
    g->bands.push_back(gb);
  }
  GatherRank me;
  me.rank = static_cast<int>(rank); me.device = t->device;
  g->local.push_back(me);
  try {
    t->use_device();
    if (n_ranks > 1u || g->self_rccl) {
      rtc::Rccl& nccl = need_rccl();
      ncclUniqueId uid;
      static_assert(sizeof(uid.internal) == RT_GROUP_ID_BYTES, "rt_mi355x.h: RT_GROUP_ID_BYTES");
      memcpy(uid.internal, id, sizeof uid.internal);
      RCCL_CHECK(nccl.CommInitRank(&g->local[0].comm, static_cast<int>(n_ranks), uid, static_cast<int>(rank)));   // collective over the ranks
    }
    g->allocate();
  } catch (...) {
    g->destroy();
    delete g;
    throw;
  }
  t->grp = g;
}

void member_leave(rt_tracer* t) {
  if (!t->grp) return;
  (void)hipSetDevice(t->device);
  if (t->stream_b) (void)hipStreamSynchronize(t->stream_b);
  if (t->stream) (void)hipStreamSynchronize(t->stream);
  for (GatherRank& r : t->grp->local) if (r.gstream) (void)hipStreamSynchronize(r.gstream);
  t->grp->destroy();
  delete t->grp;
  t->grp = nullptr;
}

size_t member_band_index(rt_tracer* t) {
  for (size_t k = 0; k < t->grp->bands.size(); ++k) if (t->grp->bands[k].tracer == t) return k;
  return 0;
}

}  // namespace
