#!/usr/bin/env python3
"""bench.py -- Mray/s of the trace path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W            (any N: ONE process; for N > 1 the library shards the
                                                            frame over the devices itself, rt_tracer_create_multi)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W     (one process per GPU, rt_tracer_join_group)

A "step" is one complete Trace pass of the hot path on device-resident buffers: clear the
accumulators, ONE trace-kernel launch of `samples` spp over every band (fused clear and BGRA8
conversion) and -- for N > 1 -- the RCCL gather of the finished tiles to the root device
(RayTracerImpl.cu:236-315 without the GUI hand-off).  Workload at N=1: BASELINE.json configs[2] = C3
(Cornell-box 32 triangles, thin-lens DoF, 1920x1080, 16 spp), the configuration the metric is quoted
on.  For N > 1 the headline record keeps per-GPU work fixed ("weak": the frame grows to 1920 x (1080*N),
one 1080-row band per GPU) and the same line carries a second record, `c5_strong`: BASELINE.json
configs[4], the 10k-triangle scene at 3840x2160x256 spp as ONE frame in N row bands.  If the box has
fewer devices than N (a 1-GPU box), the single-process form places the bands round-robin on what is
there and says so (`config.devices`).

Before the W warmup steps every device is brought to its steady clocks with the VALU calibration loop of the `valu`
record (`clock_preheat`, 150 ms, no step of the path in it: a cold MI355X needs ~25 ms of load to ramp, which is longer than
a short run's whole timed region; tools/clock_ramp.py, --no-preheat).

Prints ONE JSON line on rank 0 with
  roofline      HBM: algorithmic bytes per launch / live HIP-event kernel time vs 8 TB/s (contractual
                bound), frac_wall = the same bytes / wall time per step, and the R = 48 accounting beside it
  valu          the binding bound: VALU instructions issued per launch (PMC pass, stamped with the kernel
                source hash: null + stale when the loaded library is a different build) against the wave64
                issue rate this device sustains (calibrated live); `algorithmic` = the reference's
                full-scan flops for context (a ratio, not a utilisation)
  gather_ms     N > 1: device time of one tile gather on the root's gather stream
  c5_strong     N > 1: the strong-scaling record; measured after the headline under a watchdog (RT_MI355X_C5_TIMEOUT, 300 s):
                if it or the final barrier blocks, rank 0 still prints the line, with the reason in c5_strong.error
                (a sharded run that makes no progress before the headline is measured ends after RT_MI355X_BENCH_TIMEOUT,
                900 s, with value null and the phase it was in)
  warm_lists    NOT the headline: the same steps with the library's default list reuse across Traces
  full_path_all_tiles  NOT the headline: the same steps with certain-winner tiles switched off (every tile traces its rays)
  cpu_baseline  the oracle (scalar CPU port of the reference kernel) timed on this box's cores, N = 1 only
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3       # fp32 vector peak (spec)
RNG_STATE_BYTES = 24           # persisted per pixel: d + v[5] (the reference's curandState_t is 48)
PREHEAT_MS = 150.0             # VALU calibration loop before the warmup steps (see preheat())
FLOP_BY_EXIT = (20, 30, 46, 52)   # culled at det / rejected at u / at v / full
FLOP_PER_RAY_SETUP = 100          # ray generation + shading, per ray

WORKLOADS = {"C2": "C2: 1 sphere, pinhole, 512x512, 1 spp",
             "C3": "C3: Cornell-box 32 triangles, thin-lens DoF, 1920x1080, 16 spp",
             "C4": "C4: 10k random triangles, 3840x2160, 64 spp",
             "C5": "C5: 10k random triangles, 3840x2160, 256 spp, one frame in row bands"}


def algorithmic_bytes(width, rows, n_tris, n_spheres, rng_bytes=RNG_STATE_BYTES):
    """per launch, W*H*(2*R + 32) + 48*N_tri + 16*N_sph.
    32 = render-buffer RMW 12+12 + sample-count RMW 4+4."""
    return width * rows * (2 * rng_bytes + 32) + 48 * n_tris + 16 * n_spheres


def host_threads():
    """Threads for the CPU baseline: the box's CPU share (cgroup quota if any, else 16)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("RT_BENCH_CPU_THREADS", "16"))))


def cpu_baseline(cfg, tris, spheres, rows, threads):
    """The oracle (CPU port of the reference kernel) on a bounded sample: `rows` centred
    rows of the same frame, one launch of cfg['samples'] spp, `threads` host threads.
    RNG-state creation is outside the timed region, as on the GPU."""
    from oracle import oracle_py as orc
    H = cfg["height"]
    row0 = (H - rows) // 2
    o = orc.OracleTracer(cfg["width"], H, cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"],
                         seed=cfg["seed"], row0=row0, rows=rows, contract=orc.FMA, nthreads=threads)
    if tris.shape[0]:
        o.upload_scene(tris)
    if spheres.shape[0]:
        o.upload_spheres(spheres)
    t0 = time.perf_counter()
    o.launch(cfg["samples"])                     # sizing pass (also warms the caches), not reported
    probe = time.perf_counter() - t0
    launches = max(1, min(64, int(round(20.0 / max(probe * threads, 1e-3)))))   # ~20 core-seconds
    t0 = time.perf_counter()
    for _ in range(launches):
        o.launch(cfg["samples"])
    dt = time.perf_counter() - t0
    rays = cfg["width"] * rows * cfg["samples"] * launches
    # the same oracle on ONE thread, on a 64-row slice (~2 core-seconds), for a per-core figure
    one_rows = min(rows, 64)
    o1 = orc.OracleTracer(cfg["width"], H, cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=cfg["seed"],
                          row0=(H - one_rows) // 2, rows=one_rows, contract=orc.FMA, nthreads=1)
    if tris.shape[0]:
        o1.upload_scene(tris)
    if spheres.shape[0]:
        o1.upload_spheres(spheres)
    t1 = time.perf_counter()
    o1.launch(cfg["samples"])
    one = cfg["width"] * one_rows * cfg["samples"] / (time.perf_counter() - t1) / 1e6
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mray/s", "cores": threads, "kind": "port",
            "value_1core": round(one, 3),
            "sample": "%d centred rows of the %dx%d frame (rows %d..%d), %d launches x %d spp = %d rays, %.1f s wall "
                      "(%.0f core-seconds); oracle/oracle.c, gcc -O2 -ffp-contract=off, scalar, row-threaded"
                      % (rows, cfg["width"], H, row0, row0 + rows - 1, launches, cfg["samples"], rays, dt, dt * threads)}


def preheat(devices, min_ms=PREHEAT_MS):
    """Bring the devices to their steady clocks before the warmup steps: the VALU calibration loop of the `valu` record
    (rt_dbg_valu_peak, ~3 ms per run) repeated for min_ms on every device at once.  tools/clock_ramp.py: from a cold process
    a C3 step takes 118 us and reaches its steady 98.5 us only after ~250 steps (25 ms of load) -- longer than the whole
    timed region of a short run -- and after 70 ms of this loop the first steps are at the steady value.  No step of the
    path runs here.  Returns {device: (lane_fma_per_s, clock_ghz, ms)} of each device's last run."""
    import threading
    from raytracertest_amd import api
    out = {}

    def run(dev):
        t0 = time.perf_counter()
        while True:
            lane_fma, ghz = api.dbg_valu_peak(dev)
            ms = (time.perf_counter() - t0) * 1e3
            if ms >= min_ms:
                break
        out[dev] = (lane_fma, ghz, ms)

    threads = [threading.Thread(target=run, args=(d,)) for d in sorted(set(devices))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return out


def timed_steps(job, steps, warmup):
    """W untimed steps, then exactly K steps bracketed by (device drain + barrier) on both sides; MAX over ranks."""
    for _ in range(warmup):
        job.step()
    job.finish()
    job.tracer.KernelTime(reset=True)
    job.tracer.GatherTime(reset=True)
    job.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        job.step()
    job.finish()
    job.barrier()
    elapsed = job.max_over_ranks(time.perf_counter() - t0)
    kernel_ms, launches = job.tracer.KernelTime(reset=True)
    gather_ms, gathers = job.tracer.GatherTime(reset=True)
    return {"elapsed": elapsed, "kernel_ms": kernel_ms, "launches": launches, "gather_ms": gather_ms, "gathers": gathers}


def stamped(path, config, kernel_hash):
    """(entry, stale): the profiles/*.json entry for `config` if it was collected on THIS kernel build."""
    try:
        doc = json.load(open(path))
    except (OSError, ValueError):
        return None, False
    entry = doc.get(config)
    if not entry:
        return None, False
    if doc.get("kernel_source_hash") != kernel_hash:
        return None, True
    return entry, False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C3", choices=["C2", "C3", "C4", "C5"],
                    help="C3 (default, the metric's configuration; weak scaling for --gpus N, with a C5 strong-scaling record "
                         "beside it); C5 = the C4 scene at 256 spp, ONE 3840x2160 frame in N row bands (BASELINE configs[4])")
    ap.add_argument("--cpu-rows", type=int, default=-1, help="rows of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-valu", action="store_true", help="skip the instrumented launch + VALU calibration")
    ap.add_argument("--no-warm", action="store_true", help="skip the extra warm_lists measurement (profiling runs: headline launches only)")
    ap.add_argument("--no-c5", action="store_true", help="N > 1: skip the c5_strong record")
    ap.add_argument("--no-preheat", action="store_true", help="start the warmup steps on a cold device (clocks not ramped)")
    ap.add_argument("--samples-in-flight", type=int, default=0)
    ap.add_argument("--lds-chunk", type=int, default=0)
    args = ap.parse_args()

    import raytracertest_amd as R
    from raytracertest_amd import api, scenes
    from raytracertest_amd.dist import RowBandJob

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    devices = None
    if world > 1:
        args.gpus = world                                  # one process per GPU: the launcher decides
        if local_rank >= R.device_count():
            sys.exit("bench.py: rank %d has no HIP device %d (one process per GPU needs %d devices; on a smaller box run "
                     "`python bench.py --gpus %d` without a launcher)" % (rank, local_rank, world, world))
    elif args.gpus > 1:                                    # one process, the library shards the frame over the devices
        n_dev = R.device_count()
        if n_dev < 1:
            sys.exit("bench.py: no HIP device")
        devices = [k % n_dev for k in range(args.gpus)]
    n_parts = args.gpus
    sharding = "single GPU" if n_parts == 1 else (
        "one process per GPU (rt_tracer_join_group), RCCL gather of BGRA8 tiles to rank 0" if world > 1 else
        "one process, %d row bands over %d device(s) (rt_tracer_create_multi), RCCL gather of BGRA8 tiles to the root device"
        % (n_parts, len(set(devices))))

    cfg = dict(scenes.CONFIGS[args.config])
    tris, spheres = scenes.scene_for(args.config)
    n_tris = tris.shape[0] // 3
    weak = args.config != "C5"

    # A sharded run that blocks for good (a collective whose peer never arrives) ends with a reason instead of the
    # launcher's kill: after RT_MI355X_BENCH_TIMEOUT seconds rank 0 says in which phase, every rank leaves.
    phase = {"name": "setting up the job"}
    if n_parts > 1:
        def stuck():
            if phase["name"] is None:
                return
            sys.stderr.write("bench.py rank %d: no progress while %s\n" % (rank, phase["name"]))
            if rank == 0:
                print(json.dumps({"metric": "Mray/s at %dx%dx%dspp" % (cfg["width"], cfg["height"], cfg["samples"]), "value": None,
                                  "unit": "Mray/s", "n_gpus": n_parts, "error": "timed out while %s" % phase["name"]}), flush=True)
            sys.stderr.flush()
            os._exit(3)

        whole = threading.Timer(float(os.environ.get("RT_MI355X_BENCH_TIMEOUT", "900")), stuck)
        whole.daemon = True
        whole.start()
    try:
        job = RowBandJob(cfg, tris, spheres, world=world, rank=rank, local_rank=local_rank, weak=weak, devices=devices,
                         samples_in_flight=args.samples_in_flight, lds_chunk=args.lds_chunk)
    except Exception as e:
        # a sharded job that cannot be set up (device missing, RCCL communicator not formed) must end the run with a
        # reason, not leave ranks blocked in a collective: say why on every rank, one JSON line on rank 0, hard exit
        sys.stderr.write("bench.py rank %d: %s: %s\n" % (rank, type(e).__name__, e))
        if rank == 0:
            print(json.dumps({"metric": "Mray/s at %dx%dx%dspp" % (cfg["width"], cfg["height"], cfg["samples"]), "value": None,
                              "unit": "Mray/s", "n_gpus": n_parts, "error": "%s: %s" % (type(e).__name__, e)}), flush=True)
        sys.stderr.flush()
        os._exit(2)
    # Headline: every step is a from-scratch Trace pass -- the tile candidate lists (a camera-dependent
    # acceleration structure the library keeps between Traces by default) are NOT carried from step to step.
    job.tracer.SetListReuse(False)
    phase["name"] = "measuring the headline steps"
    heat = {} if args.no_preheat else preheat(devices if devices else [local_rank])
    res = timed_steps(job, args.steps, args.warmup)
    phase["name"] = None                                   # the headline is in hand: from here on emit() is the way out
    elapsed, kernel_ms, launches = res["elapsed"], res["kernel_ms"], res["launches"]
    bands = job.tracer.Bands()
    band0_rows = bands[0]["rows"]

    warm = None
    if n_parts == 1 and args.config in ("C2", "C3") and not args.no_warm:
        # the library's default behaviour for repeated Traces of an unchanged view: lists built once, then reused
        job.tracer.SetListReuse(True)
        for _ in range(max(args.warmup, 2)):
            job.step()
        job.finish()
        job.tracer.KernelTime(reset=True)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            job.step()
        job.finish()
        dt = time.perf_counter() - t1
        wk_ms, wk_n = job.tracer.KernelTime(reset=True)
        warm = {"value": round(cfg["width"] * job.rows * cfg["samples"] * cfg["iterations"] * args.steps / dt / 1e6, 2), "unit": "Mray/s",
                "ms_per_step": round(dt / args.steps * 1e3, 5), "kernel_us": round(wk_ms / max(wk_n, 1) * 1e3, 2),
                "note": "NOT the headline: same steps with the tile candidate lists kept between Traces (library default, "
                        "rt_tracer_set_list_reuse): classification and ray-family work happen once, results identical"}
        job.tracer.SetListReuse(False)
    launch_info = job.tracer.Info()
    version = R.load_library().rt_version().decode()
    kernel_hash = version.split("kernels=")[-1].rstrip(")") if "kernels=" in version else None
    full_path = None
    c5 = None
    emitted = threading.Lock()
    state = {"done": False}

    def emit(c5):
        """The one JSON line (rank 0).  A function so that the watchdog of the extra N > 1 record can still print the headline."""
        with emitted:
            if state["done"]:
                return
            state["done"] = True
            rays_band0 = cfg["width"] * band0_rows * cfg["samples"] * cfg["iterations"]
            total_rays = cfg["width"] * cfg["height"] * (n_parts if weak else 1) * cfg["samples"] * cfg["iterations"]
            value = total_rays * args.steps / elapsed / 1e6

            if rank == 0:
                avg_kernel_s = kernel_ms / max(launches, 1) / 1e3
                b_alg = algorithmic_bytes(cfg["width"], band0_rows, n_tris, spheres.shape[0])        # one GPU's band
                b_alg48 = algorithmic_bytes(cfg["width"], band0_rows, n_tris, spheres.shape[0], 48)
                achieved = b_alg / avg_kernel_s / 1e9
                step_s = elapsed / args.steps
                traffic, traffic_stale = stamped(os.path.join(ROOT, "profiles", "hbm_traffic.json"), args.config, kernel_hash)
                split = band0_rows >= 128 and os.environ.get("RT_MI355X_NO_SPLIT") != "1"
                out = {
                    "metric": "Mray/s at %dx%dx%dspp" % (cfg["width"], cfg["height"], cfg["samples"]),
                    "value": round(value, 2), "unit": "Mray/s", "n_gpus": n_parts, "steps": args.steps,
                    "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 5),
                    "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f32",
                    "data": "synthetic",
                    "clock_preheat": ({"ms": round(max(h[2] for h in heat.values()), 1), "clock_ghz": round(min(h[1] for h in heat.values()), 3),
                                       "note": "before the W warmup steps every device ran the VALU calibration loop of the valu record for this long, so "
                                               "that the timed steps run at the device's steady clocks (tools/clock_ramp.py: a cold device needs ~250 "
                                               "steps = 25 ms of load to get there); no step of the path runs in it; --no-preheat turns it off"}
                                      if heat else None),
                    "config": {"workload": WORKLOADS[args.config],
                               "image": "%dx%d per GPU (row band of a %dx%d frame)" % (
                                   cfg["width"], band0_rows, cfg["width"], cfg["height"] * (n_parts if weak else 1)),
                               "triangles": n_tris, "spheres": int(spheres.shape[0]), "samples_per_launch": cfg["samples"],
                               "launch": launch_info, "math_mode": "fma", "rng_seed": cfg["seed"],
                               "sharding": sharding, "library": version},
                    "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(achieved / HBM_PEAK_GBS, 5),
                                 "frac_wall": round(b_alg / step_s / 1e9 / HBM_PEAK_GBS, 5),
                                 "traffic": traffic.get("bytes_per_launch") if traffic else None,
                                 "kernel": "trace_kernel", "kernel_us": round(avg_kernel_s * 1e6, 2),
                                 "kernels_per_launch": 2 if split else 1,
                                 "kernel_Mray_s": round(rays_band0 / avg_kernel_s / 1e6, 2),
                                 "algorithmic_bytes_per_launch": b_alg, "rng_state_bytes": RNG_STATE_BYTES,
                                 "reference_layout_R48": {"algorithmic_bytes_per_launch": b_alg48,
                                                          "frac": round(b_alg48 / avg_kernel_s / 1e9 / HBM_PEAK_GBS, 5),
                                                          "note": "what the same kernel time would read with the reference's 48-byte curandState_t "
                                                                  "accounting (SURVEY 8d); those bytes are NOT moved: the build persists 24 B/pixel"},
                                 "note": "contractual bound; the path is fp32-VALU-issue-bound by construction (SURVEY.md 0.5, BASELINE.md 2): see "
                                         "valu.  A launch runs as two half-frame kernels on two streams that execute concurrently "
                                         "(profiles/r02_c3_overlap.csv): kernel_us is the sampled duration of one of them (what rocprofv3 lists per "
                                         "dispatch), frac = the launch's algorithmic bytes / kernel_us / peak, frac_wall = the same bytes / "
                                         "ms_per_step / peak (clear, both kernels, conversion, launch gaps: everything a step costs)"},
                }
                if traffic_stale:
                    out["roofline"]["traffic_stale"] = True      # profiles/hbm_traffic.json belongs to another kernel build
                if n_parts > 1:
                    out["config"]["devices"] = [b["device"] for b in bands] if bands and len(bands) > 1 else list(range(n_parts))
                    out["gather_ms"] = round(res["gather_ms"] / res["gathers"], 4) if res["gathers"] else 0.0
                    out["gather_note"] = ("device time of one gather on the root's gather stream (HIP events around the grouped ncclSend/ncclRecv; "
                                          "includes waiting for the slowest peer's tile); it overlaps the next step's tracing.  0 with every band "
                                          "on the root device: those tiles are written in place by the trace kernel")
                    if c5 is not None:
                        out["c5_strong"] = c5
                if warm is not None:
                    out["warm_lists"] = warm
                if full_path is not None:
                    out["full_path_all_tiles"] = full_path
                if not args.no_valu and n_parts == 1 and args.config != "C5":
                    # instrumented launch of the reference's own algorithm (every ray scans the whole list,
                    # reference-order tests) on a scratch tracer: exit points per test
                    g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"],
                                    cfg["aperture"], seed=cfg["seed"], device=local_rank, no_filter=True, no_binning=True)
                    if tris.shape[0]:
                        g.UploadScene(tris)
                    if spheres.shape[0]:
                        g.UploadSpheres(spheres)
                    st_samples = cfg["samples"] if args.config != "C4" else 4
                    st = g.TraceStats(st_samples)
                    g.close()
                    exits = [st["exit_det"], st["exit_u"], st["exit_v"], st["exit_hit"]]
                    scale = cfg["samples"] / st_samples
                    flop = scale * sum(f * e for f, e in zip(FLOP_BY_EXIT, exits)) + FLOP_PER_RAY_SETUP * rays_band0
                    lane_fma, ghz = heat[local_rank][:2] if local_rank in heat else api.dbg_valu_peak(local_rank)
                    rate_peak = lane_fma / 64.0                      # wave64 VALU instructions per second, whole device
                    tf = flop / avg_kernel_s / 1e12
                    issued, issued_stale = stamped(os.path.join(ROOT, "profiles", "valu_issue.json"), args.config, kernel_hash)
                    valu = {"bound": "fp32 VALU issue", "unit": "Ginst/s (wave64)", "peak": round(rate_peak / 1e9, 2),
                            "peak_note": "wave64 instruction rate of an 8-chain v_fma_f32 loop at 8 waves/SIMD on this device, measured now",
                            "clock_ghz_under_load": round(ghz, 3), "achieved": None, "frac": None}
                    if issued:
                        n_inst = issued["valu_wave_instructions_per_launch"]
                        valu.update({"achieved": round(n_inst / avg_kernel_s / 1e9, 2), "frac": round(n_inst / avg_kernel_s / rate_peak, 4),
                                     "valu_wave_instructions_per_launch": n_inst,
                                     "lane_instructions_per_ray": round(n_inst * 64.0 / rays_band0, 1),
                                     "note": "SQ_INSTS_VALU per launch (rocprofv3 --pmc, own pass, same kernel build) / kernel time"})
                    elif issued_stale:
                        valu["stale"] = True                         # profiles/valu_issue.json belongs to another kernel build
                    valu["algorithmic"] = {
                        "full_scan_tflops_equivalent": round(tf, 2), "spec_peak_tflops": VALU_PEAK_TFLOPS,
                        "ratio_to_spec_peak": round(tf / VALU_PEAK_TFLOPS, 4),
                        "tests_per_s": round(scale * sum(exits) / avg_kernel_s, 1),
                        "exit_fractions": [round(e / max(sum(exits), 1), 4) for e in exits],
                        "algorithmic_flop_per_launch": int(flop),
                        "note": "NOT a utilisation: the flops of the reference's full scan (every ray x every triangle priced by its exit point, "
                                "20/30/46/52, FMA = 2, + 100 per ray) divided by this kernel's time.  The kernel skips the triangles its per-tile "
                                "classification proves missed, so the ratio may exceed 1"}
                    g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"],
                                    cfg["aperture"], seed=cfg["seed"], device=local_rank)
                    if tris.shape[0]:
                        g.UploadScene(tris)
                    sb = g.TraceStats(st_samples)
                    g.close()
                    waves = ((cfg["width"] + 7) // 8) * ((cfg["height"] + 7) // 8)
                    valu["binning"] = {"candidates_per_tile": round(sb["bin_candidates"] / max(sb["bin_rounds"], 1), 2),
                                       "of_triangles": n_tris, "classification_rounds_per_tile": round(sb["bin_rounds"] / waves, 3)}
                    tl = sb.get("tiles_by_list", {})
                    if sum(tl.values()):
                        valu["binning"]["tiles"] = {k: round(v / sum(tl.values()), 4) for k, v in tl.items()}
                        valu["binning"]["tiles_note"] = ("share of the 8x8 tiles with a CERTAIN winner (one triangle every ray of the tile's family certainly hits "
                                                         "and that is certainly the farthest hit: their samples keep only the RNG draws and the additions), and of "
                                                         "the others by the length of their candidate list")
                    out["valu"] = valu
                if n_parts == 1 and args.cpu_rows != 0:
                    rows = args.cpu_rows if args.cpu_rows > 0 else min(cfg["height"], {"C2": 512, "C3": 1080, "C4": 8, "C5": 2}[args.config])
                    out["cpu_baseline"] = cpu_baseline(cfg, tris, spheres, rows, host_threads())
                print(json.dumps(out), flush=True)

    watchdog = None
    if n_parts > 1:
        # Everything from here on is extra (the headline is measured): a rank that blocks in it -- a collective whose peer
        # died, a communicator that does not form -- must not cost the run its line.  After RT_MI355X_C5_TIMEOUT seconds
        # rank 0 prints the headline with the reason in c5_strong and every rank leaves.
        limit = float(os.environ.get("RT_MI355X_C5_TIMEOUT", "300"))

        def give_up():
            if rank == 0:
                emit({"workload": WORKLOADS["C5"], "error": "the extra record or the final barrier did not finish within %.0f s" % limit})
            sys.stdout.flush()
            os._exit(0)

        watchdog = threading.Timer(limit, give_up)
        watchdog.daemon = True
        watchdog.start()
    job.close(destroy_group=False)

    if n_parts == 1 and args.config in ("C3",) and not args.no_warm:
        # the same cold steps with every tile on the full path (RT_FLAG_NO_SURE_HIT: tiles with a certain winner generate their
        # rays and run their tests anyway) -- what the kernel does per ray when nothing can be proven, next to the headline
        fp = R.RayTracer((cfg["width"], cfg["height"]), (0.0, 0.0, 0.0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"],
                         seed=cfg["seed"], device=local_rank, no_sure_hit=True)
        if tris.shape[0]:
            fp.UploadScene(tris)
        fp.SetListReuse(False)
        for _ in range(max(args.warmup, 2)):
            fp.TraceEnqueue(cfg["iterations"], cfg["samples"])
        fp.Sync()
        fp.KernelTime(reset=True)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            fp.TraceEnqueue(cfg["iterations"], cfg["samples"])
        fp.Sync()
        dt = time.perf_counter() - t1
        fk_ms, fk_n = fp.KernelTime(reset=True)
        fp.close()
        full_path = {"value": round(cfg["width"] * cfg["height"] * cfg["samples"] * cfg["iterations"] * args.steps / dt / 1e6, 2), "unit": "Mray/s",
                     "ms_per_step": round(dt / args.steps * 1e3, 5), "kernel_us": round(fk_ms / max(fk_n, 1) * 1e3, 2),
                     "note": "NOT the headline: the same steps with RT_FLAG_NO_SURE_HIT -- every tile generates its rays and runs its tests, "
                             "also the tiles whose winner is certain for the whole ray family (identical image)"}
    # N > 1: the strong-scaling record of BASELINE configs[4] beside the weak headline
    if n_parts > 1 and args.config == "C3" and not args.no_c5:
        try:
            cfg5 = dict(scenes.CONFIGS["C5"])
            tris5, sph5 = scenes.scene_for("C5")
            job5 = RowBandJob(cfg5, tris5, sph5, world=world, rank=rank, local_rank=local_rank, weak=False, devices=devices)
            job5.tracer.SetListReuse(False)
            steps5, warm5 = max(2, min(args.steps, 20)), max(1, min(args.warmup, 2))
            # the bands of this scene are not equally expensive (tools/band_balance.py: mean/max = 0.83 with equal rows in 8
            # bands): two rounds of "measure, re-partition the rows" before the timed steps (rt_tracer_rebalance / rt_balance_rows)
            rows5 = None
            for _ in range(2):
                job5.tracer.KernelTime(reset=True)
                for _ in range(2):
                    job5.step()
                job5.finish()
                rows5 = job5.rebalance()
            r5 = timed_steps(job5, steps5, warm5)
            rays5 = cfg5["width"] * cfg5["height"] * cfg5["samples"] * cfg5["iterations"]
            c5 = {"workload": WORKLOADS["C5"], "scaling": "strong", "value": round(rays5 * steps5 / r5["elapsed"] / 1e6, 2), "unit": "Mray/s",
                  "steps": steps5, "warmup": warm5, "ms_per_step": round(r5["elapsed"] / steps5 * 1e3, 4),
                  "kernel_ms_band0": round(r5["kernel_ms"] / max(r5["launches"], 1), 4),
                  "gather_ms": round(r5["gather_ms"] / r5["gathers"], 4) if r5["gathers"] else 0.0,
                  "image": "%dx%d frame in %d row bands" % (cfg5["width"], cfg5["height"], n_parts),
                  "rows_per_band": rows5, "partition": "balanced on the bands' measured kernel times (rt_balance_rows, multiples of 8 rows)"}
            job5.close(destroy_group=False)
        except Exception as e:             # the headline must survive a failure of the extra record (same on every rank)
            c5 = {"workload": WORKLOADS["C5"], "error": "%s: %s" % (type(e).__name__, e)}
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()

    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        emit(c5)


if __name__ == "__main__":
    main()
