#!/usr/bin/env python3
"""Summarises the profiler passes of tools/profile_round.sh into profiles/: kernel stats CSVs, per-kernel means of the PMC
passes, the overlap of the half-frame kernels, and the derived profiles/hbm_traffic.json + profiles/valu_issue.json that
bench.py reads (stamped with the kernel source hash of the library the passes ran on).
  python tools/pmc_summary.py <tag> <raw profiler output dir>

A LAUNCH of the hot path is several dispatches: the trace kernel as two half-frame kernels on two streams, the small scenes'
list builder (region_lists_kernel / tile_lists_kernel, on its own stream) or the dense scenes' macro_bin_kernel ahead of each
half.  Per-launch figures = the sum of a counter over all dispatches of those kernels / the number of launches, the number
of launches = trace-kernel waves in total / waves of one frame."""
import collections, csv, glob, json, os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, src = sys.argv[1], sys.argv[2]
out = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
PATH_KERNELS = ("trace_kernel", "lists_kernel", "macro_bin_kernel", "macro_bounds_kernel", "super_bin_kernel")


def newest(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return f[-1] if f else None


def short(name):
    return name.split("(")[0].replace("void ", "")


for sub, name in (("stats", "c3"), ("stats_c4", "c4")):
    f = newest(os.path.join(sub, "**", "*kernel_stats.csv"))
    if f:
        shutil.copy(f, os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)))
shutil.copy(os.path.join(out, "bench_c3_under_rocprof.json"), os.path.join(dst, "%s_bench_c3_under_rocprof.json" % tag))
lib = json.loads(open(os.path.join(out, "bench_c3.json")).read().strip().splitlines()[-1])["config"].get("library", "")
khash = lib.split("kernels=")[-1].rstrip(")") if "kernels=" in lib else None


def read_pass(sub):
    """{(kernel, counter): [values per dispatch]} of one --pmc pass"""
    f = newest(os.path.join(sub, "**", "*counter_collection.csv"))
    agg = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg


def per_launch(agg, counter, frame_waves, waves_agg=None):
    """(sum over the path's kernels of `counter` / launches, {kernel: share per launch}, launches)"""
    wa = waves_agg if waves_agg is not None else agg
    tw = sum(sum(v) for (k, c), v in wa.items() if c == "SQ_WAVES" and "trace_kernel" in k)
    launches = tw / frame_waves if tw else 0.0
    parts = {}
    for (k, c), v in agg.items():
        if c == counter and any(p in k for p in PATH_KERNELS):
            parts[k] = parts.get(k, 0.0) + sum(v)
    if not launches:                               # a pass without SQ_WAVES: dispatches of the trace kernel / kernels per launch
        return None, parts, 0.0
    return sum(parts.values()) / launches, {k: v / launches for k, v in parts.items()}, launches


rows = []
passes = {p: read_pass(p) for p in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_fetch_c4", "pmc_write_c4", "pmc_sq_c4")}
for p in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for (k, c), v in sorted(passes[p].items()):
        rows.append([p, k, c, len(v), "%.1f" % (sum(v) / len(v)), "%.1f" % min(v), "%.1f" % max(v)])
with open(os.path.join(dst, "%s_c3_pmc_summary.csv" % tag), "w", newline="") as f:
    w = csv.writer(f); w.writerow(["pass", "kernel", "counter", "dispatches", "mean", "min", "max"]); w.writerows(rows)


def launches_by_dispatch(agg, kpl):
    n = max((len(v) for (k, c), v in agg.items() if "trace_kernel" in k), default=0)
    return n / kpl if kpl else 0.0


def config(prefix, W, H, spp, n_tris):
    frame_waves = ((W + 7) // 8) * ((H + 7) // 8)
    sq = passes["pmc_sq" + prefix]
    if not sq:
        return None, None
    valu, valu_parts, launches = per_launch(sq, "SQ_INSTS_VALU", frame_waves)
    tk = [k for (k, c) in sq if "trace_kernel" in k and c == "SQ_WAVES"][0]
    kpl = max(1, round(frame_waves / (sum(sq[(tk, "SQ_WAVES")]) / len(sq[(tk, "SQ_WAVES")]))))
    issue = {"kernels_per_launch": {short(k): round(len(sq[(k, "SQ_WAVES")]) / launches, 2) for (k, c) in sq if c == "SQ_WAVES" and any(p in k for p in PATH_KERNELS)},
             "valu_wave_instructions_per_launch": int(valu),
             "by_kernel": {k: int(v) for k, v in valu_parts.items()},
             "waves": frame_waves, "valu_per_wave": round(valu / frame_waves, 1),
             "lane_instructions_per_ray": round(valu * 64 / (W * H * spp), 1)}
    for c, key in (("SQ_INSTS_SALU", "salu"), ("SQ_INSTS_LDS", "lds")):
        v, _, _ = per_launch(sq, c, frame_waves)
        if v is not None:
            issue[key] = int(v)
    fa, wa = passes["pmc_fetch" + prefix], passes["pmc_write" + prefix]
    traffic = None
    if fa and wa:
        def total(agg, counter):
            n = launches_by_dispatch(agg, kpl)
            return sum(sum(v) for (k, c), v in agg.items() if c == counter and any(p in k for p in PATH_KERNELS)) / n
        f_kb, w_kb = total(fa, "FETCH_SIZE"), total(wa, "WRITE_SIZE")
        traffic = {"bytes_per_launch": int(round((f_kb * 2.0 + w_kb) * 1024)), "fetch_size_kb_raw": round(f_kb, 1), "fetch_correction": 2.0,
                   "write_size_kb": round(w_kb, 1), "algorithmic_bytes_per_launch": W * H * (2 * 24 + 32) + 48 * n_tris,
                   "kernels": sorted({short(k) for (k, c) in fa if any(p in k for p in PATH_KERNELS)})}
    return issue, traffic


i3, t3 = config("", 1920, 1080, 16, 32)
i4, t4 = config("_c4", 3840, 2160, 64, 10000)
if t3:
    t3["note"] = ("trace kernels + the list builder; bench launches treat the accumulators as zero (no accumulator read): reads = 24 B/pixel RNG state, "
                  "writes = 24 B RNG + 16 B RGBA + 4 B count + 4 B BGRA8, + the tiles' lists written and read once")
if t4:
    t4["note"] = "trace kernels + macro_bin_kernel: the macro / block list traffic and the dense-scene kernel's scratch on top of the per-pixel state"
method = ("rocprofv3 --kernel-trace --pmc <counter> in separate passes around `python3 bench.py --steps 5 --warmup 1 --cpu-rows 0 --no-valu --no-warm --no-parity` "
          "(%s, tools/profile_round.sh); per-launch = sum over every dispatch of the path's kernels / launches. " % tag)
json.dump({"kernel_source_hash": khash,
           "_method": method + "Counter unit KB; FETCH_SIZE x2 on gfx950 as MI355X_MICROARCH.md prescribes (factor calibrated in round 1 on convert_kernel, "
                               "profiles/r01_pmc_c3_fetch.csv). Per-kernel means: profiles/%s_c3_pmc_summary.csv." % tag,
           **({"C3": t3} if t3 else {}), **({"C4": t4} if t4 else {})}, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
json.dump({"kernel_source_hash": khash,
           "_method": method + "Wave-level instruction counts (one count per wave64 instruction).",
           **({"C3": i3} if i3 else {}), **({"C4": i4} if i4 else {})}, open(os.path.join(dst, "valu_issue.json"), "w"), indent=1)
print("kernels=%s" % khash)
for name, i, t in (("C3", i3, t3), ("C4", i4, t4)):
    if i:
        print("%s: VALU wave-instructions/launch %d = %.1f lane-instructions per ray; by kernel %s" % (name, i["valu_wave_instructions_per_launch"], i["lane_instructions_per_ray"], i["by_kernel"]))
    if t:
        print("%s: traffic %d B/launch = %.3f x algorithmic (%d)" % (name, t["bytes_per_launch"], t["bytes_per_launch"] / t["algorithmic_bytes_per_launch"], t["algorithmic_bytes_per_launch"]))
# period, kernel durations and the stagger of the two half-frame kernels over time, from the kernel trace of the --stats pass
subprocess.run([sys.executable, os.path.join(root, "tools", "step_periods.py"), os.path.join(src, "stats")], check=False,
               stdout=open(os.path.join(dst, "%s_c3_step_periods.txt" % tag), "w"))
