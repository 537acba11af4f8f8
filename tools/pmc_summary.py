#!/usr/bin/env python3
"""Summarises gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/: kernel stats CSV, bench lines,
per-kernel means of the PMC passes, and the derived profiles/hbm_traffic.json + profiles/valu_issue.json
that bench.py reads.  python tools/pmc_summary.py r01_f"""
import collections, csv, glob, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
for name in ("bench_c3.json", "bench_c4.json", "bench_c3_under_rocprof.json"):
    shutil.copy(os.path.join(src, name), os.path.join(dst, "%s_%s" % (tag, name)))
stats = sorted(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime, reverse=True)
shutil.copy(stats[0], os.path.join(dst, "%s_c3_kernel_stats.csv" % tag))
stats4 = sorted(glob.glob(os.path.join(src, "stats_c4", "*", "*kernel_stats.csv")), key=os.path.getmtime, reverse=True)
if stats4:
    shutil.copy(stats4[0], os.path.join(dst, "%s_c4_kernel_stats.csv" % tag))
# the kernel build the profiled runs used (rt_version() carries a hash of the kernel sources + flags); bench.py
# prints the PMC-derived figures only when the library it loaded has the same hash
lib = json.loads(open(os.path.join(src, "bench_c3.json")).read().strip().splitlines()[-1])["config"].get("library", "")
khash = lib.split("kernels=")[-1].rstrip(")") if "kernels=" in lib else None
rows = []
means = {}
for p in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = max(glob.glob(os.path.join(src, p, "*", "*counter_collection.csv")), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        rows.append([p, k, c, len(v), "%.1f" % (sum(v) / len(v)), "%.1f" % min(v), "%.1f" % max(v)])
        means[(k, c)] = sum(v) / len(v)
with open(os.path.join(dst, "%s_c3_pmc_summary.csv" % tag), "w", newline="") as f:
    w = csv.writer(f); w.writerow(["pass", "kernel", "counter", "dispatches", "mean", "min", "max"]); w.writerows(rows)
tk = [k for (k, c) in means if "trace_kernel" in k][0]
# a launch of a tall frame runs as two half-frame kernels (two streams): per-LAUNCH figures = per-dispatch means x kernels per launch
kpl = max(1, round(((1920 + 7) // 8) * ((1080 + 7) // 8) / means[(tk, "SQ_WAVES")]))
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"):
    means[(tk, c)] *= kpl
fetch_kb, write_kb = means[(tk, "FETCH_SIZE")], means[(tk, "WRITE_SIZE")]
npix = 1920 * 1080
alg = npix * (2 * 24 + 32) + 48 * 32
traffic = int(round((fetch_kb * 2.0 + write_kb) * 1024))
# C4 (optional passes): the dense-scene kernel + its macro pre-pass
c4_traffic = {}
f4f = sorted(glob.glob(os.path.join(src, "pmc_fetch_c4", "*", "*counter_collection.csv")), key=os.path.getmtime)
f4w = sorted(glob.glob(os.path.join(src, "pmc_write_c4", "*", "*counter_collection.csv")), key=os.path.getmtime)
if f4f and f4w:
    def per_launch(path, counter):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and ("trace_kernel" in r["Kernel_Name"] or "macro_bin_kernel" in r["Kernel_Name"]):
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        return sum(2.0 * sum(v) / len(v) for v in agg.values())          # two half-frame dispatches per launch
    f_kb, w_kb = per_launch(f4f[-1], "FETCH_SIZE"), per_launch(f4w[-1], "WRITE_SIZE")
    alg4 = 3840 * 2160 * (2 * 24 + 32) + 48 * 10000
    c4_traffic = {"C4": {"bytes_per_launch": int(round((f_kb * 2.0 + w_kb) * 1024)), "fetch_size_kb_raw": round(f_kb, 1), "fetch_correction": 2.0,
                         "write_size_kb": round(w_kb, 1), "algorithmic_bytes_per_launch": alg4,
                         "note": "trace kernel + macro_bin_kernel; above the algorithmic bytes by the spill stores of the 128-VGPR dense-scene kernel's "
                                 "classification prologue (scratch 56 B/lane, DESIGN.md 6) and the macro / block list traffic"}}
json.dump({
    "kernel_source_hash": khash,
    "_method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE around `python3 bench.py --steps 5 --warmup 1 "
               "--cpu-rows 0 --no-valu` (%s, tools/profile_round.sh). Counter unit KB; FETCH_SIZE x2 on gfx950 as MI355X_MICROARCH.md prescribes "
               "(factor calibrated in round 1 on convert_kernel, profiles/r01_pmc_c3_fetch.csv). Per-kernel means: profiles/%s_c3_pmc_summary.csv." % (tag, tag),
    "C3": {"bytes_per_launch": traffic, "fetch_size_kb_raw": round(fetch_kb, 1), "fetch_correction": 2.0, "write_size_kb": round(write_kb, 1),
           "algorithmic_bytes_per_launch": alg,
           "note": "bench launches treat the accumulators as zero (no accumulator read): reads = 24 B/pixel RNG state, writes = 24 B RNG + 16 B RGBA + 4 B count + 4 B BGRA8"},
    **c4_traffic},
    open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
valu = means[(tk, "SQ_INSTS_VALU")]
c4 = {}
f4 = sorted(glob.glob(os.path.join(src, "pmc_sq_c4", "*", "*counter_collection.csv")), key=os.path.getmtime, reverse=True)
if f4:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f4[0])):
        agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    m4 = {k: sum(v) / len(v) for k, v in agg.items()}
    t4 = [k for (k, c) in m4 if "trace_kernel" in k][0]
    b4 = [k for (k, c) in m4 if "macro_bin_kernel" in k]
    kpl4 = max(1, round(((3840 + 7) // 8) * ((2160 + 7) // 8) / m4[(t4, "SQ_WAVES")]))
    for key in list(m4):
        if key[0] == t4 or (b4 and key[0] == b4[0]):
            m4[key] *= kpl4
    v4 = m4[(t4, "SQ_INSTS_VALU")] + (m4[(b4[0], "SQ_INSTS_VALU")] if b4 else 0.0)
    c4 = {"C4": {"kernels_per_launch": kpl4, "valu_wave_instructions_per_launch": int(v4), "of_which_macro_bin_kernel": int(m4[(b4[0], "SQ_INSTS_VALU")]) if b4 else 0,
                 "waves": int(m4[(t4, "SQ_WAVES")]), "valu_per_wave": round(m4[(t4, "SQ_INSTS_VALU")] / m4[(t4, "SQ_WAVES")], 1),
                 "lane_instructions_per_ray": round(v4 * 64 / (3840 * 2160 * 64), 1)}}
json.dump({
    "kernel_source_hash": khash,
    "_method": "rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE "
               "(%s, own pass). Per-launch means for the C3 trace kernel; wave-level instruction counts (one count per wave64 instruction)." % tag,
    "C3": {"kernels_per_launch": kpl, "valu_wave_instructions_per_launch": int(valu), "salu": int(means[(tk, "SQ_INSTS_SALU")]), "lds": int(means[(tk, "SQ_INSTS_LDS")]),
           "waves": int(means[(tk, "SQ_WAVES")]), "valu_per_wave": round(valu / means[(tk, "SQ_WAVES")], 1),
           "lane_instructions_per_ray": round(valu * 64 / (npix * 16), 1)}, **c4},
    open(os.path.join(dst, "valu_issue.json"), "w"), indent=1)
print("traffic %d B/launch (algorithmic %d), VALU wave-instructions/launch %d (%.0f per wave)" % (traffic, alg, valu, valu / means[(tk, "SQ_WAVES")]))

# overlap of the two half-frame kernels of a launch, from the kernel trace of the --stats pass
import subprocess
subprocess.run([sys.executable, os.path.join(root, "tools", "overlap.py"), os.path.join(src, "stats"),
                os.path.join(dst, "%s_c3_overlap.csv" % tag)], check=False)
