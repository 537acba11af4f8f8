#!/usr/bin/env python3
"""Dispatch timeline of a rocprofv3 --kernel-trace run: per dispatch the short kernel name, queue, start relative to the
first listed dispatch and duration (us); then per kernel name the mean duration.
  python tools/kernel_timeline.py gpurun_out/<dir> [first] [count]"""
import csv, glob, os, re, sys
src = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
files = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
if not files:
    sys.exit("no *kernel_trace.csv under " + src)
rows = []
for r in csv.DictReader(open(files[-1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"])
    name = re.sub(r"^void rtk::", "", name)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), name))
rows.sort()
sel = rows[first:first + count]
t0 = sel[0][0] if sel else 0
for s, e, q, n in sel:
    print("%10.2f  +%8.2f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n[:90]))
by = {}
for s, e, q, n in rows:
    by.setdefault(n, []).append((e - s) / 1e3)
print()
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print("%8d x %9.2f us mean  (total %10.1f us)  %s" % (len(v), sum(v) / len(v), sum(v), n[:100]))
if rows:
    print("span of all dispatches: %.1f us" % ((max(r[1] for r in rows) - rows[0][0]) / 1e3))
