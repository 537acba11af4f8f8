#!/usr/bin/env python3
"""Step period over time from a rocprofv3 --kernel-trace run: launch-to-launch distance of the trace kernels of one queue,
averaged over windows of 20 steps, next to the list builder's duration and its lead over the trace kernel that consumes it.
  python tools/step_periods.py gpurun_out/<dir>"""
import csv, glob, os, sys
src = sys.argv[1]
f = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
T, L = {}, []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    if "trace_kernel" in n:
        T.setdefault(q, []).append((s, e))
    elif "lists_kernel" in n:
        L.append((s, e))
qs = sorted(T, key=lambda q: -len(T[q]))[:2]
a = sorted(T[qs[0]])
b = sorted(T[qs[1]]) if len(qs) > 1 else []
L.sort()
print("trace kernels per queue:", {q: len(T[q]) for q in T}, "list builds:", len(L))
W = 20
for i in range(0, len(a) - W, W):
    per = (a[i + W][0] - a[i][0]) / W / 1e3
    da = sum(e - s for s, e in a[i:i + W]) / W / 1e3
    db = sum(e - s for s, e in b[i:i + W]) / W / 1e3 if len(b) >= i + W else 0.0
    dl = sum(e - s for s, e in L[i:i + W]) / W / 1e3 if len(L) >= i + W else 0.0
    # lead: end of list build k to start of trace kernel k on the first queue (builds and launches are 1:1 in the headline loop)
    lead = sum(a[k][0] - L[k][1] for k in range(i, min(i + W, len(L)))) / W / 1e3 if len(L) >= i + W else 0.0
    gap_a = sum(a[k + 1][0] - a[k][1] for k in range(i, i + W)) / W / 1e3
    off = sum(b[k][0] - a[k][0] for k in range(i, min(i + W, len(b)))) / W / 1e3 if len(b) >= i + W else 0.0
    print("steps %4d..%4d: period %6.1f us  trace A %5.1f  B %5.1f  gap on A %5.1f  build %5.1f  build->trace lead %7.1f  B starts %6.1f after A" % (i, i + W, per, da, db, gap_a, dl, lead, off))
