#!/usr/bin/env python3
"""Overlap of the two half-frame trace kernels of a split launch, from a rocprofv3 --kernel-trace run.

  python tools/overlap.py gpurun_out/<tag>/stats profiles/<tag>_c3_overlap.csv

A launch of a tall frame runs as two half-frame kernels on two streams (DESIGN.md 4.1 "Split launches").  The
kernel trace lists every dispatch with its start/end timestamp and queue; consecutive trace_kernel dispatches on
different queues whose intervals intersect are the two halves of one launch.  Per launch: the two durations, the
time both were running, the span from the first start to the last end; last row: the means and the overlap
fraction = both-running time / span -- what turns the per-dispatch average of rocprofv3 --stats into a per-launch time."""
import csv, glob, os, sys

src, dst = sys.argv[1], sys.argv[2]
files = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
if not files:
    sys.exit("no *kernel_trace.csv under " + src)
rows = []
for r in csv.DictReader(open(files[-1])):
    if "trace_kernel" in r["Kernel_Name"]:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
rows.sort()
launches, i = [], 0
while i < len(rows):
    a = rows[i]
    if i + 1 < len(rows) and rows[i + 1][2] != a[2] and rows[i + 1][0] < a[1]:
        b = rows[i + 1]
        both = max(0, min(a[1], b[1]) - max(a[0], b[0]))
        launches.append((a[0], a[1] - a[0], b[1] - b[0], both, max(a[1], b[1]) - a[0], a[2], b[2]))
        i += 2
    else:
        launches.append((a[0], a[1] - a[0], 0, 0, a[1] - a[0], a[2], ""))
        i += 1
t0 = launches[0][0] if launches else 0
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["launch", "start_us", "half_a_us", "half_b_us", "both_running_us", "span_us", "queue_a", "queue_b"])
    for n, l in enumerate(launches):
        w.writerow([n, "%.2f" % ((l[0] - t0) / 1e3)] + ["%.2f" % (v / 1e3) for v in l[1:5]] + [l[5], l[6]])
    pairs = [l for l in launches if l[2] > 0]
    if pairs:
        m = [sum(l[k] for l in pairs) / len(pairs) / 1e3 for k in (1, 2, 3, 4)]
        gaps = [(pairs[k + 1][0] - pairs[k][0]) / 1e3 for k in range(len(pairs) - 1)]
        w.writerow(["mean of %d split launches" % len(pairs), "", "%.2f" % m[0], "%.2f" % m[1], "%.2f" % m[2], "%.2f" % m[3],
                    "overlap_fraction=%.3f" % (m[2] / m[3]),
                    "launch_to_launch_us=%.2f" % (sum(gaps) / max(len(gaps), 1))])
        print("split launches %d: halves %.1f / %.1f us, both running %.1f us, span %.1f us, overlap %.3f, launch-to-launch %.1f us"
              % (len(pairs), m[0], m[1], m[2], m[3], m[2] / m[3], sum(gaps) / max(len(gaps), 1)))
