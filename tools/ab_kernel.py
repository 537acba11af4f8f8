#!/usr/bin/env python3
"""A/B of two builds of the library on the bench workload, alternating in separate processes:
  python tools/ab_kernel.py libA.so libB.so [rounds] [config]"""
import os, subprocess, sys, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = sys.argv[4] if len(sys.argv) > 4 else "C3"
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, RT_MI355X_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-valu", "--cpu-rows", "0", "--config", cfg],
                             env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        res[l].append(json.loads(out)["roofline"]["kernel_us"])
for l in libs:
    v = res[l]
    print("%-50s kernel us: %s  min %.1f  mean %.1f" % (os.path.basename(l), " ".join("%.1f" % x for x in v), min(v), sum(v) / len(v)))
