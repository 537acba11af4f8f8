#!/bin/bash
# One PMC pass (wave-level instruction counts and cycles) over a few headline steps; prints per-kernel means.
#   bash tools/pmc_quick.sh [bench.py arguments]
set -o pipefail
export TMPDIR=/tmp
S=/tmp/rt_pmc_quick; rm -rf "$S"; mkdir -p "$S" gpurun_out
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$S" -- python3 bench.py --cpu-rows 0 --no-valu --no-warm --no-parity --steps 5 --warmup 1 "$@" > gpurun_out/pmc_quick.log 2>&1 || { tail -5 gpurun_out/pmc_quick.log; exit 1; }
python3 - "$S" <<'PY'
import csv, glob, os, sys, collections
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-60:]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "trace_kernel" in k or "lists_kernel" in k or "macro" in k:
        n = len(next(iter(c.values())))
        m = {cn: sum(v) / len(v) for cn, v in c.items()}
        us = m.get("GRBM_GUI_ACTIVE", 0) / 8 / 2.36e3
        print("%s  x%d  waves %.0f  VALU %.3fM  SALU %.3fM  LDS %.3fM  VALU/wave %.0f  ~%.1f us alone -> %.0f Ginst/s" % (
            k, n, m.get("SQ_WAVES", 0), m.get("SQ_INSTS_VALU", 0) / 1e6, m.get("SQ_INSTS_SALU", 0) / 1e6, m.get("SQ_INSTS_LDS", 0) / 1e6,
            m.get("SQ_INSTS_VALU", 0) / max(1, m.get("SQ_WAVES", 1)), us, m.get("SQ_INSTS_VALU", 0) / max(us, 1e-9) / 1e3))
PY
