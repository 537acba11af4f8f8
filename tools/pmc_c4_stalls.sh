#!/bin/bash
# Where the waves of the dense-scene trace kernel spend their cycles (kept lists: the trace kernels alone), two PMC passes.
#   bash tools/pmc_c4_stalls.sh [C4]      (on the GPU box, from the repo root; separate --pmc passes, kernel trace only)
set -o pipefail
export TMPDIR=/tmp
CFG=${1:-C4}
S=/tmp/rt_pmc_stalls; rm -rf $S; mkdir -p $S gpurun_out/pmc_stalls
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $S/a -- python3 tools/c4_warm.py $CFG > gpurun_out/pmc_stalls/a.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $S/b -- python3 tools/c4_warm.py $CFG > gpurun_out/pmc_stalls/b.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU --output-format csv -d $S/c -- python3 tools/c4_warm.py $CFG > gpurun_out/pmc_stalls/c.log 2>&1 || exit 1
python3 - <<'PY'
import collections, csv, glob
for sub in "abc":
    f = sorted(glob.glob("/tmp/rt_pmc_stalls/%s/**/*counter_collection.csv" % sub, recursive=True))[-1]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
    for k in agg:
        if "trace_kernel" in k or "lists_kernel" in k or "macro" in k:
            print(k, "dispatches", n[k], {c: "%.4g" % (v / max(n[k], 1)) for c, v in sorted(agg[k].items())})
PY
