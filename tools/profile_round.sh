#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01_f
# Passes are separate runs: --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; --pmc SQ_*.
# Output under gpurun_out/<tag>/; summarise afterwards with tools/pmc_summary.py <tag>.
set -o pipefail
TAG=${1:-r01_x}
export TMPDIR=/tmp
O=$PWD/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
B="python3 bench.py --cpu-rows 0 --no-valu --no-warm"   # profiled runs: headline launches only
python3 bench.py > "$O/bench_c3.json" 2> "$O/bench_c3.err" || exit 1
python3 bench.py --config C4 --steps 20 --warmup 3 > "$O/bench_c4.json" 2> "$O/bench_c4.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- $B --steps 50 --warmup 5 > "$O/stats.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- $B --steps 5 --warmup 1 > "$O/pmc_fetch.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- $B --steps 5 --warmup 1 > "$O/pmc_write.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$O/pmc_sq" -- $B --steps 5 --warmup 1 > "$O/pmc_sq.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$O/pmc_sq_c4" -- $B --config C4 --steps 3 --warmup 1 > "$O/pmc_sq_c4.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_c4" -- $B --config C4 --steps 10 --warmup 2 > "$O/stats_c4.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_c4" -- $B --config C4 --steps 3 --warmup 1 > "$O/pmc_fetch_c4.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_c4" -- $B --config C4 --steps 3 --warmup 1 > "$O/pmc_write_c4.log" 2>&1 || exit 1
grep '^{"metric"' "$O/stats.log" > "$O/bench_c3_under_rocprof.json"
echo "done: $O"
