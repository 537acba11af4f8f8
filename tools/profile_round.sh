#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r03_a
# Passes are separate runs: --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; --pmc SQ_*  (never combined with
# other trace domains).  Order: profiler passes -> tools/pmc_summary.py (writes profiles/hbm_traffic.json and
# profiles/valu_issue.json stamped with the kernel hash) -> the bench lines, so that every line this script keeps carries the
# PMC figures of the build it ran on.  Everything to keep ends up under gpurun_out/<tag>/profiles/ (copy it into profiles/).
set -o pipefail
TAG=${1:-r04_x}
export TMPDIR=/tmp
O=$PWD/gpurun_out/$TAG
S=/tmp/rt_prof_$TAG                       # raw profiler output stays on the box (the kernel traces are large)
rm -rf "$O" "$S"; mkdir -p "$O" "$S"
B="python3 bench.py --cpu-rows 0 --no-valu --no-warm --no-parity --no-c4 --no-e2e"   # profiled runs: headline launches only
python3 bench.py --no-valu --cpu-rows 0 --no-c4 --no-e2e > "$O/bench_c3.json" 2> "$O/bench_c3.err" || exit 1      # (library hash for the summary)
echo "stats C3";  rocprofv3 --kernel-trace --stats --output-format csv -d "$S/stats" -- $B --steps 50 --warmup 5 > "$O/stats.log" 2>&1 || exit 1
echo "pmc C3";    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$S/pmc_fetch" -- $B --steps 5 --warmup 1 > "$O/pmc_fetch.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$S/pmc_write" -- $B --steps 5 --warmup 1 > "$O/pmc_write.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$S/pmc_sq" -- $B --steps 5 --warmup 1 > "$O/pmc_sq.log" 2>&1 || exit 1
echo "pmc C4";    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$S/pmc_sq_c4" -- $B --config C4 --steps 3 --warmup 1 > "$O/pmc_sq_c4.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$S/stats_c4" -- $B --config C4 --steps 10 --warmup 2 > "$O/stats_c4.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$S/pmc_fetch_c4" -- $B --config C4 --steps 3 --warmup 1 > "$O/pmc_fetch_c4.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$S/pmc_write_c4" -- $B --config C4 --steps 3 --warmup 1 > "$O/pmc_write_c4.log" 2>&1 || exit 1
grep '^{"metric"' "$O/stats.log" > "$O/bench_c3_under_rocprof.json"
python3 tools/step_periods.py "$S/stats" > "profiles/${TAG}_c3_step_periods.txt" 2>&1 || true
echo "summary";   python3 tools/pmc_summary.py "$TAG" "$S" > "$O/pmc_summary.log" 2>&1 || { cat "$O/pmc_summary.log"; exit 1; }
echo "bench";     python3 bench.py > "profiles/${TAG}_bench_c3.json" 2> "$O/bench_c3_final.err" || exit 1
python3 bench.py --steps 20 --warmup 5 > "profiles/${TAG}_bench_c3_steps20.json" 2>> "$O/bench_c3_final.err" || exit 1
python3 bench.py --config C4 --steps 30 --warmup 5 > "profiles/${TAG}_bench_c4.json" 2> "$O/bench_c4_final.err" || exit 1
mkdir -p "$O/profiles"
cp profiles/${TAG}_* profiles/hbm_traffic.json profiles/valu_issue.json "$O/profiles/"
cat "$O/pmc_summary.log"
echo "done: $O/profiles"
