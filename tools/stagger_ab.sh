#!/bin/bash
# Short runs from idle (the driver's 5 warmup + 20 timed steps) with and without the stagger of the first split launch:
# us per step of N runs each, and long runs (phase lock shows as a bimodal distribution).  bash tools/stagger_ab.sh [runs]
N=${1:-8}
for s in 0 1; do
  echo "== RT_MI355X_NO_STAGGER=$s: $N x (--steps 20 --warmup 5), then $N x (--steps 400 --warmup 20)"
  for k in 20:5 400:20; do
    out=""
    for i in $(seq $N); do
      v=$(RT_MI355X_NO_STAGGER=$s python3 bench.py --steps ${k%%:*} --warmup ${k##*:} --no-valu --cpu-rows 0 --no-warm --no-parity 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.readline())['ms_per_step']*1000,1))")
      out="$out $v"
    done
    echo "  steps $k:$out"
  done
done
