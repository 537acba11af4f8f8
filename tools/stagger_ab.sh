#!/bin/bash
# Short runs from idle (the driver's 5 warmup + 20 timed steps) and long runs, us per step of N runs each:
#   stagger of the first split launch by a delay kernel (default) / behind the upper half's end / none.
# Phase lock shows as a bimodal distribution of the long runs.  bash tools/stagger_ab.sh [runs]
N=${1:-8}
for mode in delay event none; do
  case $mode in delay) E="";; event) E="RT_MI355X_STAGGER_EVENT=1";; none) E="RT_MI355X_NO_STAGGER=1";; esac
  echo "== stagger: $mode"
  for k in 20:5 400:20; do
    out=""
    for i in $(seq $N); do
      v=$(env $E python3 bench.py --steps ${k%%:*} --warmup ${k##*:} --no-valu --cpu-rows 0 --no-warm --no-parity 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.readline())['ms_per_step']*1000,1))")
      out="$out $v"
    done
    echo "  steps $k:$out"
  done
done
