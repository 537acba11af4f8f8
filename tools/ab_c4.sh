# alternating A/B of two library builds on C4: bash tools/ab_c4.sh libA.so libB.so
for rep in 1 2 3; do for l in "$1" "$2"; do
  RT_MI355X_LIB=$PWD/$l python bench.py --config C4 --steps 10 --warmup 2 --no-valu --cpu-rows 0 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['ms_per_step'])"
done; done
