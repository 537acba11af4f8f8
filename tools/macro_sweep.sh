for t in 128x64 64x32 64x64 128x32 256x64 256x128 32x32; do
  RT_MI355X_MACRO_TILE=$t python bench.py --config C4 --steps 10 --warmup 2 --no-valu --cpu-rows 0 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['value'], d['roofline']['kernel_us'])"
done
