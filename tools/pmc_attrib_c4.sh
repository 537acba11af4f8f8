#!/bin/bash
# Attribution of the dense-scene kernel's HBM traffic above the algorithmic bytes: FETCH_SIZE / WRITE_SIZE per launch for the K = 4 product
# kernel (scratch) and the K = 2 variant (no scratch).  bash tools/pmc_attrib_c4.sh
set -o pipefail
export TMPDIR=/tmp
B="python3 bench.py --cpu-rows 0 --no-valu --no-warm --no-parity --config C4 --steps 3 --warmup 1"
for v in k4 k2; do
  X=""; [ $v = k2 ] && X="--samples-in-flight 2"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${v}_$c -- $B $X > /dev/null 2>&1
    python3 - <<PY
import csv,glob,collections
f=sorted(glob.glob('/tmp/pmc_${v}_$c/**/*counter_collection.csv',recursive=True))[-1]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name']=='$c': agg[r['Kernel_Name'].split('(')[0][-60:]].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'trace_kernel' in k or 'macro' in k: print('$v $c',k,len(v),'mean KB',round(sum(v)/len(v),1),'total per launch MB', round(sum(v)/ (len([1 for kk in agg if 'trace_kernel' in kk]) and (len(agg[[kk for kk in agg if 'trace_kernel' in kk][0]])/2.0))/1024,1))
PY
  done
done
