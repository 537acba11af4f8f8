#!/bin/bash
# durations of the dense scene's build chain (super / macro / wave lists) and trace kernels, with and without the super level
export TMPDIR=/tmp
O=$PWD/gpurun_out/chain; mkdir -p $O
for v in "" "--no-super"; do
  d=/tmp/chain_prof$v; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/c4_warm.py C4 $v > $O/run$v.log 2>&1 || exit 1
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo "== C4 $v"; grep lists $O/run$v.log; cut -d, -f1-4,6,7 $f | grep -v valu_peak | head -8
done
