#!/bin/bash
# A/B of builds of the dense build chain (libraries built with EXTRA=...): C4 step time, lists rebuilt / kept
for l in librt_mi355x.so librt_sf2.so librt_sf8.so librt_prio.so librt_mi355x.so; do
  echo "== $l"; RT_MI355X_LIB=$PWD/raytracertest_amd/lib/$l python3 tools/c4_warm.py 2>&1 | grep lists
done
