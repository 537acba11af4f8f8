#!/bin/bash
# A/B of the list ring (slots:stride of the free events) on the all-certain frame and on C3, lists rebuilt every step.
for cfg in 3:1 4:2 6:3 8:4 8:1 8:2; do
  echo "== ring $cfg"
  RT_MI355X_LIST_RING=$cfg python3 tools/sure_floor.py 400 2>/dev/null | grep -A1 "rebuilt" | grep us_per_step
done
