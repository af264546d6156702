#!/bin/bash
# the shape-aware numbering (SHAPE_ORDER) against round 5's (class only), one variant per process: scripts/r06_shape_order.sh <out>
out=$1; : > $out
for case in ragged64 ragged20 ragged12 ragged4 poly64 poly3_64 poly20 poly4 bin100k_64 mid64; do
  for v in "new=" "old=SHAPE_ORDER:0" "new=" "old=SHAPE_ORDER:0"; do
    timeout -k 10 200 python scripts/tune_one.py $case $v 2>/dev/null >> $out
  done
done
for case in ragged4 ragged8 poly4; do
  for v in "new_sorted=SORT_LEVELS:1" "old_id=SHAPE_ORDER:0"; do
    timeout -k 10 200 python scripts/tune_one.py $case $v 2>/dev/null >> $out
  done
done
cut -c1-170 $out
