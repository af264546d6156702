#!/bin/bash
out=$1; : > $out
for case in ragged64 ragged20 ragged12 ragged8 ragged4 ragged2 bigpoly3_64 poly64 poly4 mid64 bin100k_64 bin100k_4; do
  for v in "default=" "old=SHAPE_ORDER:0"; do
    timeout -k 10 200 python scripts/tune_one.py $case $v 2>/dev/null >> $out
  done
done
cut -c1-170 $out
