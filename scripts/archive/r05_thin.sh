#!/bin/bash
# thin ends as subtree blocks: parity test, then the ragged secondaries A/B (NO_THIN against the default)
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "thin_ends or height_order or td_tail or fuzz or random" > gpurun_out/r05u_tests.txt 2>&1 || { tail -40 gpurun_out/r05u_tests.txt; exit 1; }
tail -3 gpurun_out/r05u_tests.txt
: > gpurun_out/r05u_thin_ab.txt
for case in ${CASES:-ragged4 ragged12 ragged2 poly4 mid4}; do
  timeout -k 10 300 python scripts/r05_tune_ab.py $case default= off=NO_THIN:1 s128=THIN_BLOCK_NODES:128 s512=THIN_BLOCK_NODES:512 s1k=THIN_BLOCK_NODES:1024 t2k=THIN_UNITS:2048 t1k=THIN_UNITS:1024 >> gpurun_out/r05u_thin_ab.txt 2>&1
done
cat gpurun_out/r05u_thin_ab.txt
