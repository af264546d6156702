#!/bin/bash
# Profiles one bench workload on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats, then the two PMC passes for HBM traffic (separate runs, as the guide prescribes).
# Outputs land in gpurun_out/{prof_kt,pmc_fetch,pmc_write}; condense them with profiles/summarize_rocprof.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-cfg4}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_kt $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
python3 $R/bench.py --workload $W --steps 10 --warmup 3 --no-secondary > $R/gpurun_out/bench_$W.json 2> $R/gpurun_out/bench_$W.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -o run -- python3 $R/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_kt.json 2> $R/gpurun_out/prof_kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o run -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $R/gpurun_out/pmc_fetch.json 2> $R/gpurun_out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o run -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $R/gpurun_out/pmc_write.json 2> $R/gpurun_out/pmc_write.err
ls $R/gpurun_out/prof_kt $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
tail -c 400 $R/gpurun_out/bench_$W.json
