#!/bin/bash
# SQ counters of the fused eigen marginal pass: scripts/r06_eigen_pmc.sh "<levels k columns>"
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/eig_pmc_a $R/gpurun_out/eig_pmc_b
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/eig_pmc_a -o run -- python3 $R/scripts/r06_eigen_k61.py $1 0 > /dev/null 2> $R/gpurun_out/eig_pmc_a.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/eig_pmc_b -o run -- python3 $R/scripts/r06_eigen_k61.py $1 0 > /dev/null 2> $R/gpurun_out/eig_pmc_b.err
python3 - <<PY
import csv, collections
for d in ('$R/gpurun_out/eig_pmc_a', '$R/gpurun_out/eig_pmc_b'):
    rows = list(csv.DictReader(open(d + '/run_counter_collection.csv')))
    disp = collections.OrderedDict()
    for r in rows:
        dd = disp.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'].split('(')[0].replace('void ', '')[:34], 'grid': int(r['Grid_Size'])})
        dd[r['Counter_Name']] = dd.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
    # the widest launches of each mode, once
    seen = set()
    for k, v in disp.items():
        if 'eigen_gemm_kernel' in v['name']:
            big = {a: round(b / 1e6, 2) for a, b in v.items() if a not in ('name', 'grid')}
            key = (v['name'], v['grid'])
            if max(big.values()) > 20 and key not in seen:
                seen.add(key)
                print(k, v['name'], v['grid'], big)
PY
