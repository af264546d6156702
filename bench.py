#!/usr/bin/env python3
"""
Benchmark of the ML-ACR hot path on MI355X (contract: see the task description; metric of BASELINE.json).

Workload (BASELINE config 4, weak scaling): synthetic balanced tree with 1 048 576 tips (2 097 151 nodes), k = 64
states, F81 with independent frequencies per character, 32 characters per GPU (= 256 characters on 8 GPUs).
One "step" = one full marginal pass over the rank's characters: per-branch transition data, bottom-up sweep
(log-likelihoods returned to the host), top-down sweep, marginal likelihoods and posteriors for every node, left in HBM.
With N > 1 GPUs the characters are sharded over the ranks (no data-path collective); the summed log-likelihood is
all-reduced over RCCL after every step, as the host optimiser would need it.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)
BYTES_BU = 16                    # algorithmic bytes per node.state.char, bottom-up sweep
BYTES_TD = 32                    # ... top-down + marginals + posterior part of the full pass (48 - 16)

WORKLOADS = {
    # name: (tree levels, k, characters per GPU)
    'cfg4': (20, 64, 32),
    'cfg4_small': (14, 64, 8),
    'cfg2': (16, 4, 1),
}


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=5)
    p.add_argument('--warmup', type=int, default=2)
    p.add_argument('--workload', default='cfg4', choices=sorted(WORKLOADS))
    p.add_argument('--chars-per-gpu', type=int, default=None)
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-kernel-timing', action='store_true',
                   help='leave the HIP-event brackets off: the sweeps are then replayed as hipGraphs (the library\'s '
                        'default outside profiling); roofline fields are null')
    p.add_argument('--cpu-baseline-levels', type=int, default=None,
                   help='tree levels of the subtree the CPU baseline is timed on (default: about 15 s of work)')
    p.add_argument('--cpu-baseline-cores', type=int, default=None,
                   help='worker processes of the CPU baseline (default: the cores this process may run on, at most 16)')
    return p.parse_args()


def _cpu_baseline_one(job):
    """One character of the workload on a balanced subtree through the oracle (runs in a worker process: numpy only)."""
    k, levels, model, char = job
    from oracle import pastml_oracle as orc
    from pastml_amd import synthetic
    flat = synthetic.balanced_forest(levels)
    if model == 'JC':
        spec = dict(kind=0, pi=np.ones(k) / k)
    else:
        spec = dict(kind=0, pi=synthetic.f81_frequencies(k, char))
    masks = synthetic.one_hot_masks(flat, k, synthetic.tip_states(flat.n_tips, k, char)).astype(int)
    t0 = time.perf_counter()
    r = orc.full_marginal_pass(flat, masks, spec)
    return time.perf_counter() - t0, flat.n_nodes, flat.n_tips, float(r['loglik'])


def cpu_model_name():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(k, levels, model, cores=None):
    """
    PastML-style numpy CPU path (oracle/pastml_oracle.py, the per-node restatement of pastml/ml.py) timed on this
    box's host cores.  Two figures on a bounded sample of the same workload (full marginal pass of one character per
    task on a balanced subtree): one thread, and a pool of `cores` worker processes with one character each, which
    is how the reference spreads characters (acr.py:210-231; it uses threads, processes are the kinder reading).
    """
    import multiprocessing as mp
    dt1, n_nodes, n_tips, lnl = _cpu_baseline_one((k, levels, model, 0))
    single = n_nodes * k / dt1
    if cores is None:
        # a 1-GPU box gives us 16 of the host's cores (os.cpu_count() reports all of them)
        cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()))
    out = dict(value=single, unit='node*state*char/s', cores=1, kind='port',
               sample='1 character, balanced {}-tip tree ({} nodes), k={}, full marginal pass (BU+TD+posteriors), '
                      'numpy per-node port of pastml/ml.py, {:.1f} s on 1 of {} host cores'
                      .format(n_tips, n_nodes, k, dt1, os.cpu_count()),
               seconds=dt1, us_per_node=dt1 / n_nodes * 1e6, loglik=lnl, single_core_value=single,
               cpu_model=cpu_model_name(), host_cores=os.cpu_count())
    if cores > 1:
        pl = max(10, levels - 1)  # half the tree per task: the pool leg costs about half the single-thread leg
        ctx = mp.get_context('spawn')  # fresh interpreters: numpy and the oracle only
        t0 = time.perf_counter()
        with ctx.Pool(cores) as pool:
            res = pool.map(_cpu_baseline_one, [(k, pl, model, c) for c in range(cores)])
        wall = time.perf_counter() - t0
        work = max(r[0] for r in res)  # slowest task, without interpreter start-up
        pooled = sum(r[1] for r in res) * k / work
        out.update(value=pooled, cores=cores, seconds=dt1 + wall,
                   sample='{} characters in {} worker processes, one each, balanced {}-tip tree ({} nodes), k={}, full '
                          'marginal pass (BU+TD+posteriors), numpy per-node port of pastml/ml.py: slowest task {:.1f} s '
                          '(pool wall {:.1f} s); single thread on a {}-tip tree: {:.3g} units/s in {:.1f} s; host '
                          'reports {} cores'.format(cores, cores, res[0][2], res[0][1], k, work, wall, n_tips, single,
                                                    dt1, os.cpu_count()))
    return out


def main():
    args = parse_args()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node {} '
                             '--master-addr 127.0.0.1 --master-port P bench.py --gpus {} ...'.format(args.gpus, args.gpus))
        raise SystemExit('--gpus {} but WORLD_SIZE={}'.format(args.gpus, world))

    levels, k, cpg = WORKLOADS[args.workload]
    if args.chars_per_gpu:
        cpg = args.chars_per_gpu
    model = 'JC' if args.workload == 'cfg2' else 'F81'

    # CPU baseline first, while this process has not touched the GPU yet (its worker processes are started from a
    # process without a GPU context)
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cl = args.cpu_baseline_levels
        if cl is None:
            cl = min(levels, 17 if k >= 32 else 16)  # 10-20 s of single-thread numpy on the GPU box
        cpu = cpu_baseline(k, cl, model, args.cpu_baseline_cores)

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        if os.environ.get('BENCH_ALL_RANKS_ON_GPU0') is None:
            torch.cuda.set_device(local_rank)
        backend = os.environ.get('BENCH_DIST_BACKEND', 'nccl')  # 'gloo' only for dry runs of the N > 1 path
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend=backend)

    from pastml_amd import hip, synthetic
    from pastml_amd.sharding import shard_characters, allreduce_sum

    flat = synthetic.balanced_forest(levels)
    N = flat.n_nodes
    chars = list(shard_characters(cpg * world, rank, world))  # contiguous block of characters per rank

    eng = hip.Engine(flat, cpg, k, device=local_rank if os.environ.get('BENCH_ALL_RANKS_ON_GPU0') is None else 0)
    if model == 'JC':
        specs = [dict(kind=0, pi=np.ones(k) / k) for _ in chars]
    else:
        specs = [dict(kind=0, pi=synthetic.f81_frequencies(k, c)) for c in chars]
    eng.set_models([(s, (1.0, 0.0, 1.0)) for s in specs])
    eng.set_tip_states(np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in chars]))
    eng.sync()

    gpu_index = local_rank if os.environ.get('BENCH_ALL_RANKS_ON_GPU0') is None else 0
    dev = 'cuda:{}'.format(gpu_index)
    torch.zeros(1, device=dev)  # initialise torch's context on this GPU before timing
    red_dev = dev if (dist is None or dist.get_backend() == 'nccl') else 'cpu'

    def step():
        # model parameters are re-sent every step, as an optimiser iteration would: forces the per-branch
        # transition data to be recomputed inside the step
        eng.set_models([(s, (1.0, 0.0, 1.0)) for s in specs])
        lnl = eng.bottom_up(True)                       # inputs resident in HBM; returns ln L per character
        eng.top_down_marginals(posterior=False, lh=False)  # TD + marginals + posteriors, outputs stay in HBM
        # the one collective of the path: summed log-likelihood over the ranks (RCCL all-reduce of 8 bytes)
        total = allreduce_sum(float(lnl.sum()), device=red_dev)
        return total, lnl

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng.profile_enable(not args.no_kernel_timing)
    for w in (0, 1, 2):
        eng.profile_read(w, reset=True)
    fence()
    t0 = time.perf_counter()
    total = None
    for _ in range(args.steps):
        total, lnl = step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    bu_ms, bu_launches = eng.profile_read(0)
    td_ms, td_launches = eng.profile_read(1)
    prep_ms, prep_launches = eng.profile_read(2)
    held, free = eng.memory()

    if rank == 0:
        units_per_step = N * k * cpg * world
        value = units_per_step * args.steps / dt
        # dominant kernel: the top-down + marginals level kernel (td_f81_kernel): every non-root node is finished once
        td_bytes_per_step = BYTES_TD * (N - 1) * k * cpg
        bu_bytes_per_step = BYTES_BU * (N - 1) * k * cpg
        td_gbs = td_bytes_per_step * args.steps / (td_ms * 1e-3) / 1e9 if td_ms > 0 else None
        bu_gbs = bu_bytes_per_step * args.steps / (bu_ms * 1e-3) / 1e9 if bu_ms > 0 else None
        # HBM bytes per launch from the committed PMC passes (profiles/traffic.json holds bytes per step of the same
        # workload; null if the workload's shape was changed on the command line or no profile is committed)
        traffic = None
        tpath = os.path.join(REPO, 'profiles', 'traffic.json')
        if os.path.exists(tpath) and cpg == WORKLOADS[args.workload][2] and td_launches > 0:
            try:
                per_step = json.load(open(tpath)).get(args.workload, {}).get('td_bytes_per_step')
                traffic = per_step / (td_launches / args.steps) if per_step else None
            except Exception:
                traffic = None
        out = {
            'metric': 'ACR nodes*states*chars/sec (full marginal pass: P(t) + bottom-up + top-down + posteriors)',
            'value': value,
            'unit': 'node*state*char/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': '{}: balanced tree, {} tips ({} nodes), k={} states, {}, '
                                   '{} characters per GPU ({} in total), marginal (BU+TD+posteriors)'
                                   .format({'cfg4': 'BASELINE config 4 shard', 'cfg2': 'BASELINE config 2',
                                            'cfg4_small': 'reduced config 4 (16 384 tips)'}[args.workload],
                                           flat.n_tips, N, k, model, cpg, cpg * world),
                       'tips': int(flat.n_tips), 'nodes': int(N), 'states': k, 'chars_per_gpu': cpg,
                       'chars_total': cpg * world, 'substitution_model': model,
                       'sharding': 'characters over ranks, no data-path collective; 1 all-reduce (8 B) per step'},
            'loglik_sum': total,
            'roofline': {
                'bound': 'hbm', 'kernel': 'td_f81_kernel (top-down + marginals + posteriors, one launch per depth level)',
                'achieved': td_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': (td_gbs / HBM_PEAK_GBS) if td_gbs else None,
                'traffic': traffic,
                # real HBM rate of the kernel = PMC bytes per launch / measured launch time (the cherry-fused schedule
                # moves fewer bytes than the algorithmic 32 B/unit model, hence achieved can exceed the real rate)
                'traffic_gbs': (traffic / (td_ms / max(1, td_launches) * 1e-3) / 1e9) if traffic and td_ms > 0 else None,
                'traffic_frac': (traffic / (td_ms / max(1, td_launches) * 1e-3) / 1e9 / HBM_PEAK_GBS)
                if traffic and td_ms > 0 else None,
                'algorithmic_bytes_per_launch': td_bytes_per_step / max(1, td_launches / args.steps),
                'avg_launch_ms': td_ms / max(1, td_launches),
                'launches': td_launches,
                'bytes_per_unit': BYTES_TD,
            },
            'roofline_bottom_up': {
                'kernel': 'bu_f81_kernel (bottom-up, one launch per height level)', 'achieved': bu_gbs,
                'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': (bu_gbs / HBM_PEAK_GBS) if bu_gbs else None,
                'avg_launch_ms': bu_ms / max(1, bu_launches), 'launches': bu_launches, 'bytes_per_unit': BYTES_BU,
            },
            'kernel_ms_per_step': {'bottom_up': bu_ms / args.steps, 'top_down': td_ms / args.steps,
                                   'prep': prep_ms / args.steps},
            'device_memory_gb': held / 1e9,
        }
        if cpu is not None:
            out['cpu_baseline'] = cpu
            out['speedup_vs_cpu_baseline'] = value / cpu['value']
            out['speedup_vs_cpu_baseline_single_core'] = value / cpu['single_core_value']
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
