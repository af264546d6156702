#!/usr/bin/env python3
"""
Benchmark of the ML-ACR hot path on MI355X (contract: see the task description; metric of BASELINE.json).

Workload (BASELINE config 4, weak scaling): synthetic balanced tree with 1 048 576 tips (2 097 151 nodes), k = 64
states, F81 with independent frequencies per character, 32 characters per GPU (= 256 characters on 8 GPUs).
One "step" = one full marginal pass over the rank's characters: model parameters to the device, per-branch transition
data, bottom-up sweep (log-likelihoods returned to the host), top-down sweep, marginal likelihoods and posteriors for
every node (left in HBM), and -- the one collective of the path -- the all-reduce of the summed log-likelihood over
RCCL, issued by the library on the sweep's own stream (pml_allreduce_loglik).

N > 1: one process per GPU.  Either the launcher provides RANK / LOCAL_RANK / WORLD_SIZE (torch.distributed.run), or
`python bench.py --gpus N` starts the N rank processes itself (before anything touches a GPU) and relays rank 0's line.

Prints ONE JSON line on rank 0.  After the timed region the posteriors of EVERY column are validated on a strided
node sample (rows sum to one, every node sees the column's total likelihood, observed tips keep their state), and,
on one GPU, the other BASELINE configs are timed in the same process (`secondary`).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)
FP64_MFMA_PEAK_TFLOPS = 78.6     # dense FP64 matrix-core peak (same guide)
# SURVEY.md 8d's bytes per node.state.char of the REFERENCE's schedule (every vector of every node written and read
# back): kept only to say how much traffic the schedule below avoids -- never used for `frac`
REFERENCE_SCHEDULE_BYTES = {'bottom_up': 16, 'top_down': 32}
# Schedule-independent yardsticks printed next to `frac` (which rests on this schedule's own byte model):
HBM_ACHIEVABLE_GBS = 6300.0      # the guide's attainable HBM bandwidth (hipMemsetAsync here: 6.2 TB/s, profiles/r05i_write_streams.txt)
OUTPUT_FLOOR_BYTES_PER_UNIT = 8  # every posterior entry written once: no schedule of a full marginal pass moves less
SURVEY_8D_BYTES_PER_UNIT = 48    # the reference schedule's algorithmic bytes of a full marginal pass

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
    p.add_argument('--no-secondary', action='store_true', help='skip the cfg2 / cfg3 / cfg5-shaped timings')
    p.add_argument('--no-validate', action='store_true', help='skip the post-run validation of every column')
    p.add_argument('--no-kernel-timing', action='store_true',
                   help='leave the HIP-event brackets off: the sweeps are then replayed as hipGraphs (the library\'s '
                        'default outside profiling); roofline fields are null')
    p.add_argument('--cpu-baseline-levels', type=int, default=None,
                   help='tree levels of the subtree the CPU baseline is timed on (default 18: 262 144 tips)')
    p.add_argument('--cpu-baseline-cores', type=int, default=None,
                   help='worker processes of the CPU baseline (default: the cores this process may run on, at most 32)')
    return p.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle as the checker/baseline: never part of the product path)
# ---------------------------------------------------------------------------------------------------------------------
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


def under_profiler():
    """rocprofv3 preloads its tool library, which initialises the GPU before main(): no child processes then."""
    env = os.environ
    if any(k.startswith(('ROCPROF', 'ROCPROFILER', 'ROCP_')) for k in env):
        return True
    return any(t in env.get('LD_PRELOAD', '') for t in ('rocprof', 'roctracer', 'roctx'))


def cpu_baseline(k, levels, model, cores=None):
    """
    PastML-style numpy CPU path (oracle/pastml_oracle.py, the per-node restatement of pastml/ml.py) timed on this
    box's host cores, after BASELINE.md section 3: one thread (us per node), and a pool of worker processes with one
    character each, which is how the reference spreads characters (acr.py:210-231; it uses threads, processes are the
    kinder reading).  Workers = the cores this process may run on, at most 32; both legs on the same tree of `levels`
    levels (default 18: 262 144 tips -- about 20 s per character on the GPU box's EPYC; the full 1 048 576-tip tree of
    section 3 would take the reference 359 s per character, tests/golden/make_golden.py).
    """
    import multiprocessing as mp
    dt1, n_nodes, n_tips, lnl = _cpu_baseline_one((k, levels, model, 0))
    single = n_nodes * k / dt1
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
    if cores is None:
        cores = max(1, min(32, affinity))   # (os.cpu_count() reports the whole host; a 1-GPU box gives us a share of it)
    out = dict(value=single, unit='node*state*char/s', cores=1, kind='port',
               sample='1 character, balanced {}-tip tree ({} nodes), k={}, full marginal pass (BU+TD+posteriors), '
                      'numpy per-node port of pastml/ml.py, {:.1f} s on 1 of {} host cores'
                      .format(n_tips, n_nodes, k, dt1, os.cpu_count()),
               seconds=dt1, us_per_node=dt1 / n_nodes * 1e6, loglik=lnl, single_core_value=single,
               cpu_model=cpu_model_name(), host_cores=os.cpu_count(), affinity_cores=affinity, tips=int(n_tips))
    if cores > 1:
        ctx = mp.get_context('spawn')  # fresh interpreters: numpy and the oracle only
        t0 = time.perf_counter()
        with ctx.Pool(cores) as pool:
            res = pool.map(_cpu_baseline_one, [(k, levels, model, c) for c in range(cores)], chunksize=1)
        wall = time.perf_counter() - t0
        work = max(r[0] for r in res)  # slowest task, without interpreter start-up
        pooled = sum(r[1] for r in res) * k / work
        out.update(value=pooled, cores=cores, seconds=dt1 + wall, pool_us_per_node=work / res[0][1] * 1e6,
                   sample='{} characters in {} worker processes (= the {} cores this process may run on, at most 32), one '
                          'each, balanced {}-tip tree ({} nodes), k={}, full marginal pass (BU+TD+posteriors), numpy '
                          'per-node port of pastml/ml.py: slowest task {:.1f} s = {:.1f} us/node (pool wall {:.1f} s); '
                          'single thread on the same tree: {:.1f} us/node, {:.3g} units/s in {:.1f} s; host reports {} '
                          'cores'.format(cores, cores, affinity, res[0][2], res[0][1], k, work, work / res[0][1] * 1e6,
                                         wall, dt1 / n_nodes * 1e6, single, dt1, os.cpu_count()))
    # BASELINE.md section 3 asks for 2 x cores characters at FULL tree size: the reference needs 359 s per character there
    # (tests/golden/make_golden.py, 1 048 576 tips: 171 us per node), so a bounded sample on a quarter of the tips still
    # flatters the CPU and the speed-up computed from it is a lower bound
    out['sample'] += ('; a quarter of the full tree\'s tips: the reference itself took 359 s per character at 1 048 576 '
                      'tips (171 us/node, tests/golden/make_golden.py) -- the speed-up from this sample is a lower bound')
    return out


# ---------------------------------------------------------------------------------------------------------------------
# compulsory HBM bytes of THIS schedule, per column and sweep
# ---------------------------------------------------------------------------------------------------------------------
def two_level_nodes(flat, k, n_cols):
    """
    The nodes the library's level schedule runs as two-level units (pml_tree_upload): stored nodes with two
    stored children that each carry two cherries of two tips, ids of the four cherries and of the eight tips consecutive;
    only for lane groups of 8 and more (17 <= k <= 64), where they are at least 64 and a sixteenth of the stored nodes,
    and only where the sweeps run level launches (not one launch per sweep, not subtree blocks).  Returns a boolean
    array over the nodes (empty selection if the schedule does not apply).
    """
    N = flat.n_nodes
    nc = np.asarray(flat.n_children)
    fc = np.asarray(flat.first_child)
    parent = np.asarray(flat.parent)
    sup = np.zeros(N, dtype=bool)
    if not 17 <= k <= 64 or os.environ.get('PASTML_HIP_NO_SUPER'):
        return sup
    internal = nc > 0
    tip = ~internal
    n_tip_children = np.zeros(N, dtype=np.int64)
    np.add.at(n_tip_children, parent[(parent >= 0) & tip], 1)
    cherry = internal & (n_tip_children == nc) & (parent >= 0)
    stored = internal & ~cherry
    n_stored = int(stored.sum())
    if N <= 2048 or (256 < n_stored <= 131072 and n_stored * n_cols <= 160000):
        return sup
    two = np.flatnonzero(stored & (nc == 2))
    a = fc[two]
    ok = cherry[a] & cherry[a + 1] & (nc[a] == 2) & (nc[a + 1] == 2) & (fc[a + 1] == fc[a] + 2)
    pair = np.zeros(N, dtype=bool)
    pair[two[ok]] = True
    ok = pair[a] & pair[a + 1]
    cand, a = two[ok], a[ok]
    ok = (fc[a + 1] == fc[a] + 2) & (fc[fc[a + 1]] == fc[fc[a]] + 4)
    sup[cand[ok]] = True
    if sup.sum() < 64 or sup.sum() * 16 < n_stored:
        sup[:] = False
    return sup


def fused_height(flat, stored):
    """Fused height of the stored nodes: 1 + the largest height among stored children (deepest nodes first)."""
    N = flat.n_nodes
    nc = np.asarray(flat.n_children)
    fc = np.asarray(flat.first_child)
    depth = np.asarray(flat.depth)
    fh = np.zeros(N, dtype=np.int64)
    for d in range(int(depth.max()), -1, -1):
        idx = np.flatnonzero(stored & (depth == d))
        if not len(idx):
            continue
        h = np.zeros(len(idx), dtype=np.int64)
        for j in range(int(nc[idx].max())):
            has = nc[idx] > j
            ch = fc[idx[has]] + j
            h[has] = np.maximum(h[has], np.where(stored[ch], fh[ch], 0))
        fh[idx] = h + 1
    return fh


def stacked_nodes(flat, stored, sup, absorbed, fh=None):
    """
    The nodes the library runs as stacked units (pml_tree_upload) once the two-level nodes `sup` (whose
    children `absorbed` have no vector in memory) are known: in ascending fused height, a node with two children that are
    plain units with two stored children each (vectors in memory) takes its children over -- on levels of 1 024 to
    65 536 nodes, and only where the stacked units take over at least half of what the two-level units leave.
    Returns (stacked, taken) boolean arrays.
    """
    N = flat.n_nodes
    nc = np.asarray(flat.n_children)
    fc = np.asarray(flat.first_child)
    depth = np.asarray(flat.depth)
    stacked = np.zeros(N, dtype=bool)
    taken = np.zeros(N, dtype=bool)
    if os.environ.get('PASTML_HIP_NO_STACK') or not stored.any():
        return stacked, taken
    if fh is None:
        fh = fused_height(flat, stored)
    level_size = np.bincount(fh[stored], minlength=int(fh.max()) + 1)
    novec = absorbed.copy()
    gone = sup | absorbed
    lo = int(os.environ.get('PASTML_HIP_STACK_MIN', 1024))
    for h in range(2, int(fh.max()) + 1):
        if not lo <= level_size[h] <= 65536:
            continue
        cand = np.flatnonzero(stored & (fh == h) & ~gone & ~taken & (nc == 2))
        if not len(cand):
            continue
        ok = np.ones(len(cand), dtype=bool)
        for j in (0, 1):
            ch = fc[cand] + j
            plain = stored[ch] & ~gone[ch] & ~stacked[ch] & ~taken[ch] & (nc[ch] == 2) & (level_size[fh[ch]] <= 65536)
            g = fc[ch]
            g0 = np.where(plain, g, 0)
            plain &= stored[g0] & ~novec[g0] & stored[g0 + 1] & ~novec[g0 + 1]
            ok &= plain
        n = cand[ok]
        stacked[n] = True
        for j in (0, 1):
            taken[fc[n] + j] = True
            novec[fc[n] + j] = True
    if 'PASTML_HIP_STACK_MIN' not in os.environ and int(stacked.sum()) * 6 < int(stored.sum()) - 3 * int(sup.sum()):
        stacked[:] = False
        taken[:] = False
    return stacked, taken


def schedule_bytes(flat, k, n_cols, narrow_limit=None):
    """
    Bytes the F81-family kernels must move per sweep for one column, from the tree itself: only *stored* internal
    nodes (internal nodes that are not cherries; roots always) have a bottom-up vector in HBM, tips are 8-byte masks,
    cherries are rebuilt in registers, top-down vectors are never written, every node's posterior is written once.
    Two-level units (two_level_nodes): the two children of such a node have no bottom-up vector in HBM either, and their
    posterior rows are not read back.  Returns per-sweep totals and the parts of the level kernels and of the two-level
    launches.
    """
    ks = k + (k & 1) if k >= 2 else k
    W = (k + 63) // 64
    vec = 8 * ks
    N = flat.n_nodes
    nc = np.asarray(flat.n_children)
    fc = np.asarray(flat.first_child)
    parent = np.asarray(flat.parent)
    internal = nc > 0
    tip = ~internal
    # cherry: non-root internal node whose children are all tips
    n_tip_children = np.zeros(N, dtype=np.int64)
    np.add.at(n_tip_children, parent[(parent >= 0) & tip], 1)
    cherry = internal & (n_tip_children == nc) & (parent >= 0)
    stored = internal & ~cherry
    sup = two_level_nodes(flat, k, n_cols)
    n_sup = int(sup.sum())
    gone = sup.copy()                                   # nodes that are not units of the level lists any more
    gone[fc[sup]] = True
    gone[fc[sup] + 1] = True
    absorbed = gone & ~sup
    wide = 17 <= k <= 64 and not os.environ.get('PASTML_HIP_NO_SUPER')
    level_schedule = not (N <= 2048 or (256 < int(stored.sum()) <= 131072 and int(stored.sum()) * n_cols <= 160000))
    fh = fused_height(flat, stored) if wide else None
    stacked, taken = (stacked_nodes(flat, stored, sup, absorbed, fh) if wide
                      else (np.zeros(N, dtype=bool), np.zeros(N, dtype=bool)))
    if n_sup == 0 and stacked.any() and not level_schedule:
        stacked[:] = False                              # (the level schedule is not used at all: two_level_nodes' rule)
        taken[:] = False
    n_stack = int(stacked.sum())
    gone = gone | stacked | taken
    unit = stored & ~gone                               # units of the level kernels
    nonroot = parent >= 0
    pmask = np.zeros(N, dtype=bool)
    pmask[nonroot] = unit[parent[nonroot]]              # node is a child of a level-kernel unit
    gp = np.where(nonroot, parent, 0)
    cmask = np.zeros(N, dtype=bool)
    cmask[nonroot] = cherry[parent[nonroot]] & pmask[gp[nonroot]]   # tip of a cherry child of a level-kernel unit
    n_units = int(unit.sum())
    ch_stored = int((pmask & stored).sum())
    ch_cherry = int((pmask & cherry).sum())
    ch_tip = int((pmask & tip).sum())
    tips_of_cherries = int(cmask.sum())
    n_children_of_units = ch_stored + ch_cherry + ch_tip
    # bottom-up, per unit: descriptor 32; per child E, mask, S, exponent (32); stored child's vector; per tip of
    # a cherry child E, mask, S (24); writes: vector, S, exponent; S + exponent of every cherry child (16)
    bu_levels = n_units * (32 + vec + 16) + n_children_of_units * 32 + ch_stored * vec + tips_of_cherries * 24 + ch_cherry * 16
    # top-down, per unit: descriptor 32, its posterior + sum + exponent (vec + 16); per child the same gather
    # (32) + posterior, sum, exponent written (vec + 16); stored child: its bottom-up vector; tips of cherry children:
    # gather 24 + posterior, sum, exponent written
    td_levels = n_units * (32 + vec + 16) + n_children_of_units * (32 + vec + 16) + ch_stored * vec \
        + tips_of_cherries * (24 + vec + 16)
    # two-level unit, bottom-up: descriptor 32; E + mask of the two children (32), own mask (8); per child its two
    # cherries' E and mask (32) and its four tips' E, mask, S (96); writes: pi . v of the four cherries (32), pi . v +
    # exponent of the two children (32), the node's vector, pi . v, exponent (vec + 16)
    bu_two = n_sup * (32 + 32 + 8 + 2 * (32 + 96) + 32 + 32 + vec + 16)
    # top-down: one unit per child.  Shared by the two: descriptor 32, the node's row, sum, exponent (vec + 16); per
    # child: E, S, mask, exponent (32), its cherries' E, S, mask (48), its tips' E, S, mask (96); writes 7 rows with sum
    # and exponent (the child, two cherries, four tips)
    td_two = n_sup * (32 + vec + 16) + 2 * n_sup * (32 + 48 + 96 + 7 * (vec + 16))
    # stacked unit, bottom-up: descriptor 32; E, S, exponent of the four grandchildren (96) and their vectors (4 vec), E of
    # the two children and the three own masks (40); written: pi . v + exponent of the children (32), the node's vector,
    # pi . v, exponent (vec + 16)
    bu_stack = n_stack * (32 + 96 + 4 * vec + 40 + 32 + vec + 16)
    # top-down, one unit per child: descriptor 32 and the node's row (vec + 16) once per node; per child E, S, mask,
    # exponent (32), the grandchildren's E, S, mask, exponent (64) and vectors (2 vec); written: 3 rows with sums and
    # exponents (the child and its two children)
    td_stack = n_stack * (32 + vec + 16) + 2 * n_stack * (32 + 64 + 2 * vec + 3 * (vec + 16))
    bu_levels += bu_stack                               # (they run between the level launches, in the same HIP-event slots)
    td_levels += td_stack
    bu, td = bu_levels + bu_two, td_levels + td_two
    # per-branch data: dist in (8, read once per chunk of columns a thread walks: run_prep's cpy), E out (8); tips: mask
    # in (8 W), S out (8)
    cpy, bx = 1, (N + 255) // 256
    while cpy < 8 and cpy * 2 <= n_cols and bx * ((n_cols + 2 * cpy - 1) // (2 * cpy)) >= 4096:
        cpy *= 2
    prep = N * (8 + 8.0 / cpy) + int(tip.sum()) * (8 * W + 8)
    return dict(bottom_up=bu, top_down=td, prep=prep, total=bu + td + prep, vec_bytes=vec, n_stored=int(stored.sum()),
                n_cherries=int(cherry.sum()), n_tips=int(tip.sum()), n_two_level=n_sup, n_stacked=n_stack,
                bottom_up_levels=bu_levels, bottom_up_two_level=bu_two, top_down_levels=td_levels,
                top_down_two_level=td_two, rows_two_level=14 * n_sup,
                per_unit=dict(bottom_up=bu / (N * k), top_down=td / (N * k), prep=prep / (N * k)))


def library_forest(eng, flat):
    """The forest in the numbering the library works in (pml_tree_order): what the byte model of its schedules must look at
    -- which nodes form two-level or stacked units depends on which sibling groups are neighbours."""
    order = eng.node_order()
    return flat if np.array_equal(order, np.arange(flat.n_nodes)) else flat.renumbered(order)


def csrc_digest():
    """
    sha256 over the sources of the profiled kernels (the F81-family sweeps and the device helpers they are made of):
    profiles/traffic.json is only valid for the sources it was measured on -- pml_api.hip (schedules, launch geometry)
    included since round 4.
    """
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(REPO, 'pastml_amd', 'csrc')
    for name in ('pml_device.h', 'pml_kernels_f81.h', 'pml_api.hip'):
        h.update(name.encode())
        with open(os.path.join(d, name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------------------------------
def library_source_digest():
    """Digest of the library's sources as the tree holds them now (what pastml_amd/build.py compiles in)."""
    from pastml_amd import build
    return build.source_digest()


def run_cpu_baseline(args, levels, k, model):
    """The CPU baseline of this run (None if switched off): called from a process that has not touched a GPU."""
    if args.no_cpu_baseline or under_profiler():
        return None
    cl = args.cpu_baseline_levels
    if cl is None:
        cl = min(levels, 18)  # 262 144 tips: about 20 s of single-thread numpy per character on the GPU box
    return cpu_baseline(k, cl, model, args.cpu_baseline_cores)


def spawn_ranks(args):
    """
    `python bench.py --gpus N` without a launcher: start the N rank processes (this process never touches a GPU).  The CPU
    baseline is timed HERE, once, before the ranks exist -- no rank competes with it for the host cores -- and handed to
    rank 0 through a file (PASTML_BENCH_CPU_BASELINE), so the N > 1 line carries it like the N = 1 line does.
    """
    import socket
    levels, k, _ = WORKLOADS[args.workload]
    cpu = run_cpu_baseline(args, levels, k, 'JC' if args.workload == 'cfg2' else 'F81')
    cpu_path = None
    if cpu is not None:
        fd, cpu_path = tempfile.mkstemp(prefix='pastml_amd_cpu_baseline_', suffix='.json')
        with os.fdopen(fd, 'w') as f:
            json.dump(cpu, f)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    rdzv = tempfile.mkdtemp(prefix='pastml_amd_rdzv_')
    procs = []
    fd, out_path = tempfile.mkstemp(prefix='pastml_amd_rank0_', suffix='.stdout')
    with os.fdopen(fd, 'wb') as out0:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
                       MASTER_PORT=str(port), PASTML_AMD_RDZV_DIR=rdzv, HSA_ENABLE_IPC_MODE_LEGACY='0')
            if cpu_path is not None:
                env['PASTML_BENCH_CPU_BASELINE'] = cpu_path
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        # All ranks are watched: the first one to exit non-zero ends the job -- the others may be waiting for it inside a
        # collective, where nothing will ever wake them -- and so does the job's time limit.  (Fresh child processes
        # only; this parent never touches a GPU.)
        limit = float(os.environ.get('BENCH_RANKS_TIMEOUT', '1800'))
        t0 = time.time()
        codes = [None] * len(procs)
        failed = None
        while any(c is None for c in codes) and failed is None:
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.poll()
                    if codes[i] not in (None, 0):
                        failed = 'rank {} exited with code {}'.format(i, codes[i])
            if failed is None and time.time() - t0 > limit:
                failed = 'no result after {:.0f} s'.format(limit)
            if failed is None:
                time.sleep(0.05)
        if failed is not None:
            sys.stderr.write('bench.py: {}; stopping the other ranks\n'.format(failed))
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t1 = time.time()
            while any(p.poll() is None for p in procs) and time.time() - t1 < 10:
                time.sleep(0.05)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            codes = [p.wait() for p in procs]
    with open(out_path, 'rb') as f:
        sys.stdout.write(f.read().decode())
    sys.stdout.flush()
    import shutil
    shutil.rmtree(rdzv, ignore_errors=True)
    for path in (out_path, cpu_path):
        try:
            if path:
                os.remove(path)
        except OSError:
            pass
    if failed is not None:
        return max(1, max(abs(c) for c in codes if c is not None) % 256 or 1)
    return 0


def reference_fixture(flat, k):
    """
    The reference's own numbers for characters 0 and 1 of config 4 at full size (tests/golden/synthetic_cfg4_full.npz,
    written by tests/golden/make_golden.py::case_cfg4_full from the real PastML: ln L and posteriors at every 4 099th
    node), or None when this run is not that tree.
    """
    path = os.path.join(REPO, 'tests', 'golden', 'synthetic_cfg4_full.npz')
    if k != 64 or flat.n_tips != (1 << 20) or not os.path.exists(path):
        return None
    return np.load(path, allow_pickle=False)


def validate_columns(eng, flat, k, tip_states, lnl, stride=4099, chars=None):
    """
    After the timed region: every column's results on a strided node sample + all tips of the sample.  Size-independent
    properties (no oracle needed at this size): posterior rows sum to 1; log10(sum LH) - LH_SF equals the column's
    ln L / ln 10 at every sampled node (pastml/ml.py:468-483); an observed tip's posterior is the unit vector of its
    state.  A wrong column stride or a column left untouched fails here.  Columns that hold character 0 or 1 of the
    full-size workload (chars: the character of every column) are also compared with what the reference itself computed
    (reference_fixture): ln L to 1e-11, the sampled posteriors to 1e-9 relative.
    """
    from pastml_amd import hip
    N = flat.n_nodes
    first_tip = int(flat.tips[0])
    worst = dict(row_sum=0.0, total_lh=0.0)
    ref = reference_fixture(flat, k) if chars is not None and stride == 4099 else None
    ref_seen = dict(characters=[], max_rel_loglik_error=0.0, max_rel_posterior_error=0.0)
    for c in range(eng.n_cols):
        post = eng.download_strided(hip.BUF_POSTERIOR, c, 0, stride)
        lhs = eng.download_strided(hip.BUF_LH_SUM, c, 0, stride)
        lsf = eng.download_strided(hip.BUF_LH_SF, c, 0, stride)
        ids = np.arange(0, N, stride)
        rs = np.abs(post.sum(axis=1) - 1).max()
        tl = np.abs((np.log10(lhs) - lsf) / (lnl[c] / np.log(10)) - 1).max()
        worst['row_sum'] = max(worst['row_sum'], float(rs))
        worst['total_lh'] = max(worst['total_lh'], float(tl))
        if not (rs < 1e-12 and tl < 1e-11):
            raise SystemExit('validation failed in column {}: |row sum - 1| = {:.3g}, total-likelihood mismatch {:.3g}'
                             .format(c, rs, tl))
        if ref is not None and int(chars[c]) in (0, 1):
            ch = int(chars[c])
            want_l, want_p = float(ref['c{}_loglik'.format(ch)]), ref['c{}_posterior'.format(ch)]
            el = abs(lnl[c] / want_l - 1)
            nz = want_p > 0
            ep = float(np.abs(post[nz] / want_p[nz] - 1).max()) if post.shape == want_p.shape else np.inf
            if not (el < 1e-11 and ep < 1e-9 and np.array_equal(post == 0, ~nz)):
                raise SystemExit('validation failed in column {}: character {} differs from the reference run: ln L rel {:.3g}, '
                                 'posteriors rel {:.3g}'.format(c, ch, el, ep))
            ref_seen['characters'].append(ch)
            ref_seen['max_rel_loglik_error'] = max(ref_seen['max_rel_loglik_error'], float(el))
            ref_seen['max_rel_posterior_error'] = max(ref_seen['max_rel_posterior_error'], ep)
        is_tip = ids >= first_tip
        if np.array_equal(np.asarray(flat.tips), np.arange(first_tip, N)) and is_tip.any():
            want = tip_states[c][ids[is_tip] - first_tip]
            got = post[is_tip]
            if not (np.array_equal(got.argmax(axis=1), want) and np.all(got.max(axis=1) == 1.0)):
                raise SystemExit('validation failed in column {}: an observed tip lost its state'.format(c))
    out = dict(columns=eng.n_cols, node_stride=stride, max_row_sum_error=worst['row_sum'],
               max_total_likelihood_rel_error=worst['total_lh'], tips_checked=True)
    if ref is not None:
        out['against_reference_run'] = ref_seen  # (empty on ranks whose shard holds neither character)
    return out


# ---------------------------------------------------------------------------------------------------------------------
def secondary_measurements(device):
    """The other BASELINE configs, timed on the same GPU in the same process (after the main region)."""
    from pastml_amd import hip, synthetic
    out = {}

    def timed(fn, reps, eng):
        fn()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        eng.sync()
        return (time.perf_counter() - t0) / reps * 1e3

    # ---- cfg2: 65 536 tips, JC k=4, 1 character, marginal
    flat = synthetic.balanced_forest(16)
    k = 4
    with hip.Engine(flat, 1, k, device=device) as eng:
        spec = dict(kind=0, pi=np.ones(k) / k)
        eng.set_tip_states(synthetic.tip_states(flat.n_tips, k, 0))

        def cfg2():
            eng.set_models([(spec, (1.0, 0.0, 1.0))])
            eng.marginal_pass(posterior=False, lh=False)
        ms = timed(cfg2, 50, eng)
        sb = schedule_bytes(flat, k, 1)
        out['cfg2'] = dict(workload='BASELINE config 2: balanced 65 536-tip tree, JC k=4, 1 character, marginal '
                                    '(BU+TD+posteriors), hipGraph replay', ms_per_pass=ms,
                           value=flat.n_nodes * k / (ms * 1e-3), unit='node*state*char/s',
                           roofline=dict(bound='hbm (launch-latency-bound in practice: {} level launches)'
                                         .format(flat.n_bu_levels + flat.n_td_levels),
                                         model_bytes=sb['total'], achieved=sb['total'] / (ms * 1e-3) / 1e9,
                                         peak=HBM_PEAK_GBS, unit='GB/s', frac=sb['total'] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS))
    # ---- a ragged tree: random binary tree of 262 144 tips, 32 characters -- what real inputs look like next
    #      to the balanced benchmark tree; k = 64 (the headline's lane shape) and k = 4
    from pastml_amd.tree import FlatForest
    flat = FlatForest.random(262144, seed=3, max_arity=2, n_trees=1)
    ragged = {}
    for k in (64, 4):
        C = 32
        with hip.Engine(flat, C, k, device=device) as eng:
            specs = [(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(C)]
            eng.set_tip_states(np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)]))

            def ragged_pass():
                eng.set_models(specs)
                eng.marginal_pass(posterior=False, lh=False)
            ms = timed(ragged_pass, 20, eng)
            eng.set_models(specs)
            ms_bu = timed(lambda: eng.bottom_up(True), 20, eng)
            sb = schedule_bytes(library_forest(eng, flat), k, C)
            gb = sb['total'] * C / 1e9
            ragged['k{}'.format(k)] = dict(
                ms_per_pass=ms, ms_bottom_up=ms_bu, value=flat.n_nodes * k * C / (ms * 1e-3), unit='node*state*char/s',
                schedule=dict(zip(('level_schedule', 'n_two_level', 'n_stacked'), eng.schedule_info())),
                roofline=dict(bound='hbm', model_gb=gb, achieved=gb / (ms * 1e-3), peak=HBM_PEAK_GBS, unit='GB/s',
                              frac=gb / (ms * 1e-3) / HBM_PEAK_GBS,
                              bottom_up=dict(model_gb=sb['bottom_up'] * C / 1e9,
                                             achieved=sb['bottom_up'] * C / 1e9 / (ms_bu * 1e-3))))
    # (HBM counters of the k = 64 pass, profiles/r05t_ragged_tree_memory_counters.md: 25.2 GB moved for the 22.2 GB of this model
    # since the library numbers the nodes in height order -- round 4: 28.8 GB, every gathered 8-byte scalar cost a 128-byte line)
    ragged['k64']['roofline']['traffic_note'] = ('rocprofv3 FETCH_SIZE / WRITE_SIZE, round 5: 25.2 GB per pass = 1.14 x the model '
                                                 '(profiles/r05t_ragged_tree_memory_counters.md; round 4: 28.8 GB = 1.30 x)')
    out['ragged262k'] = dict(workload='random binary tree, 262 144 tips (FlatForest.random seed 3), 32 characters, F81 with '
                                      'per-character frequencies, marginal pass (model upload + BU + TD + posteriors)', **ragged)
    # ---- a forest with polytomies (round 5): 100 000 tips, at most 3 children per node, 2 trees, 16 characters -- the lane
    #      shapes follow the forest's arity (16 lanes per unit take four children on the lane-parallel path)
    flat = FlatForest.random(100000, seed=5, max_arity=3, n_trees=2)
    poly = {}
    for k in (64, 20, 4):
        C = 16
        with hip.Engine(flat, C, k, device=device) as eng:
            specs = [(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(C)]
            eng.set_tip_states(np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)]))

            def poly_pass():
                eng.set_models(specs)
                eng.marginal_pass(posterior=False, lh=False)
            ms = timed(poly_pass, 50, eng)
            poly['k{}'.format(k)] = dict(ms_per_pass=ms, value=flat.n_nodes * k * C / (ms * 1e-3), unit='node*state*char/s')
    out['polytomies100k'] = dict(workload='random forest of 2 trees, 100 000 tips, at most 3 children per node (FlatForest.random '
                                          'seed 5), {} nodes, 16 characters, F81 with per-character frequencies, marginal pass; round '
                                          '4\'s lane shapes and level lists: k = 64 1.77 ms, k = 20 1.33 ms '
                                          '(profiles/r05y_lane_shapes_and_sorted_levels.txt)'.format(flat.n_nodes), **poly)
    # ---- cfg3: 262 144 tips, JTT k=20, joint sweep (P(t) built and folded in registers on the FP64 vector units,
    #      pml_kernels_eigen_joint.h) + back-trace
    from pastml_amd.models.JTTModel import JTT_FREQUENCIES, JTT_RATE_MATRIX
    from pastml_amd.models.generator import get_diagonalisation
    flat = synthetic.balanced_forest(18)
    k = 20
    d, A, Ainv = get_diagonalisation(JTT_FREQUENCIES, JTT_RATE_MATRIX)
    spec = dict(kind=2, pi=JTT_FREQUENCIES, d=d, A=A, Ainv=Ainv)
    with hip.Engine(flat, 1, k, device=device) as eng:
        eng.set_models([(spec, (1.0, 0.0, 1.0))])
        eng.set_tip_states(synthetic.tip_states(flat.n_tips, k, 0))

        def cfg3():
            eng.joint_pass(copy_out=False)
        ms = timed(cfg3, 20, eng)
        ms_sweep = timed(lambda: eng.bottom_up(False), 20, eng)
        # the marginal pass of the same problem: no P(t) is formed, msg = A (e o (A^-1 v)) as two small GEMMs per 16 nodes
        ms_marginal = timed(lambda: eng.marginal_pass(posterior=False, lh=False), 20, eng)
        # P(t) = A diag(exp(d t)) A^-1 per branch.  The kernels execute less: an observed tip needs one
        # column of P (2 k^2 flops).  FP64 peak: 78.6 TFLOP/s on the matrix cores AND on the vector units (measured
        # equal, and the two do not run concurrently: scripts/ub/overlap.hip) -- the joint sweep runs on the vector units.
        flops = 2.0 * k ** 3 * (flat.n_nodes - 1)
        # executed: an observed tip needs one column of P (2 k^2 flops), an internal node all of it (2 k^3)
        flops_executed = 2.0 * k ** 3 * (flat.n_nodes - flat.n_tips - 1) + 2.0 * k ** 2 * flat.n_tips
        # compulsory bytes of the fused sweep: message (8 k) written + read per non-root node, arg-max row (k bytes),
        # mask word, branch length, exponent
        bytes_ = (flat.n_nodes - 1) * (2 * 8 * k + k + 8 + 8 + 8 + 8)
        out['cfg3'] = dict(workload='BASELINE config 3: balanced 262 144-tip tree, JTT k=20, 1 character, joint (Pupko) '
                                    'sweep + back-trace; P(t) built and folded in registers on the FP64 VECTOR units: '
                                    'BASELINE\'s "MFMA tile path" was measured and replaced -- FP64 matrix cores and vector '
                                    'units share one issue port at the same 12.8 FMA/clk/SIMD and do not overlap '
                                    '(profiles/r02e_fp64_mfma_valu_overlap.txt), and the 16x16 tiles pad k=20 to 32',
                           ms_per_pass=ms,
                           ms_joint_sweep=ms_sweep, ms_marginal_pass=ms_marginal,
                           value=flat.n_nodes * k / (ms * 1e-3), unit='node*state*char/s',
                           roofline=dict(bound='fp64-valu', units='v_fma_f64 (same FP64 peak as the matrix cores)',
                                         flops=flops, flops_executed=flops_executed,
                                         achieved=flops / (ms_sweep * 1e-3) / 1e12,
                                         achieved_executed=flops_executed / (ms_sweep * 1e-3) / 1e12,
                                         peak=FP64_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
                                         frac=flops / (ms_sweep * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                         frac_executed=flops_executed / (ms_sweep * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                         hbm_model_bytes=bytes_,
                                         hbm_frac=bytes_ / (ms_sweep * 1e-3) / 1e9 / HBM_PEAK_GBS))
    # ---- the P(t) batch north_star names: P(t) = A diag(exp(d t')) A^-1 materialised for every branch (reference:
    #      pastml/models/generator.py:54-65, CustomRatesModel.py:70-79), on cfg3's shape (JTT, k = 20) and on k = 32.  The
    #      kernel's own time by HIP events (profile slot 2); bytes out = 8 k ks per branch ("P(t) kernel on its
    #      own: 8 B in, 8 k^2 B out"), which a write stream cannot push beyond ~0.70 of the HBM peak
    #      (profiles/r02g_write_stream_ceiling.txt)
    flat = synthetic.balanced_forest(18)
    pij = {}
    rng = np.random.default_rng(11)
    for name, k in (('jtt_k20', 20), ('custom_k32', 32)):
        if k == 20:
            pi_, (d_, A_, Ainv_) = JTT_FREQUENCIES, get_diagonalisation(JTT_FREQUENCIES, JTT_RATE_MATRIX)
        else:
            pi_ = rng.dirichlet(np.ones(k) * 4)
            r_ = np.triu(rng.uniform(0.1, 2.0, size=(k, k)), 1)
            d_, A_, Ainv_ = get_diagonalisation(pi_, r_ + r_.T)
        spec = dict(kind=2, pi=pi_, d=d_, A=A_, Ainv=Ainv_)
        with hip.Engine(flat, 1, k, device=device) as eng:
            reps = 20
            eng.set_models([(spec, (1.0, 0.0, 1.0))])
            eng.pij_batch()
            eng.sync()
            eng.profile_enable(True)
            eng.profile_read(2, reset=True)
            for _ in range(reps):
                eng.set_models([(spec, (1.0, 0.0, 1.0))])   # (marks the batch stale: the next call recomputes it)
                eng.pij_batch()
            eng.sync()
            ms_total, launches = eng.profile_read(2, reset=True)
            eng.profile_enable(False)
            ms = ms_total / max(1, launches)
            ks = 4 * ((k + 3) // 4)
            bytes_out = flat.n_nodes * 8.0 * k * ks
            flops = 2.0 * k ** 3 * flat.n_nodes
            pij[name] = dict(k=k, branches=flat.n_nodes - 1, launches=int(launches), ms_per_batch=ms,
                             kernel='pij_eigen_mfma_kernel (v_mfma_f64_16x16x4_f64; one launch per batch)',
                             value=flat.n_nodes / (ms * 1e-3), unit='branch/s',
                             roofline=dict(bound='hbm', bytes_per_branch=8 * k * ks, model_bytes=bytes_out,
                                           achieved=bytes_out / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s',
                                           frac=bytes_out / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           mfma_tflops=flops / (ms * 1e-3) / 1e12,
                                           mfma_frac=flops / (ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS))
    out['pij_batch'] = dict(workload='P(t) of every branch of a balanced 262 144-tip tree ({} nodes) for one column, eigen '
                                     'models: pml_pij_batch, results left in HBM'.format(flat.n_nodes), **pij)
    # ---- cfg5-shaped optimiser gradient: real HIV1C tree (3 619 tips), Loc (k = 12): the 14 likelihoods of one
    #      L-BFGS-B gradient (13 free parameters + the point itself) in one batched bottom-up sweep
    nwk = os.path.join(REPO, 'tests', 'golden', 'data', 'hiv1c', 'pastml_phyml_tree.nwk')
    if os.path.exists(nwk):
        from pastml_amd.tree import read_tree, get_flat_forest
        flat = get_flat_forest([read_tree(nwk)])
        k, cols = 12, 14
        rng = np.random.default_rng(5)
        with hip.Engine(flat, cols, k, device=device) as eng:
            eng.set_tip_states(np.tile(rng.integers(0, k, size=flat.n_tips), (cols, 1)))
            pis = rng.dirichlet(np.ones(k) * 5, size=cols)

            def grad():
                eng.set_models([(dict(kind=0, pi=pis[c]), (5.5 + 1e-8 * c, 0.0, 1.0)) for c in range(cols)])
                eng.bottom_up(True)
            ms = timed(grad, 200, eng)
            sb = schedule_bytes(library_forest(eng, flat), k, cols)
            b = (sb['bottom_up'] + sb['prep']) * cols
            out['cfg5_gradient'] = dict(workload='BASELINE config 5 shape: HIV1C tree ({} tips), k=12, one L-BFGS-B '
                                                 'finite-difference gradient = {} likelihoods in one batched bottom-up '
                                                 'sweep (model upload included)'.format(flat.n_tips, cols),
                                        ms_per_gradient=ms, value=flat.n_nodes * k * cols / (ms * 1e-3),
                                        unit='node*state*char/s',
                                        roofline=dict(bound='hbm (latency-bound: the tree has {} height levels)'
                                                      .format(flat.n_bu_levels), model_bytes=b,
                                                      achieved=b / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s',
                                                      frac=b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS))
    # ---- a codon-sized eigen model: CUSTOM_RATES, k = 61, sum sweeps fused (two matrix-core GEMMs per 16
    #      nodes, constant operands in LDS) against the generic path (P(t) of every branch materialised in HBM)
    out['eigen_k61'] = eigen_k61_measurement(device)
    out['eigen_k128'] = eigen_wide_measurement(device)
    # ---- an optimisation of an eigen model with free frequencies: CUSTOM_RATES, k = 20, on the HIV1C
    #      tree -- every point of every finite-difference gradient needs its own eigendecomposition on the host
    #      (CustomRatesModel.py:62-68); host / device split, batched diagonalisation against one per point
    if os.path.exists(nwk):
        out['custom_rates_opt'] = custom_rates_optimisation(nwk)
    # ---- cfg4 with the observed tips' posteriors left implicit (PML_OPT_IMPLICIT_TIP_POSTERIORS): NOT the headline --
    #      the headline step writes the posterior row of every node; here the unit-vector rows of observed tips (half
    #      of all nodes of a binary tree) are not written by the sweep but when somebody reads the table.  Own byte model.
    levels, k, cpg = WORKLOADS['cfg4']
    flat = synthetic.balanced_forest(levels)
    with hip.Engine(flat, cpg, k, device=device) as eng:
        eng.set_option(hip.OPT_IMPLICIT_TIP_POSTERIORS, 1)
        specs = [(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(cpg)]
        tip_states = np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(cpg)])
        eng.set_tip_states(tip_states)
        last = {}

        def implicit_step():
            eng.set_models(specs)
            last['lnl'] = eng.marginal_pass(posterior=False, lh=False)[0]
        ms = timed(implicit_step, 10, eng)
        sb = schedule_bytes(flat, k, cpg)
        # every tip of this workload is observed: its posterior row (vec bytes) is not written
        model_bytes = (sb['total'] - flat.n_tips * sb['vec_bytes']) * cpg
        # the readers still see the full table: the strided sample of every column, tips included, as after the headline
        validation = validate_columns(eng, flat, k, tip_states, last['lnl'])
        out['cfg4_implicit_tip_posteriors'] = dict(
            workload='BASELINE config 4 shard with PML_OPT_IMPLICIT_TIP_POSTERIORS: as the headline step, but the unit-vector '
                     'posterior rows of the {} observed tips per character are written on request, not by the sweep '
                     '(separate entry: the headline value always includes them)'.format(flat.n_tips),
            ms_per_step=ms, value=flat.n_nodes * k * cpg / (ms * 1e-3), unit='node*state*char/s',
            roofline=dict(bound='hbm', model_bytes=model_bytes, achieved=model_bytes / (ms * 1e-3) / 1e9,
                          peak=HBM_PEAK_GBS, unit='GB/s', frac=model_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          byte_model='compulsory bytes of the headline schedule minus one posterior row per observed tip'),
            validation=validation)
    # ---- cfg5 end to end: ONE acr() over the 91 usable columns of the HIV1C annotation table (MPPA + F81, parameter
    #      optimisation of every character, 3 619 tips), against the reference's own time for the same columns
    #      (tests/golden/hiv1c_all.npz: measured when the fixtures were made, 1 CPU thread)
    meta = os.path.join(REPO, 'tests', 'golden', 'data', 'hiv1c', 'metadata_all.tab.gz')
    gold = os.path.join(REPO, 'tests', 'golden', 'hiv1c_all.npz')
    if os.path.exists(nwk) and os.path.exists(meta):
        import pandas as pd
        from pastml_amd.acr import acr
        from pastml_amd.batch import run_tasks
        from pastml_amd.tree import read_tree
        df = pd.read_csv(meta, sep='\t', index_col=0, header=0)
        df.index = df.index.map(str)
        best = None
        for _ in range(2):   # (the first call also pays for the library's first launches of these shapes)
            tree = read_tree(nwk)
            np.random.seed(239)
            t0 = time.perf_counter()
            res = acr(tree, df.copy(), prediction_method='MPPA', model='F81')
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, dict(run_tasks.last_stats), res)
        dt, st, res = best
        ref_s = ref_cols = None
        worst = None
        signed = []
        if os.path.exists(gold):
            z = np.load(gold)
            names = list(z['columns']) if 'columns' in z.files else []
            secs = [float(z[k]) for k in z.files if k.endswith('_reference_seconds')]
            ref_s, ref_cols = float(np.sum(secs)), len(secs)
            for r in res:
                if r['character'] in names:
                    key = 'c{}_loglik'.format(names.index(r['character']))
                    if key in z.files:
                        sd = (r['log_likelihood'] - float(z[key])) / abs(float(z[key]))   # signed: > 0 = ours is the better optimum
                        worst = abs(sd) if worst is None else max(worst, abs(sd))
                        signed.append((sd, str(r['character'])))
        out['cfg5_acr'] = dict(workload='BASELINE config 5: HIV1C tree (3 619 tips), all {} usable annotation columns, '
                                        'MPPA + F81 with parameter optimisation, one acr() call, 1 GPU'.format(len(res)),
                               seconds=dt, characters=len(res), groups=st.get('groups'), sweep_rounds=st.get('rounds'),
                               likelihood_evaluations=st.get('sweeps'),
                               optimise_seconds=st.get('optimise_all_groups_s'),
                               cpu_baseline=dict(kind='reference', seconds=ref_s, columns=ref_cols, cores=1,
                                                 sample='pastml.acr.acr() of the reference itself, column by column, '
                                                        'timed when tests/golden/hiv1c_all.npz was generated '
                                                        '(tests/golden/make_golden.py)'),
                               speedup_vs_reference=(ref_s / dt) if ref_s else None,
                               max_rel_loglik_difference_to_reference=worst,
                               # signed (ours - reference) / |reference|: a positive value is a better optimum than the
                               # reference's own; every column more than 1e-6 away is named
                               worst_rel_loglik_shortfall=min([d for d, _ in signed] + [0.0]) if signed else None,
                               best_rel_loglik_gain=max([d for d, _ in signed] + [0.0]) if signed else None,
                               columns_beyond_1e6={c: d for d, c in signed if abs(d) > 1e-6})
    return out


def eigen_k61_measurement(device, levels=16, k=61, C=4, compare=1):
    """Marginal pass (and joint sweep) of a 61-state eigen model on a balanced tree, fused against materialised P(t)."""
    from pastml_amd import hip, synthetic
    from pastml_amd.models._eigen import get_diagonalisation
    flat = synthetic.balanced_forest(levels)
    rng = np.random.default_rng(61)
    rates = np.triu(rng.uniform(0.05, 3.0, size=(k, k)), 1)
    rates = rates + rates.T
    specs = []
    for c in range(C):
        pi = rng.dirichlet(np.ones(k) * 4)
        d, a, ainv = get_diagonalisation(pi, rates)
        specs.append((dict(kind=2, pi=pi, d=d, A=a, Ainv=ainv), (1.0, 0.0, 1.0)))
    tips = np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)])
    ks = k + (k & 1)
    res = {}
    # (materialised: round 5's path for every k > 32 -- P(t) of every branch built by the one-thread-per-entry kernel, NO_PIJ_WIDE;
    # materialised_wide: the same sweeps behind round 6's matrix-core P(t) batch, what 65 <= k <= 256 run today)
    for label, tune in (('fused', {}), ('materialised', dict(NO_EIGEN_GEMM=1, NO_EIGEN_JOINT_VALU=1, NO_PIJ_WIDE=1)),
                        ('materialised_wide', dict(NO_EIGEN_GEMM=1, NO_EIGEN_JOINT_VALU=1)))[:3 if compare else 1]:
        with hip.Engine(flat, C, k, device=device, tune=tune) as eng:
            eng.set_tip_states(tips)

            def marginal():
                eng.set_models(specs)
                return eng.marginal_pass(posterior=False, lh=False)[0]

            def joint():
                eng.set_models(specs)
                return eng.bottom_up(False)
            def bottom_up():
                eng.set_models(specs)
                return eng.bottom_up(True)
            out = {}
            for name, fn, reps in (('marginal', marginal, 10), ('bottom_up', bottom_up, 10), ('joint', joint, 5))[:3 if compare else 2]:
                lnl = fn()
                eng.sync()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                eng.sync()
                out['ms_' + name] = (time.perf_counter() - t0) / reps * 1e3
                out['lnl_' + name] = [float(v) for v in lnl]
            res[label] = out
    f = res['fused']
    g = res.get('materialised', f)
    f.setdefault('ms_joint', None)
    rel = max(abs(a - b) / abs(b) for a, b in zip(f['lnl_marginal'], g['lnl_marginal']))
    # algorithmic bytes of the fused marginal pass per node and column: bottom-up writes the vector and the message and
    # reads every message once (3 vectors), top-down reads the parent's TD and BU vectors, the node's own vector and
    # message and writes TD and posterior (6 vectors); flops: two k x k products per sweep, padded to the 16-row tiles
    vec = 8.0 * ks
    bytes_pass = 9 * vec * flat.n_nodes * C
    flops = 2 * 2 * 2.0 * 64 * 64 * flat.n_nodes * C
    ms = f['ms_marginal']
    return dict(workload='CUSTOM_RATES-shaped eigen model, k = {} states, balanced {}-tip tree ({} nodes), {} characters: '
                         'marginal pass (BU + TD + posteriors) and joint sweep; fused = sum sweeps as P v = A (e o (A^-1 v)) on '
                         'the FP64 matrix cores, operands in LDS (pml_kernels_eigen_gemm.h), joint sweep with P(t) built and folded in '
                         'registers on the vector units (pml_kernels_eigen_joint.h); materialised = P(t) of every '
                         'branch in HBM ({:.1f} GB), the path of every k > 32 before round 6'
                         .format(k, flat.n_tips, flat.n_nodes, C, flat.n_nodes * C * k * ks * 8 / 1e9),
                ms_marginal_pass=ms, ms_bottom_up_sweep=f['ms_bottom_up'], ms_marginal_pass_materialised=g['ms_marginal'],
                ms_marginal_pass_materialised_wide_pij=res.get('materialised_wide', {}).get('ms_marginal'),
                ms_joint_sweep_materialised_wide_pij=res.get('materialised_wide', {}).get('ms_joint'),
                speedup_marginal=g['ms_marginal'] / ms,
                speedup_joint=(g['ms_joint'] / f['ms_joint']) if f.get('ms_joint') and g.get('ms_joint') else None,
                ms_joint_sweep=f['ms_joint'], ms_joint_sweep_materialised=g.get('ms_joint'),
                max_rel_loglik_difference=rel, loglik_fused=f['lnl_marginal'], loglik_materialised=g['lnl_marginal'],
                value=flat.n_nodes * k * C / (ms * 1e-3), unit='node*state*char/s',
                roofline=dict(bound='hbm', model_bytes=bytes_pass, achieved=bytes_pass / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS,
                              unit='GB/s', frac=bytes_pass / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              mfma_tflops=flops / (ms * 1e-3) / 1e12,
                              mfma_frac=flops / (ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                              byte_model='9 vectors of {} B per node and column (BU: vector + message written, message read; '
                                         'TD: parent TD + BU, own BU + message read, TD + posterior written)'.format(int(vec))))


def eigen_wide_measurement(device, levels=14, k=128, C=4):
    """An eigen model beyond 64 states: the marginal pass with the sum sweeps fused (two matrix-core GEMMs per 16 nodes, ONE matrix
    in LDS: a reversible model's A^-1 is A transposed and rescaled), on P(t) of every branch in HBM built by round 6's matrix-core
    batch (pij_eigen_wide_kernel), and on P(t) built by the one-thread-per-entry kernel of rounds 1 - 5 (once)."""
    from pastml_amd import hip, synthetic
    from pastml_amd.models._eigen import get_diagonalisation
    flat = synthetic.balanced_forest(levels)
    rng = np.random.default_rng(k)
    rates = np.triu(rng.uniform(0.05, 3.0, size=(k, k)), 1)
    rates = rates + rates.T
    specs = []
    for c in range(C):
        pi = rng.dirichlet(np.ones(k) * 4)
        d, a, ainv = get_diagonalisation(pi, rates)
        specs.append((dict(kind=2, pi=pi, d=d, A=a, Ainv=ainv), (1.0, 0.0, 1.0)))
    tips = np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)])
    res = {}
    for label, tune, reps in (('fused', {}, 5), ('pij_wide', dict(NO_EIGEN_GEMM=1), 5),
                              ('round5', dict(NO_EIGEN_GEMM=1, NO_PIJ_WIDE=1), 1)):
        with hip.Engine(flat, C, k, device=device, tune=tune) as eng:
            eng.set_tip_states(tips)

            def batch():
                eng.set_models(specs)
                eng.pij_batch(copy_out=False)

            def marginal():
                eng.set_models(specs)
                return eng.marginal_pass(posterior=False, lh=False)[0]
            out = {}
            for name, fn in (('pij', batch), ('marginal', marginal)):
                lnl = fn()
                eng.sync()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                eng.sync()
                out['ms_' + name] = (time.perf_counter() - t0) / reps * 1e3
            out['lnl'] = [float(v) for v in lnl]
            res[label] = out
    f, w, g = res['fused'], res['pij_wide'], res['round5']
    flops = 2.0 * k ** 3 * flat.n_nodes * C
    ks = k + (k & 1)
    return dict(workload='CUSTOM_RATES-shaped eigen model, k = {} states, balanced {}-tip tree ({} branches), {} characters, model upload '
                         'included.  marginal pass: sum sweeps fused (pml_kernels_eigen_gemm.h, one matrix in LDS) / on P(t) of every '
                         'branch in HBM ({:.1f} GB) built on the FP64 matrix cores (pml_kernels_pij_wide.h) / the same built by the '
                         'per-entry kernel of rounds 1 - 5'.format(k, flat.n_tips, flat.n_nodes, C, flat.n_nodes * C * k * ks * 8 / 1e9),
                ms_marginal_pass=f['ms_marginal'], ms_marginal_pass_pij_in_hbm=w['ms_marginal'], ms_marginal_pass_round5=g['ms_marginal'],
                ms_pij_batch=w['ms_pij'], ms_pij_batch_round5=g['ms_pij'], speedup_pij_batch=g['ms_pij'] / w['ms_pij'],
                speedup_marginal_vs_round5=g['ms_marginal'] / f['ms_marginal'],
                speedup_marginal_vs_pij_in_hbm=w['ms_marginal'] / f['ms_marginal'],
                max_rel_loglik_difference=max(max(abs(a - b) / abs(b) for a, b in zip(x['lnl'], g['lnl'])) for x in (f, w)),
                value=flat.n_nodes * k * C / (f['ms_marginal'] * 1e-3), unit='node*state*char/s',
                roofline_pij_batch=dict(bound='mfma', achieved=flops / (w['ms_pij'] * 1e-3) / 1e12, peak=FP64_MFMA_PEAK_TFLOPS,
                                        unit='TFLOP/s', frac=flops / (w['ms_pij'] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                        flop_model='2 k^3 per branch and character'))


def custom_rates_optimisation(nwk, k=20):
    """acr() of one 20-state character under CUSTOM_RATES with free frequencies: seconds, and where they go."""
    import tempfile
    from pastml_amd.acr import acr
    from pastml_amd.models import _eigen
    from pastml_amd.tree import read_tree
    rng = np.random.default_rng(11)
    states = np.array(['s{:02d}'.format(i) for i in range(k)])
    rates = np.triu(rng.uniform(0.2, 3.0, size=(k, k)), 1)
    rates = rates + rates.T
    weights = rng.dirichlet(np.ones(k) * 2)
    fd, rate_file = tempfile.mkstemp(prefix='pastml_amd_rates_', suffix='.txt')
    os.close(fd)
    _eigen.save_matrix(states, rates, rate_file)
    results = {}
    try:
        for label, env in (('per_point', '0'), ('batched', '1')):
            os.environ['PASTML_AMD_EIG_BATCH'] = env
            best = None
            for _ in range(2):
                tree = read_tree(nwk)
                draw = np.random.default_rng(12)
                for tip in tree:
                    tip.add_feature('cr', {states[draw.choice(k, p=weights)]})
                host = dict(s=0.0, calls=0, points=0)
                plain = _eigen.CustomRatesModel.kernel_points

                def timed_points(self, vectors, _plain=plain, _host=host):
                    t0 = time.perf_counter()
                    try:
                        return _plain(self, vectors)
                    finally:
                        _host['s'] += time.perf_counter() - t0
                        _host['calls'] += 1
                        _host['points'] += len(vectors)
                _eigen.CustomRatesModel.kernel_points = timed_points
                try:
                    np.random.seed(239)
                    t0 = time.perf_counter()
                    res = acr(tree, columns=['cr'], column2states={'cr': states}, prediction_method='MPPA', model='CUSTOM_RATES',
                              column2rates={'cr': rate_file})[0]
                    dt = time.perf_counter() - t0
                finally:
                    _eigen.CustomRatesModel.kernel_points = plain
                if best is None or dt < best['seconds']:
                    best = dict(seconds=dt, host_points_seconds=host['s'], point_batches=host['calls'], points=host['points'],
                                ms_per_batch=host['s'] / max(1, host['calls']) * 1e3, log_likelihood=float(res['log_likelihood']))
            results[label] = best
    finally:
        os.environ.pop('PASTML_AMD_EIG_BATCH', None)
        os.unlink(rate_file)
    a, b = results['per_point'], results['batched']
    return dict(workload='CUSTOM_RATES, k={} with free frequencies ({} parameters), one character on the HIV1C tree (3 619 '
                         'tips), MPPA: one acr() call; host = decoding the optimiser\'s points incl. the '
                         'eigendecompositions (numpy LAPACK), the rest = device sweeps + L-BFGS-B'.format(k, k + 0),
                seconds=b['seconds'], host_points_seconds=b['host_points_seconds'], ms_per_point_batch=b['ms_per_batch'],
                point_batches=b['point_batches'], points=b['points'], log_likelihood=b['log_likelihood'],
                per_point=a, speedup_host_points=a['host_points_seconds'] / b['host_points_seconds'],
                speedup_total=a['seconds'] / b['seconds'], same_optimum=a['log_likelihood'] == b['log_likelihood'],
                note='batched = the distinct frequency vectors of a batch of points diagonalised by ONE stacked '
                     'numpy.linalg.eig / inv call (same LAPACK per matrix, same bits) and no re-diagonalisation of an '
                     'unchanged vector; LAPACK itself (dgeev + dgesv, ~85 us per 20 x 20 matrix) is the floor: threads do not '
                     'help (OpenBLAS serialises its small calls, profiles/r06c_eigen_host.txt)')


# ---------------------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if 'RANK' not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit('--gpus {} but WORLD_SIZE={}'.format(args.gpus, world))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

    levels, k, cpg = WORKLOADS[args.workload]
    if args.chars_per_gpu:
        cpg = args.chars_per_gpu
    model = 'JC' if args.workload == 'cfg2' else 'F81'
    profiled = under_profiler()

    # CPU baseline first, while this process has not touched the GPU yet (its worker processes are started from a
    # process without a GPU context); never under a profiler, whose preloaded library has already initialised the GPU
    # N > 1: bench.py's own parent has timed it already (spawn_ranks) and says where; under a launcher of somebody else's
    # (torch.distributed.run) rank 0 times it now, before it touches its GPU, while the other ranks build their engines and
    # then wait for it at the communicator's rendezvous (whose time limit is raised for that).
    cpu = None
    handed = os.environ.get('PASTML_BENCH_CPU_BASELINE')
    if rank == 0 and handed and os.path.exists(handed):
        with open(handed) as f:
            cpu = json.load(f)
    elif rank == 0:
        cpu = run_cpu_baseline(args, levels, k, model)
    if world > 1:
        os.environ.setdefault('PASTML_AMD_RDZV_TIMEOUT', '900')

    from pastml_amd import hip, synthetic, sharding

    all_on_gpu0 = os.environ.get('BENCH_ALL_RANKS_ON_GPU0') is not None   # dry runs of N > 1 on a one-GPU box (gloo)
    gpu_index = 0 if all_on_gpu0 else local_rank
    flat = synthetic.balanced_forest(levels)
    N = flat.n_nodes
    chars = list(sharding.shard_characters(cpg * world, rank, world))  # contiguous block of characters per rank

    eng = hip.Engine(flat, cpg, k, device=gpu_index)
    if model == 'JC':
        specs = [dict(kind=0, pi=np.ones(k) / k) for _ in chars]
    else:
        specs = [dict(kind=0, pi=synthetic.f81_frequencies(k, c)) for c in chars]
    eng.set_models([(s, (1.0, 0.0, 1.0)) for s in specs])
    tip_states = np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in chars])
    eng.set_tip_states(tip_states)
    eng.sync()
    if os.environ.get('BENCH_TEST_DIE_RANK') == str(rank):   # (tests: a rank that dies before it joins the communicator)
        os._exit(7)
    # the communicator lives on the engine's own context: the all-reduce is stream-ordered behind the sweep
    comm = sharding.init(device=gpu_index, engine=eng)

    def step():
        # model parameters are re-sent every step, as an optimiser iteration would: forces the per-branch
        # transition data to be recomputed inside the step
        eng.set_models([(s, (1.0, 0.0, 1.0)) for s in specs])
        # bottom-up sweep (ln L per character back to the host) + top-down sweep, marginals and posteriors (outputs stay
        # in HBM): one call, one host round trip (pml_marginal_pass)
        lnl = eng.marginal_pass(posterior=False, lh=False)[0]
        # the one collective of the path: summed log-likelihood over the ranks (RCCL all-reduce of 8 bytes)
        total = comm.allreduce_loglik(lnl)
        return total, lnl

    def fence():
        eng.sync()
        hip.device_sync(gpu_index)
        comm.barrier()
        hip.device_sync(gpu_index)

    for _ in range(args.warmup):
        step()
    eng.profile_enable(not args.no_kernel_timing)
    for w in range(5):
        eng.profile_read(w, reset=True)
    fence()
    t0 = time.perf_counter()
    total = lnl = None
    for _ in range(args.steps):
        total, lnl = step()
    fence()
    dt_own = time.perf_counter() - t0
    dt = float(comm.allreduce([dt_own], op='max')[0])
    # who ran where (every N > 1 line says it: the first run on a real multi-GPU node must be readable from its output):
    # per-rank step time and device UUID -- two ranks with one UUID share a GPU --, and what the library's communicator is
    per_rank = None
    comm_report = dict(collective=comm.name)
    if world > 1:
        slots = np.zeros((world, 17))
        slots[rank, 0] = dt_own / args.steps * 1e3
        slots[rank, 1:] = list(bytes.fromhex(hip.device_uuid(gpu_index)))
        slots = comm.allreduce(slots.ravel()).reshape(world, 17)
        per_rank = [dict(rank=r, ms_per_step=float(slots[r, 0]), device_uuid=bytes(int(b) for b in slots[r, 1:]).hex())
                    for r in range(world)]
        comm_report['distinct_devices'] = len({p['device_uuid'] for p in per_rank})
    if comm.name == 'rccl':
        info = eng.comm_info()
        comm_report.update(rccl_ranks=info['rccl_ranks'], backend=info['backend'], world=info['world'])

    # kernel time by HIP events on the library's stream: the level kernels' launches of the two sweeps (0, 1), the
    # per-branch pass (2), the launches of the two-level units (3 top-down, 4 bottom-up)
    bul_ms, bul_launches = eng.profile_read(0)
    tdl_ms, tdl_launches = eng.profile_read(1)
    prep_ms, prep_launches = eng.profile_read(2)
    td2_ms, td2_launches = eng.profile_read(3)
    bu2_ms, bu2_launches = eng.profile_read(4)
    if world > 1:   # the roofline of an N > 1 line is the slowest rank's: every rank runs the same launches on its own shard
        bul_ms, tdl_ms, prep_ms, td2_ms, bu2_ms = (float(v) for v in
                                                   comm.allreduce([bul_ms, tdl_ms, prep_ms, td2_ms, bu2_ms], op='max'))
    bu_ms, bu_launches = bul_ms + bu2_ms, bul_launches + bu2_launches
    td_ms, td_launches = tdl_ms + td2_ms, tdl_launches + td2_launches
    eng.profile_enable(False)
    held, free = eng.memory()
    # A check that fails on one rank must not leave the others waiting in the closing collective: failures are collected,
    # agreed on by all ranks (all-reduce of a flag), and every rank leaves through the same barrier before it exits.
    failure = None
    validation = None
    if not args.no_validate:
        try:
            validation = validate_columns(eng, flat, k, tip_states, lnl, chars=chars)
        except SystemExit as e:
            failure = str(e)

    if rank == 0:
        units_per_step = N * k * cpg * world
        value = units_per_step * args.steps / dt
        # dominant kernel: the top-down + marginals level kernel (td_f81_kernel).  achieved = compulsory bytes of this
        # schedule (schedule_bytes: DESIGN.md 4b) / the kernel's own time
        sb = schedule_bytes(flat, k, cpg)
        per_step = {w: sb[w] * cpg for w in ('bottom_up', 'top_down', 'prep')}
        def rate(b, ms):
            return b * args.steps / (ms * 1e-3) / 1e9 if ms > 0 else None
        td_gbs, bu_gbs, prep_gbs = rate(per_step['top_down'], td_ms), rate(per_step['bottom_up'], bu_ms), \
            rate(per_step['prep'], prep_ms)
        # The dominant kernel.  With two-level units (the balanced tree of this workload) it is td_f81_super_kernel: ONE
        # launch per step that finishes the children of every two-level node, their cherries and their tips -- 7 of every
        # 8 posterior rows of the table.  Without them: the level kernel td_f81_kernel, averaged over its launches.
        two_level = td2_launches > 0
        if two_level:
            dom = dict(name='td_f81_super_kernel (top-down + marginals + posteriors below the two-level nodes: their '
                            'children, those children\'s cherries and tips -- one launch per step)',
                       key='td_two_level_bytes_per_step', model=sb['top_down_two_level'] * cpg, ms=td2_ms,
                       launches=td2_launches)
        else:
            dom = dict(name='td_f81_kernel (top-down + marginals + posteriors, one launch per depth level)',
                       key='td_bytes_per_step', model=per_step['top_down'], ms=td_ms, launches=td_launches)
        dom_gbs = rate(dom['model'], dom['ms'])
        # HBM bytes per launch from the committed PMC passes: only if they were measured on these kernel sources and
        # this workload shape; bench asserts that counters and model agree
        traffic = traffic_note = traffic_stamp = None
        tpath = os.path.join(REPO, 'profiles', 'traffic.json')
        if os.path.exists(tpath) and dom['launches'] > 0:
            try:
                tj = json.load(open(tpath))
                entry = tj.get(args.workload, {})
                traffic_stamp = entry.get('csrc_sha')
                if entry.get('csrc_sha') != csrc_digest():
                    traffic_note = 'profiles/traffic.json was measured on other kernel sources ({} != {})'.format(
                        entry.get('csrc_sha'), csrc_digest())
                elif entry.get('chars_per_gpu') != cpg:
                    traffic_note = 'profiles/traffic.json is for {} characters per GPU'.format(entry.get('chars_per_gpu'))
                else:
                    traffic = entry[dom['key']] / (dom['launches'] / args.steps)
                    ratio = entry[dom['key']] / dom['model']
                    traffic_note = 'PMC bytes / model bytes = {:.3f}'.format(ratio)
                    if not 0.85 <= ratio <= 1.15:
                        failure = failure or ('HBM counters ({:.3g} B per step) and the byte model ({:.3g}) of the '
                                              'dominant kernel disagree by more than 15 %'
                                              .format(entry[dom['key']], dom['model']))
            except (OSError, ValueError, KeyError, TypeError) as e:
                traffic_note = 'profiles/traffic.json unreadable: {}'.format(e)
        avg_launch_s = dom['ms'] / max(1, dom['launches']) * 1e-3
        # Figures that do not depend on this schedule's own byte model:
        #   output floor  every posterior entry written once, 8 B per node*state*char -- what ANY schedule must move
        #   SURVEY 8(d)   the reference schedule's 48 B per unit (every vector of every node written and read back); this
        #                 schedule keeps cherries / two-level children in registers and never stores TD vectors, so the
        #                 ratio comes out ABOVE the peak: the 48-byte model is not a bound for it
        #   achievable    the guide's ~6.3 TB/s of attainable HBM bandwidth next to the 8 TB/s peak
        units_dom = (float(sb['rows_two_level']) if two_level else float(N)) * k * cpg   # posterior entries the dominant kernel writes
        def independent(units_per_pass, ms_total, n_pass):
            if not ms_total or ms_total <= 0:
                return {}
            per_s = units_per_pass * n_pass / (ms_total * 1e-3)
            return {'output_floor_bytes': OUTPUT_FLOOR_BYTES_PER_UNIT * units_per_pass,
                    'output_floor_gbs': OUTPUT_FLOOR_BYTES_PER_UNIT * per_s / 1e9,
                    'frac_of_output_floor': OUTPUT_FLOOR_BYTES_PER_UNIT * per_s / 1e9 / HBM_PEAK_GBS,
                    'frac_of_output_floor_achievable': OUTPUT_FLOOR_BYTES_PER_UNIT * per_s / 1e9 / HBM_ACHIEVABLE_GBS,
                    'achievable_peak': HBM_ACHIEVABLE_GBS}
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
                       'sharding': 'characters over ranks, no data-path collective; 1 all-reduce (8 B) per step',
                       'collective': comm.name, **{k_: v_ for k_, v_ in comm_report.items() if k_ != 'collective'}},
            'per_rank': per_rank,
            'loglik_sum': total,
            'roofline': {
                'bound': 'hbm', 'kernel': dom['name'],
                'achieved': dom_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': (dom_gbs / HBM_PEAK_GBS) if dom_gbs else None,
                'traffic': traffic, 'traffic_note': traffic_note,
                'traffic_gbs': (traffic / avg_launch_s / 1e9) if traffic and dom['ms'] > 0 else None,
                'model_bytes_per_launch': dom['model'] / max(1, dom['launches'] / args.steps),
                'model_bytes_per_unit': dom['model'] / cpg / (N * k),
                'avg_launch_ms': dom['ms'] / max(1, dom['launches']),
                'launches': dom['launches'],
                'byte_model': 'compulsory bytes of this schedule (DESIGN.md 4b): posterior written for every node, '
                              'parent posterior + stored child vector read for stored (non-cherry) internal nodes only, '
                              'per-node scalars; tips are 8-byte masks, cherries and the children of two-level nodes live '
                              'in registers, TD vectors are never stored',
                'reference_schedule_bytes_avoided': (REFERENCE_SCHEDULE_BYTES['top_down'] * N * k * cpg
                                                     - per_step['top_down']),
                'frac_of_achievable': (dom_gbs / HBM_ACHIEVABLE_GBS) if dom_gbs else None,
                'output_rows_per_launch': units_dom / k / cpg,
                **independent(units_dom, dom['ms'], dom['launches']),
            },
            'roofline_top_down': {
                'kernel': 'whole top-down sweep: td_f81_kernel (one launch per depth level)'
                          + (' + td_f81_super_kernel (one launch)' if two_level else ''),
                'achieved': td_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': (td_gbs / HBM_PEAK_GBS) if td_gbs else None, 'launches': td_launches,
                'model_bytes_per_unit': sb['per_unit']['top_down'],
                'level_kernel_ms_per_step': tdl_ms / args.steps, 'two_level_ms_per_step': td2_ms / args.steps,
            },
            'roofline_bottom_up': {
                'kernel': 'whole bottom-up sweep: bu_f81_kernel (one launch per height level)'
                          + (' + bu_f81_super_kernel (one launch)' if two_level else ''),
                'achieved': bu_gbs,
                'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': (bu_gbs / HBM_PEAK_GBS) if bu_gbs else None,
                'avg_launch_ms': bu_ms / max(1, bu_launches), 'launches': bu_launches,
                'model_bytes_per_unit': sb['per_unit']['bottom_up'],
                'level_kernel_ms_per_step': bul_ms / args.steps, 'two_level_ms_per_step': bu2_ms / args.steps,
                'two_level_gbs': rate(sb['bottom_up_two_level'] * cpg, bu2_ms),
                'reference_schedule_bytes_avoided': (REFERENCE_SCHEDULE_BYTES['bottom_up'] * N * k * cpg
                                                     - per_step['bottom_up']),
            },
            'roofline_step': {
                'what': 'whole step: all kernels, launch gaps and host round trips of one marginal pass',
                'model_bytes': sum(per_step.values()),
                'achieved': sum(per_step.values()) * args.steps / dt / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': sum(per_step.values()) * args.steps / dt / 1e9 / HBM_PEAK_GBS,
                'frac_of_achievable': sum(per_step.values()) * args.steps / dt / 1e9 / HBM_ACHIEVABLE_GBS,
                **independent(float(N) * k * cpg, dt * 1e3, args.steps),
                'survey_8d_bytes': SURVEY_8D_BYTES_PER_UNIT * N * k * cpg,
                'survey_8d_gbs': SURVEY_8D_BYTES_PER_UNIT * N * k * cpg * args.steps / dt / 1e9,
                'survey_8d_ratio': SURVEY_8D_BYTES_PER_UNIT * N * k * cpg * args.steps / dt / 1e9 / HBM_PEAK_GBS,
                'survey_8d_note': 'SURVEY 8(d) counts 48 B per node*state*char (the reference schedule: every BU / TD / '
                                  'posterior vector of every node written and read back); a ratio above 1 says this '
                                  'schedule does not move that traffic -- that model is not a bound for it, the output floor is',
            },
            'kernel_ms_per_step': {'bottom_up': bu_ms / args.steps, 'top_down': td_ms / args.steps,
                                   'prep': prep_ms / args.steps},
            'prep_gbs': prep_gbs,
            'device_memory_gb': held / 1e9,
            'validation': validation,
            # which sources the library that ran was compiled from (pml_build_digest) next to what the tree holds now and
            # the stamp of the committed HBM counters
            'library': {'build_digest': hip.build_digest(), 'source_digest': library_source_digest(),
                        'traffic_stamp': traffic_stamp, 'kernel_sources_digest': csrc_digest()},
        }
        if cpu is not None:
            out['cpu_baseline'] = cpu
            out['speedup_vs_cpu_baseline'] = value / cpu['value']
            out['speedup_vs_cpu_baseline_single_core'] = value / cpu['single_core_value']
        elif profiled and not args.no_cpu_baseline:
            out['cpu_baseline'] = None
            out['cpu_baseline_note'] = 'skipped: a profiler is attached (no worker processes from a GPU-initialised process)'
    any_failure = float(comm.allreduce([1.0 if failure else 0.0], op='max')[0]) > 0
    comm.barrier()
    sharding.shutdown()
    eng.close()
    if any_failure:
        if failure:
            sys.stderr.write('bench.py, rank {}: {}\n'.format(rank, failure))
        sys.exit(3)
    if rank == 0:
        if world == 1 and not args.no_secondary and args.workload == 'cfg4':
            try:
                out['secondary'] = secondary_measurements(gpu_index)
            except Exception as e:  # the headline line must survive a failure of the extras
                out['secondary'] = {'error': repr(e)}
        print(json.dumps(out))


if __name__ == '__main__':
    main()
