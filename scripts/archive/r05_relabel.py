"""
Premise test for the height-ordered layout: the library accepts ANY breadth-first numbering -- the order
of the sibling groups inside a depth is free -- so the layout can be tried without touching a kernel: number the nodes so
that, inside every depth, the sibling groups lie in the order in which the sweeps' units read them (groups whose parents
share a fused height next to each other, in parent order; the tips of cherries by the height of the node that rebuilds the
cherry), and time the existing kernels on the relabelled forest.  Same tree, same arithmetic per node, same ln L bits.
usage: r05_relabel.py [n_tips] [k ...]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pastml_amd import hip, synthetic
from pastml_amd.tree import FlatForest


def fused_heights(flat):
    N = flat.n_nodes
    nc, fc, parent, depth = flat.n_children, flat.first_child, flat.parent, flat.depth
    internal = nc > 0
    n_tip_children = np.zeros(N, dtype=np.int64)
    np.add.at(n_tip_children, parent[(parent >= 0) & ~internal], 1)
    cherry = internal & (n_tip_children == nc) & (parent >= 0)
    stored = internal & ~cherry
    fh = np.zeros(N, dtype=np.int64)
    for d in range(int(depth.max()), -1, -1):
        idx = np.flatnonzero(stored & (depth == d))
        if not len(idx):
            continue
        fh[idx] += 1
        p = parent[idx]
        ok = p >= 0
        np.maximum.at(fh, p[ok], fh[idx][ok])
    return fh, stored, cherry


def relabel(flat, mode='height'):
    """new_of_old, old_of_new for the numbering described above (mode 'height'), or the identity ('bfs')."""
    N = flat.n_nodes
    nc, fc, parent = flat.n_children.astype(np.int64), flat.first_child.astype(np.int64), flat.parent.astype(np.int64)
    fh, stored, cherry = fused_heights(flat)
    # the unit that gathers a node's children: the node itself if it is stored, its parent if it is a cherry
    cls = np.where(stored, fh, 0)
    cls[cherry] = fh[parent[cherry]]
    if mode == 'height+shape':
        # the descriptor's shape word of the unit that does the gathering (pml_tree_upload, describe): level lists of wide
        # units are sorted by it
        tip = nc == 0
        code = np.where(tip, 0, np.where(cherry, np.where(nc > 4, 2, 1 + nc), 1))
        packed = np.minimum(nc, 15)
        ok = np.ones(N, dtype=bool)
        first2 = np.ones(N, dtype=bool)
        for j in range(4):
            has = nc > j
            ch = np.where(has, fc + j, 0)
            cj = np.where(has, code[ch], 0)
            packed = packed | (cj << (8 + 3 * j))
            ok &= ~(has & cherry[ch] & (nc[ch] > 4))
            if j >= 2:
                first2 &= ~(has & (cj == 1))
        packed = packed | (ok.astype(np.int64) << 4) | (first2.astype(np.int64) << 5)
        shape = np.where(stored, packed, 0)
        shape[cherry] = packed[parent[cherry]]
        cls = cls * (1 << 24) + shape
    level = list(range(len(flat.roots)))
    old_of_new = list(level)
    cur = np.array(level, dtype=np.int64)
    while len(cur):
        par = cur[nc[cur] > 0]
        if not len(par):
            break
        if mode != 'bfs':
            par = par[np.argsort(cls[par], kind='stable')]   # (stable: the parents' own order inside a class)
        counts = nc[par]
        starts = np.repeat(fc[par], counts)
        within = np.arange(int(counts.sum())) - np.repeat(np.cumsum(counts) - counts, counts)
        nxt = starts + within
        old_of_new.extend(nxt.tolist())
        cur = nxt
    old_of_new = np.array(old_of_new, dtype=np.int64)
    assert len(old_of_new) == N and len(np.unique(old_of_new)) == N
    new_of_old = np.empty(N, dtype=np.int64)
    new_of_old[old_of_new] = np.arange(N)
    return new_of_old, old_of_new


def relabelled_forest(flat, mode='height'):
    new_of_old, old_of_new = relabel(flat, mode)
    parent = np.where(flat.parent[old_of_new] >= 0, new_of_old[np.maximum(flat.parent[old_of_new], 0)], -1)
    n_children = flat.n_children[old_of_new]
    first_child = np.where(n_children > 0, new_of_old[np.minimum(flat.first_child[old_of_new], flat.n_nodes - 1)], 0)
    out = FlatForest(parent, n_children, first_child, flat.dist[old_of_new], np.arange(len(flat.roots)))
    return out, new_of_old, old_of_new


def main():
    ks = [int(a) for a in sys.argv[2:]] or [64, 4]
    if len(sys.argv) > 1 and sys.argv[1] == 'hiv1c':
        from pastml_amd.tree import read_tree, get_flat_forest
        repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        flat = get_flat_forest([read_tree(os.path.join(repo, 'tests', 'golden', 'data', 'hiv1c', 'pastml_phyml_tree.nwk'))])
    else:
        n_tips = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
        flat = FlatForest.random(n_tips, seed=3, max_arity=int(os.environ.get('R05_ARITY', 2)), n_trees=1)
    print('forest: %d nodes, %d tips, arity <= %d' % (flat.n_nodes, flat.n_tips, int(flat.n_children.max())))
    variants = [('bfs', flat, None, None)]
    for mode in ('height', 'height+shape'):
        f2, n_o, o_n = relabelled_forest(flat, mode)
        assert np.array_equal(f2.post_rank[n_o], flat.post_rank)
        variants.append((mode, f2, o_n, n_o))
    C = int(os.environ.get('R05_COLS', 32))
    for k in ks:
        res = {}
        for name, f, tip_perm, new_of_old in variants:
            with hip.Engine(f, C, k) as eng:
                specs = [(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(C)]
                # the same states on the same tips: tip j of the relabelled forest is old tip old_of_new[f.tips[j]]
                states = np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)])
                if tip_perm is not None:
                    old_tip_index = np.full(flat.n_nodes, -1, dtype=np.int64)
                    old_tip_index[flat.tips] = np.arange(flat.n_tips)
                    states = states[:, old_tip_index[tip_perm[f.tips]]]
                eng.set_tip_states(states)

                def one_pass():
                    eng.set_models(specs)
                    return eng.marginal_pass(posterior=False, lh=False)[0]
                lnl = one_pass()
                eng.sync()
                t0 = time.perf_counter()
                for _ in range(20):
                    one_pass()
                eng.sync()
                ms = (time.perf_counter() - t0) / 20 * 1e3
                post = eng.download_strided(hip.BUF_POSTERIOR, 3, 0, 1)
                eng.set_models(specs)
                eng.bottom_up(True)
                eng.sync()
                t0 = time.perf_counter()
                for _ in range(20):
                    eng.bottom_up(True)
                eng.sync()
                ms_bu = (time.perf_counter() - t0) / 20 * 1e3
                res[name] = (lnl, post if tip_perm is None else post[new_of_old], ms, ms_bu)
                print('k=%d %-6s numbering: marginal pass %.3f ms, bottom-up sweep %.3f ms' % (k, name, ms, ms_bu), flush=True)
        for name in ('height', 'height+shape'):
            print('   %s: ln L identical: %s; posteriors of column 3 identical: %s'
                  % (name, np.array_equal(res['bfs'][0], res[name][0]), np.array_equal(res['bfs'][1], res[name][1])))


if __name__ == '__main__':
    main()
