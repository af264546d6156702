"""
Randomised GPU parity sweep (``-m gpu``): many small random forests -- polytomies, several trees, zero-length branches,
missing / ambiguous tips, restricted internal nodes, one to three columns with their own parameters -- through the
C-ABI, against the oracle.  Shapes are drawn so that every kernel variant is hit: the F81 lane shapes (k from 2 to 130),
cherry fusion with 1-6 tips per cherry, units that leave the lane-parallel gather (more than four children, stored
children beyond the first two), the fused matrix-core sweeps of the eigen models (16 <= k <= 32) next to the
materialised-P kernels (HKY, other k), and forests small enough for the single-launch kernels as well as larger ones.
A case whose likelihood is zero must raise on both sides and name the same pair of nodes.
"""
import numpy as np
import pytest

from oracle import pastml_oracle as orc
from pastml_amd import hip
from pastml_amd.tree import FlatForest
from test_gpu_parity import LNL_RTOL, LOG10_ATOL, log_true, random_masks, random_spec

pytestmark = pytest.mark.gpu

N_CASES = int(__import__('os').environ.get('PASTML_FUZZ_CASES', '120'))


def draw_case(seed):
    rng = np.random.default_rng(50_000 + seed)
    kind = ['F81', 'F81', 'F81', 'EIGEN', 'EIGEN', 'HKY'][seed % 6]
    if kind == 'HKY':
        k = 4
    elif kind == 'EIGEN':
        k = int(rng.choice([3, 7, 16, 17, 20, 21, 24, 26, 29, 32, 33, 40, 48, 53, 61, 64, 65, 77, 96, 128, 140]))
    else:
        k = int(rng.choice([2, 3, 4, 5, 9, 16, 20, 31, 32, 33, 48, 64, 65, 100, 130, 257, 300, 512]))
    big = seed % 10 == 0
    n_tips = int(rng.integers(1500, 3000)) if big else int(rng.integers(3, 160))
    if k > 64:
        n_tips = min(n_tips, 200)
    # Zero-length branches only for the F81 family, whose P(0) is the identity exactly.  The matrix models get P(0)
    # through an eigen-decomposition or a cancelling closed form: zeros come out as +-1e-17 dust, in numpy as here but
    # not the same dust, and where a zero branch joins conflicting states that dust IS the likelihood.  Their zero
    # branches are covered by test_matrix_models_observed_tips_on_zero_branches (agreeing states).
    zero_frac = float(rng.choice([0.0, 0.0, 0.05, 0.2]))
    flat = FlatForest.random(n_tips, seed=seed, max_arity=int(rng.integers(2, 7)),
                             zero_frac=zero_frac if kind == 'F81' else 0.0, n_trees=int(rng.integers(1, 4)))
    C = int(rng.integers(1, 4))
    specs = [random_spec(kind, k, rng) for _ in range(C)]
    rates = [(float(rng.uniform(0.3, 4)), float(rng.choice([0.0, 0.0, 0.02])), float(rng.uniform(0.7, 1.0)))
             for _ in range(C)]
    masks = np.stack([random_masks(flat, k, rng, missing=float(rng.choice([0.0, 0.1, 0.3])),
                                   multi=float(rng.choice([0.0, 0.1])), internal=float(rng.choice([0.0, 0.05])))
                      for _ in range(C)])
    return kind, k, flat, specs, rates, masks


def compare_vectors(ours, ours_sf, ref, ref_sf, rows, what):
    """Every vector relative to its largest entry (entries below 1e-12 of it are rounding dust on both sides)."""
    la, lb = log_true(ours, ours_sf)[rows], log_true(ref, ref_sf)[rows]
    ma, mb = la.max(axis=1), lb.max(axis=1)
    np.testing.assert_allclose(ma, mb, rtol=0, atol=LOG10_ATOL, err_msg=what)
    with np.errstate(over='ignore', invalid='ignore'):
        np.testing.assert_allclose(10 ** (la - ma[:, None]), 10 ** (lb - mb[:, None]), rtol=1e-9, atol=1e-12,
                                   err_msg=what)


@pytest.mark.parametrize('seed', range(N_CASES))
def test_random_case(seed):
    kind, k, flat, specs, rates, masks = draw_case(seed)
    C = len(specs)
    refs, ref_errors = [], []
    for c in range(C):
        try:
            refs.append(orc.full_marginal_pass(flat, masks[c].astype(int), specs[c], *rates[c]))
            ref_errors.append(None)
        except orc.OracleLikelihoodError as e:
            refs.append(None)
            ref_errors.append(e)
        except ValueError:
            # a node whose marginal likelihoods are all zero: rescale_log takes the minimum of an empty array
            # (ml.py:151-171 does the same), so the reference has no answer for this input
            pytest.skip('the reference fails on this input (all-zero marginal likelihoods)')
    with hip.Engine(flat, C, k) as eng:
        eng.set_models(list(zip(specs, rates)))
        eng.set_masks(masks)
        if any(e is not None for e in ref_errors):
            with pytest.raises(hip.ZeroLikelihoodError) as err:
                eng.bottom_up(True)
            for c in range(C):
                if ref_errors[c] is None:
                    assert err.value.err_child[c] == -1
                else:
                    assert (err.value.err_parent[c], err.value.err_child[c]) == (ref_errors[c].parent, ref_errors[c].child)
            return
        lnl = eng.bottom_up(True)
        post, lh_sum, lh_sf = eng.top_down_marginals()
        bus = [(eng.download(hip.BUF_BU, c), eng.download(hip.BUF_BU_SF, c)) for c in range(C)]
        joint_ok = True
        try:
            lnl_j = eng.bottom_up(False)
            tables = [eng.download(hip.BUF_JOINT_TABLE, c) for c in range(C)]
            states = eng.joint_backtrace()
        except hip.ZeroLikelihoodError:
            joint_ok = False
        # MAP / MPPA selection on the device against the host restatement of ml.py:505-595, on the device's posteriors
        if joint_ok and np.all(np.isfinite(post)):
            from pastml_amd import ml
            eng.bottom_up(True)
            eng.top_down_marginals(posterior=False, lh=False)
            for method, fj in (('MPPA', True), ('MPPA', False), ('MAP', False)):
                sel, nsel = eng.select_states(method, force_joint=fj)
                for c in range(C):
                    if method == 'MAP':
                        ref_sel, ref_k = ml.select_map(post[c]), np.ones(flat.n_nodes, dtype=int)
                    else:
                        ref_sel, ref_k = ml.select_mppa(post[c], states[c].astype(np.int64) if fj else None)
                    assert np.array_equal(nsel[c], ref_k), (method, fj, c)
                    assert np.array_equal(sel[c], ref_sel), (method, fj, c)
                eng.set_masks(masks)
                eng.bottom_up(True)
                eng.top_down_marginals(posterior=False, lh=False)
    internal = ~flat.is_tip
    nonroot = flat.parent >= 0
    for c in range(C):
        r = refs[c]
        np.testing.assert_allclose(lnl[c], r['loglik'], rtol=LNL_RTOL, atol=1e-11)
        compare_vectors(bus[c][0], bus[c][1], r['bu'], r['bu_sf'], internal, 'BU col {}'.format(c))
        np.testing.assert_allclose(post[c], r['posterior'], rtol=1e-8, atol=1e-300)
        tot = np.log10(lh_sum[c]) - lh_sf[c]
        np.testing.assert_allclose(tot, r['loglik_per_tree'][flat.tree_id] / np.log(10), rtol=1e-10, atol=1e-11)
        if not joint_ok:
            with pytest.raises(orc.OracleLikelihoodError):
                orc.bottom_up(flat, masks[c].astype(int), specs[c], *rates[c], is_marginal=False)
            continue
        j = orc.bottom_up(flat, masks[c].astype(int), specs[c], *rates[c], is_marginal=False)
        np.testing.assert_allclose(lnl_j[c], j['loglik'], rtol=LNL_RTOL, atol=1e-11)
        diff = np.argwhere((tables[c] != j['joint_table']) & nonroot[:, None])
        if kind == 'F81':
            assert len(diff) == 0
            assert np.array_equal(states[c], orc.joint_backtrace(flat, j['bu'], j['joint_table'], specs[c]['pi']))
        else:
            # P(t) of the matrix models differs from numpy's in the last bits: a flip only between equal products
            for n, i in diff:
                prod = orc.pij(specs[c], flat.dist[n], *rates[c])[i] * j['bu'][n]
                assert abs(prod[tables[c][n, i]] - prod.max()) <= 1e-12 * max(prod.max(), 1e-300), (n, i)


@pytest.mark.parametrize('seed', range(max(8, N_CASES // 5)))
def test_random_case_with_two_level_units(seed):
    """The level schedule with two-level units (forced on a small forest: no subtree blocks, no single-launch sweeps, any
    number of such nodes) against the oracle: ragged forests with balanced clumps, 17 <= k <= 64, masks of every kind,
    several columns; ln L, bottom-up vectors (the clumps' inner nodes come from the download's materialisation),
    posteriors, totals; a zero likelihood names the reference's pair."""
    from test_gpu_parity import _forest_with_balanced_clumps
    rng = np.random.default_rng(70_000 + seed)
    k = int(rng.choice([17, 20, 24, 29, 31, 32, 33, 40, 48, 63, 64]))
    flat = _forest_with_balanced_clumps(int(rng.integers(20, 120)), seed=seed, clump_frac=float(rng.uniform(0.3, 0.9)))
    C = int(rng.integers(1, 4))
    specs = [random_spec('F81', k, rng) for _ in range(C)]
    rates = [(float(rng.uniform(0.3, 4)), float(rng.choice([0.0, 0.0, 0.02])), float(rng.uniform(0.7, 1.0)))
             for _ in range(C)]
    masks = np.stack([random_masks(flat, k, rng, missing=float(rng.choice([0.0, 0.1, 0.3])),
                                   multi=float(rng.choice([0.0, 0.1])), internal=float(rng.choice([0.0, 0.0, 0.05])))
                      for _ in range(C)])
    refs, ref_errors = [], []
    for c in range(C):
        try:
            refs.append(orc.full_marginal_pass(flat, masks[c].astype(int), specs[c], *rates[c]))
            ref_errors.append(None)
        except orc.OracleLikelihoodError as e:
            refs.append(None)
            ref_errors.append(e)
        except ValueError:
            pytest.skip('the reference fails on this input (all-zero marginal likelihoods)')
    tune = dict(BLOCK_NODES=0, SMALL_MANY_NODES=0, SMALL_MAX_NODES=0, SUPER_MIN=1, STACK_MIN=1)
    with hip.Engine(flat, C, k, tune=tune) as eng:
        eng.set_models(list(zip(specs, rates)))
        eng.set_masks(masks)
        eng.profile_enable(True)
        if any(e is not None for e in ref_errors):
            with pytest.raises(hip.ZeroLikelihoodError) as err:
                eng.bottom_up(True)
            for c in range(C):
                if ref_errors[c] is None:
                    assert err.value.err_child[c] == -1
                else:
                    assert (err.value.err_parent[c], err.value.err_child[c]) == (ref_errors[c].parent, ref_errors[c].child)
            return
        lnl = eng.bottom_up(True)
        post, lh_sum, lh_sf = eng.top_down_marginals()
        assert eng.profile_read(3)[1] > 0 and eng.profile_read(4)[1] > 0     # the two-level launches did run
        bus = [(eng.download(hip.BUF_BU, c), eng.download(hip.BUF_BU_SF, c)) for c in range(C)]
    internal = ~flat.is_tip
    for c in range(C):
        r = refs[c]
        np.testing.assert_allclose(lnl[c], r['loglik'], rtol=LNL_RTOL, atol=1e-11)
        compare_vectors(bus[c][0], bus[c][1], r['bu'], r['bu_sf'], internal, 'BU col {}'.format(c))
        np.testing.assert_allclose(post[c], r['posterior'], rtol=1e-8, atol=1e-300)
        tot = np.log10(lh_sum[c]) - lh_sf[c]
        np.testing.assert_allclose(tot, r['loglik_per_tree'][flat.tree_id] / np.log(10), rtol=1e-10, atol=1e-11)
