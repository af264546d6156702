"""
GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path, called through the C-ABI, against
(1) the golden vectors produced by the real reference, (2) the oracle on seeded random inputs,
(3) size-independent invariants at BASELINE.json's full sizes.

Tolerances.  north_star asks for 1e-6 relative on posteriors and log-likelihoods; we hold the kernels to
1e-9 relative on posteriors, 2e-10 in log10 units (5e-10 relative) on every likelihood vector entry and 1e-11 relative
on log-likelihoods.  Integer outputs (arg-max tables, joint states, selected states) must be identical.
"""
import os

import numpy as np
import pytest

from conftest import load_golden, golden_forest, golden_spec
from oracle import pastml_oracle as orc
from pastml_amd import hip, synthetic
from pastml_amd.tree import FlatForest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG10_ATOL = 2e-10
POST_RTOL = 1e-9
LNL_RTOL = 1e-11


def log_true(vec, sf):
    """log10 of the true (unscaled) values of a scaled vector array [N, k] with base-10 scales [N]."""
    with np.errstate(divide='ignore'):
        return np.log10(vec) - sf[:, None]


def assert_same_scaled(ours, ours_sf, ref, ref_sf, rows=None, what='', ulps=0):
    """ulps: slack in units of the spacing of the compared numbers (log10 values of -1e6 have a spacing of 2.3e-10)."""
    a, b = log_true(ours, ours_sf), log_true(ref, ref_sf)
    if rows is not None:
        a, b = a[rows], b[rows]
    assert np.array_equal(np.isneginf(a), np.isneginf(b)), what + ': zero patterns differ'
    fin = np.isfinite(b)
    if ulps:
        assert np.all(np.abs(a[fin] - b[fin]) <= LOG10_ATOL + ulps * np.spacing(np.abs(b[fin]))), what
    else:
        np.testing.assert_allclose(a[fin], b[fin], rtol=0, atol=LOG10_ATOL, err_msg=what)


# the level schedule on forests that would otherwise take the single-launch kernels or the subtree blocks, with two-level
# and stacked units from the first one on (per-engine switches: pml_ctx_set_tunable -- nothing is read from the
# environment here, and nothing is latched in the process)
LEVEL_SCHEDULE = dict(BLOCK_NODES=0, SMALL_MANY_NODES=0, SUPER_MIN=1, STACK_MIN=1)


@pytest.mark.parametrize('k', [2, 3, 6, 10, 15, 16, 20, 29, 32])
def test_pij_batch_kernels_agree(k):
    """The P(t) batch of an eigen model by each of its kernels -- the vector-unit kernel (default below 16 states, forced by
    PIJ_VALU), the matrix-core kernel (16 <= k <= 32) and the generic one (NO_PIJ_VALU below 16, odd row strides) -- against
    the oracle's A diag(exp(d t)) A^-1 and against each other (rounding only), on a forest whose row count is not a multiple of
    the kernels' pass sizes, two columns with different rates."""
    rng = np.random.default_rng(k)
    flat = FlatForest.random(333, seed=k, max_arity=3, zero_frac=0.1, n_trees=2)
    specs = [(random_spec('EIGEN', k, rng), (float(rng.uniform(0.5, 3)), 0.0, 1.0)) for _ in range(2)]
    got = {}
    for name, tune in (('default', {}), ('valu', dict(PIJ_VALU=1)), ('other', dict(NO_PIJ_VALU=1))):
        with hip.Engine(flat, 2, k, tune=tune) as eng:
            eng.set_models(specs)
            got[name] = eng.pij_batch(copy_out=True)
    for col, (spec, rates) in enumerate(specs):
        ref = np.array([orc.pij(spec, t, *rates) for t in flat.dist])
        for name in got:
            np.testing.assert_allclose(got[name][col], ref, rtol=1e-11, atol=5e-15, err_msg='{} k={}'.format(name, k))
    np.testing.assert_allclose(got['valu'], got['other'], rtol=1e-11, atol=5e-15)


@pytest.mark.parametrize('k', [33, 40, 48, 61, 64, 67, 100, 128, 130, 200, 255, 256])
def test_pij_batch_beyond_32_states(k):
    """The P(t) batch of an eigen model with more than 32 states: the matrix-core kernel with A^T in LDS slices
    (pij_eigen_wide_kernel: one slice up to 128 states, two to four beyond; odd k pads its rows to an even stride, the last
    row and column tiles are partial) against the oracle's A diag(exp(d t)) A^-1 (pastml/models/generator.py:54-65) and against
    the one-thread-per-entry kernel it replaces (NO_PIJ_WIDE), rounding only; a forest with zero-length branches whose node
    count is not a multiple of the waves per workgroup, two columns with different rates."""
    rng = np.random.default_rng(k)
    flat = FlatForest.random(101 if k > 128 else 333, seed=k, max_arity=3, zero_frac=0.1, n_trees=2)
    specs = [(random_spec('EIGEN', k, rng), (float(rng.uniform(0.5, 3)), 0.0, 1.0)) for _ in range(2)]
    got = {}
    for name, tune in (('wide', {}), ('generic', dict(NO_PIJ_WIDE=1))):
        with hip.Engine(flat, 2, k, tune=tune) as eng:
            eng.set_models(specs)
            got[name] = eng.pij_batch(copy_out=True)
    for col, (spec, rates) in enumerate(specs):
        ref = np.array([orc.pij(spec, t, *rates) for t in flat.dist])
        for name in got:
            np.testing.assert_allclose(got[name][col], ref, rtol=1e-10, atol=1e-14, err_msg='{} k={}'.format(name, k))
    np.testing.assert_allclose(got['wide'], got['generic'], rtol=1e-10, atol=1e-14)


@pytest.mark.parametrize('k', [65, 67, 80, 81, 96, 100, 113, 128])
def test_eigen_sum_sweeps_of_65_to_128_states(k):
    """
    Eigen models with 65 - 128 states (round 6): the sum sweeps as two matrix-core GEMMs per 16 nodes with ONE matrix in LDS --
    a reversible model's A^-1 is A transposed and rescaled (pml_kernels_eigen_gemm.h, EigGemm::SYM; pml_model_set_eigen checks
    the identity on what it is given) -- against the oracle (pastml/ml.py:124-148, :273-290, :454-460 with the reference's P(t),
    generator.py:54-65) and against the sweeps on P(t) materialised in HBM (NO_EIGEN_GEMM), which the fused path must not
    allocate; polytomies, several trees, missing / ambiguous tips (masks of two words), restricted internal nodes, tau > 0.
    Tolerances: ln L 1e-11 and posteriors 1e-9 as everywhere; the vectors' entries 1e-9 in log10 (elsewhere 2e-10): a product of the
    one-matrix form agrees with the reference's P(t) v to 7e-11 per entry (numpy's eigenvectors re-orthonormalised; with numpy's
    own inverse it is 2e-11), and the small entries of vectors 50 levels deep collect several of those.
    """
    rng = np.random.default_rng(6500 + k)
    flat = FlatForest.random(700, seed=k, max_arity=4, n_trees=2)
    C = 2
    specs = [random_spec('EIGEN', k, rng) for _ in range(C)]
    rates = [(float(rng.uniform(0.5, 3)), 0.0, 1.0), (float(rng.uniform(0.5, 3)), 0.01, 0.9)]
    masks = np.stack([random_masks(flat, k, rng) for _ in range(C)])
    masks[1] = synthetic.one_hot_masks(flat, k, rng.integers(0, k, size=flat.n_tips))   # every tip observed: the gathered first product
    got = {}
    for name, tune in (('fused', {}), ('materialised', dict(NO_EIGEN_GEMM=1))):
        with hip.Engine(flat, C, k, tune=tune, keep_td=True) as eng:
            eng.set_models(list(zip(specs, rates)))
            eng.set_masks(masks)
            lnl, post, lh_sum, lh_sf = eng.marginal_pass()
            held = eng.memory()[0]   # (before the top-down vectors are asked for: those of tips are filled in from P(t))
            bus = [(eng.download(hip.BUF_BU, c), eng.download(hip.BUF_BU_SF, c)) for c in range(C)]
            tds = [(eng.download(hip.BUF_TD, c), eng.download(hip.BUF_TD_SF, c)) for c in range(C)]
            lnl2 = eng.bottom_up(True)
            assert np.array_equal(lnl, lnl2)
            got[name] = (lnl, post, lh_sum, lh_sf, bus, tds, held)
    P_bytes = C * flat.n_nodes * k * (k + (k & 1)) * 8
    assert got['fused'][6] <= got['materialised'][6] - P_bytes // 2, 'the fused sweeps allocated P(t)'
    nonroot = flat.parent >= 0
    for c in range(C):
        r = orc.full_marginal_pass(flat, masks[c].astype(int), specs[c], *rates[c])
        for name in got:
            lnl, post, lh_sum, lh_sf, bus, tds, _ = got[name]
            np.testing.assert_allclose(lnl[c], r['loglik'], rtol=LNL_RTOL, atol=1e-12, err_msg=name)
            for what, (vec, sf), key in (('BU', bus[c], 'bu'), ('TD', tds[c], 'td')):
                a, b = log_true(vec, sf)[nonroot], log_true(r[key], r[key + '_sf'])[nonroot]
                assert np.array_equal(np.isneginf(a), np.isneginf(b)), '{} {} col {}: zero patterns differ'.format(what, name, c)
                fin = np.isfinite(b)
                np.testing.assert_allclose(a[fin], b[fin], rtol=0, atol=1e-9 if name == 'fused' else LOG10_ATOL,
                                           err_msg='{} {} col {}'.format(what, name, c))
            np.testing.assert_allclose(post[c], r['posterior'], rtol=POST_RTOL, atol=1e-300, err_msg=name)
            np.testing.assert_allclose(np.log10(lh_sum[c]) - lh_sf[c], r['loglik_per_tree'][flat.tree_id] / np.log(10), rtol=1e-11,
                                       atol=1e-12)


def test_eigen_model_with_a_repeated_eigenvalue_keeps_materialised_transition_matrices():
    """A 70-state model one of whose eigenvalues repeats, handed over with eigenvectors of that eigenvalue that are not orthogonal
    (any basis of the eigenspace diagonalises; numpy gives no better for a repeated eigenvalue): A^-1 is then not A transposed
    and rescaled -- pml_model_set_eigen notices and the sweeps read P(t) from HBM, bit for bit the results of NO_EIGEN_GEMM."""
    k = 70
    rng = np.random.default_rng(70)
    flat = FlatForest.random(300, seed=70, max_arity=3)
    spec = random_spec('EIGEN', k, rng)
    order = np.argsort(spec['d'])
    gaps = np.diff(spec['d'][order])
    m0, m1 = order[np.argmin(gaps)], order[np.argmin(gaps) + 1]   # the two closest eigenvalues become one
    d = spec['d'].copy()
    d[m1] = d[m0]
    a = spec['A'].copy()
    a[:, m1] += 0.7 * a[:, m0]
    spec = dict(kind=2, pi=spec['pi'], d=d, A=a, Ainv=np.linalg.inv(a))
    masks = random_masks(flat, k, rng)[None]
    got = []
    for tune in ({}, dict(NO_EIGEN_GEMM=1)):
        with hip.Engine(flat, 1, k, tune=tune) as eng:
            eng.set_models([(spec, (1.3, 0.0, 1.0))])
            eng.set_masks(masks)
            got.append(eng.marginal_pass())
    for x, y in zip(got[0], got[1]):
        assert np.array_equal(x, y)
    r = orc.full_marginal_pass(flat, masks[0].astype(int), spec, 1.3, 0.0, 1.0)
    np.testing.assert_allclose(got[0][0][0], r['loglik'], rtol=LNL_RTOL)
    np.testing.assert_allclose(got[0][1][0], r['posterior'], rtol=POST_RTOL, atol=1e-300)


# ---------------------------------------------------------------------------------------------------------------------
def test_pij_matches_reference():
    z = load_golden('pij')
    ts = z['ts']
    flat = FlatForest([-1], [0], [1], [0.0], [0])
    for i, label in enumerate(z['labels']):
        spec, rates = golden_spec(z, 'c{}_'.format(i))
        k = len(spec['pi'])
        with hip.Engine(flat, 1, k) as eng:
            eng.set_models([(spec, rates)])
            P = eng.pij(ts)
        np.testing.assert_allclose(P, z['c{}_P'.format(i)], rtol=1e-12, atol=2e-15, err_msg=str(label))


def test_pij_batch_matches_pij():
    """The per-branch matrices the sweeps consume == the explicit ones (all kinds)."""
    z = load_golden('pij')
    flat = FlatForest.random(40, seed=8, max_arity=3, zero_frac=0.2)
    for i, label in enumerate(z['labels']):
        spec, rates = golden_spec(z, 'c{}_'.format(i))
        k = len(spec['pi'])
        with hip.Engine(flat, 2, k) as eng:
            eng.set_models([(spec, rates), (spec, (rates[0] * 2, rates[1], rates[2]))])
            Pb = eng.pij_batch(copy_out=True)
            for col in (0, 1):
                np.testing.assert_allclose(Pb[col], eng.pij(flat.dist, col), rtol=1e-12, atol=2e-15, err_msg=str(label))
        ref = np.array([orc.pij(spec, t, *rates) for t in flat.dist])
        np.testing.assert_allclose(Pb[0], ref, rtol=1e-12, atol=2e-15, err_msg=str(label))


SWEEP_CASES = [('albania_F81', 'fix_'), ('albania_JC', 'fix_'), ('albania_EFT', 'fix_'),
               ('albania_F81', 'tau_'), ('albania_JC', 'tau_'), ('albania_EFT', 'tau_'),
               ('synthetic_jc_k4_L10', ''), ('synthetic_f81_k64_L8', ''), ('synthetic_jtt_k20_L8', ''),
               ('synthetic_hky_L8', ''), ('synthetic_f81_k5_L9', ''), ('synthetic_f81_k67_L5', ''),
               ('synthetic_f81_k130_L4', ''),
               ('edge_poly', ''), ('edge_zero', ''), ('edge_zero_tau', ''), ('edge_forest', '')]


@pytest.mark.parametrize('name,prefix', SWEEP_CASES)
def test_sweeps_match_reference(name, prefix):
    z = load_golden(name)
    flat = golden_forest(z)
    spec, rates = golden_spec(z, {'fix_': 'opt_', 'tau_': 'tau_', '': ''}[prefix])
    g = lambda key: z[prefix + key]
    k = len(spec['pi'])
    internal = ~flat.is_tip
    with hip.Engine(flat, 1, k) as eng:
        eng.set_models([(spec, rates)])
        eng.set_masks(g('masks_altered'))
        # ---- marginal
        lnl = eng.bottom_up(True)
        np.testing.assert_allclose(lnl[0], g('loglik'), rtol=LNL_RTOL)
        assert_same_scaled(eng.download(hip.BUF_BU), eng.download(hip.BUF_BU_SF), g('bu'), g('bu_sf'), what='BU')
        post, lh_sum, lh_sf = eng.top_down_marginals()
        # every node has its top-down vector (ml.py:273-290): those the sweeps do not store (tips; fused cherries) are
        # filled in when the buffer is asked for
        td, td_sf = eng.download(hip.BUF_TD), eng.download(hip.BUF_TD_SF)
        assert not np.isnan(td).any() and not np.isnan(td_sf).any()
        assert_same_scaled(td, td_sf, g('td'), g('td_sf'), what='TD')
        np.testing.assert_allclose(post[0], g('posterior'), rtol=POST_RTOL, atol=1e-300)
        # LH / LH_SF as the host rebuilds them: posterior * lh_sum with scale lh_sf
        assert_same_scaled(post[0] * lh_sum[0][:, None], lh_sf[0], g('lh'), g('lh_sf'), what='LH')
        # ---- joint
        eng.set_initial_masks(g('masks_initial'))
        lnl_j = eng.bottom_up(False)
        np.testing.assert_allclose(lnl_j[0], g('loglik_joint'), rtol=LNL_RTOL)
        table = eng.download(hip.BUF_JOINT_TABLE)
        nonroot = flat.parent >= 0
        assert np.array_equal(table[nonroot], g('joint_table')[nonroot])
        assert_same_scaled(eng.download(hip.BUF_BU), eng.download(hip.BUF_BU_SF), g('bu_joint'), g('bu_joint_sf'),
                           what='BU joint')
        states = eng.joint_backtrace()
        assert np.array_equal(states[0], g('joint_state'))
        eng.set_initial_masks(None)
        # ---- restricted likelihoods (when nothing is altered the selected masks are swept as they are)
        if len(g('altered_nodes')) == 0:
            for key, m in (('loglik_restricted_MAP', g('masks_map')), ('loglik_restricted_MPPA', g('masks_mppa'))):
                eng.set_masks(m)
                np.testing.assert_allclose(eng.bottom_up(True)[0], g(key), rtol=LNL_RTOL)


def test_zero_likelihood_reports_reference_pair():
    z = load_golden('edge_zero_likelihood')
    flat = golden_forest(z)
    spec, rates = golden_spec(z)
    with hip.Engine(flat, 2, len(spec['pi'])) as eng:
        eng.set_models([(spec, rates)] * 2)
        ok = np.ones_like(z['masks'])
        ok[flat.tips] = z['masks'][flat.tips]
        eng.set_masks(np.stack([ok, z['masks']]))
        with pytest.raises(hip.ZeroLikelihoodError) as e:
            eng.bottom_up(True)
        assert e.value.err_parent[0] == -1 and e.value.err_child[0] == -1
        names = z['node_names']
        msg = str(z['error_message'])
        assert 'parent node {} and its child node {}'.format(names[e.value.err_parent[1]],
                                                             names[e.value.err_child[1]]) in msg
        # the healthy column is still right
        eng.set_masks(np.stack([ok, ok]))
        lnl = eng.bottom_up(True)
        ref = orc.bottom_up(flat, ok.astype(int), spec, *rates)['loglik']
        np.testing.assert_allclose(lnl, [ref, ref], rtol=LNL_RTOL)


# ---------------------------------------------------------------------------------------------------------------------
def random_spec(kind, k, rng):
    pi = rng.dirichlet(np.ones(k) * 2)
    if kind == 'F81':
        return dict(kind=0, pi=pi)
    if kind == 'HKY':
        return dict(kind=1, pi=pi, kappa=float(rng.uniform(0.5, 8)))
    R = np.triu(rng.uniform(0.05, 3, size=(k, k)), 1)
    R = R + R.T
    d, a, ainv = orc.diagonalise(pi, R)
    return dict(kind=2, pi=pi, d=d, A=a, Ainv=ainv)


def random_masks(flat, k, rng, missing=0.1, multi=0.1, internal=0.05):
    masks = np.ones((flat.n_nodes, k), dtype=np.int8)
    for n in range(flat.n_nodes):
        u = rng.random()
        if flat.is_tip[n]:
            if u < missing:
                continue
            masks[n] = 0
            if u < missing + multi and k > 1:
                masks[n, rng.choice(k, size=2, replace=False)] = 1
            else:
                masks[n, rng.integers(k)] = 1
        elif u < internal:
            masks[n] = 0
            masks[n, rng.integers(k)] = 1
    return masks


@pytest.mark.parametrize('kind,k', [('F81', 1), ('F81', 2), ('F81', 3), ('F81', 7), ('F81', 12), ('F81', 20),
                                    ('F81', 32), ('F81', 33), ('F81', 64), ('F81', 65), ('F81', 100), ('F81', 129),
                                    ('F81', 200), ('F81', 256), ('HKY', 4), ('EIGEN', 2), ('EIGEN', 5), ('EIGEN', 20),
                                    ('EIGEN', 36), ('EIGEN', 67), ('EIGEN', 130)])
def test_random_forests_vs_oracle(kind, k):
    """Seeded random forests (polytomies, missing / ambiguous tips, restricted internal nodes), 3 columns at once."""
    rng = np.random.default_rng(1000 + k)
    n_tips = 120 if k <= 67 else 40
    flat = FlatForest.random(n_tips, seed=k, max_arity=4, zero_frac=0.0, n_trees=2)
    C = 3
    specs = [random_spec(kind, k, rng) for _ in range(C)]
    rates = [(float(rng.uniform(0.5, 3)), 0.0, 1.0), (float(rng.uniform(0.5, 3)), 0.01, 0.9), (1.0, 0.0, 1.0)]
    masks = np.stack([random_masks(flat, k, rng) for _ in range(C)])
    with hip.Engine(flat, C, k) as eng:
        eng.set_models(list(zip(specs, rates)))
        eng.set_masks(masks)
        lnl = eng.bottom_up(True)
        post, lh_sum, lh_sf = eng.top_down_marginals()
        bus = [(eng.download(hip.BUF_BU, c), eng.download(hip.BUF_BU_SF, c)) for c in range(C)]
        lnl_j = eng.bottom_up(False)
        tables = [eng.download(hip.BUF_JOINT_TABLE, c) for c in range(C)]
        states = eng.joint_backtrace()
    nonroot = flat.parent >= 0
    for c in range(C):
        r = orc.full_marginal_pass(flat, masks[c].astype(int), specs[c], *rates[c])
        np.testing.assert_allclose(lnl[c], r['loglik'], rtol=LNL_RTOL, atol=1e-12)
        assert_same_scaled(bus[c][0], bus[c][1], r['bu'], r['bu_sf'], what='BU col {}'.format(c))
        np.testing.assert_allclose(post[c], r['posterior'], rtol=POST_RTOL, atol=1e-300)
        tot = np.log10(lh_sum[c]) - lh_sf[c]
        np.testing.assert_allclose(tot, r['loglik_per_tree'][flat.tree_id] / np.log(10), rtol=1e-11, atol=1e-12)
        j = orc.bottom_up(flat, masks[c].astype(int), specs[c], *rates[c], is_marginal=False)
        np.testing.assert_allclose(lnl_j[c], j['loglik'], rtol=LNL_RTOL, atol=1e-12)
        if kind != 'EIGEN':
            assert np.array_equal(tables[c][nonroot], j['joint_table'][nonroot])
            assert np.array_equal(states[c], orc.joint_backtrace(flat, j['bu'], j['joint_table'], specs[c]['pi']))
        else:
            # eigen P(t) differs from numpy's BLAS in the last bits: allow arg-max flips only between products that
            # are equal to 1e-12
            diff = np.argwhere(tables[c] != j['joint_table'])
            for n, i in diff:
                if not nonroot[n]:
                    continue
                prod = orc.pij(specs[c], flat.dist[n], *rates[c])[i] * j['bu'][n]
                assert abs(prod[tables[c][n, i]] - prod.max()) <= 1e-12 * prod.max()


def test_deep_caterpillar_rescaling():
    """A 3000-level caterpillar drives the likelihoods through ~1e-2000: exercises the exponent bookkeeping."""
    n = 3000
    # ids in level order: level d has one internal node (2d-1 ... ) -- build with objects then flatten
    from pastml_amd.tree import TreeNode
    root = TreeNode(name='r', dist=0.0)
    cur = root
    for d in range(n):
        cur.add_child(name='t{}'.format(d), dist=0.3)
        cur = cur.add_child(name='i{}'.format(d), dist=0.05)
    cur.add_child(name='ta', dist=0.1)
    cur.add_child(name='tb', dist=0.1)
    flat = FlatForest.from_trees([root])
    k = 6
    rng = np.random.default_rng(5)
    spec = random_spec('F81', k, rng)
    masks = random_masks(flat, k, rng, missing=0.0, multi=0.0, internal=0.0)
    with hip.Engine(flat, 1, k) as eng:
        eng.set_models([(spec, (1.0, 0.0, 1.0))])
        eng.set_masks(masks)
        lnl = eng.bottom_up(True)
        post, lh_sum, lh_sf = eng.top_down_marginals()
    r = orc.full_marginal_pass(flat, masks.astype(int), spec)
    assert r['loglik'] < -4000
    np.testing.assert_allclose(lnl[0], r['loglik'], rtol=LNL_RTOL)
    np.testing.assert_allclose(post[0], r['posterior'], rtol=POST_RTOL, atol=1e-300)
    np.testing.assert_allclose(np.log10(lh_sum[0]) - lh_sf[0], r['loglik'] / np.log(10), rtol=1e-11)


def test_determinism():
    """tests/CUSTOM_RATESTest.py:59,84,89 require bit-identical reruns."""
    z = load_golden('synthetic_jtt_k20_L8')
    flat = golden_forest(z)
    spec, rates = golden_spec(z)
    outs = []
    for _ in range(2):
        with hip.Engine(flat, 1, 20) as eng:
            eng.set_models([(spec, rates)])
            eng.set_masks(z['masks_altered'])
            lnl = eng.bottom_up(True)
            post, _, _ = eng.top_down_marginals()
            outs.append((lnl.copy(), post.copy()))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])


# ---------------------------------------------------------------------------------------------------------------------
def test_cfg2_full_size_matches_reference_sample():
    """BASELINE config 2: 65 536 tips, JC k=4, marginal; reference posteriors at every 4 099th node."""
    z = load_golden('synthetic_cfg2_full')
    flat = synthetic.balanced_forest(int(z['n_levels']))
    spec, rates = golden_spec(z)
    with hip.Engine(flat, 1, 4) as eng:
        eng.set_models([(spec, rates)])
        eng.set_tip_states(synthetic.tip_states(flat.n_tips, 4, 0))
        lnl = eng.bottom_up(True)
        post, lh_sum, lh_sf = eng.top_down_marginals()
        lnl_j = eng.bottom_up(False)
        states = eng.joint_backtrace()
    s = z['sample']
    np.testing.assert_allclose(lnl[0], z['loglik'], rtol=LNL_RTOL)
    np.testing.assert_allclose(post[0][s], z['posterior'], rtol=POST_RTOL, atol=1e-300)
    np.testing.assert_allclose(lnl_j[0], z['loglik_joint'], rtol=LNL_RTOL)
    assert np.array_equal(states[0][s], z['joint_state'])
    np.testing.assert_allclose(post[0].sum(axis=1), 1, rtol=1e-12)
    np.testing.assert_allclose(np.log10(lh_sum[0]) - lh_sf[0], lnl[0] / np.log(10), rtol=1e-11)


@pytest.mark.parametrize('fused', [True, False])
def test_custom_rates_k61_matches_reference(fused):
    """
    A codon-sized eigen model: the reference's CustomRatesModel (pastml/models/CustomRatesModel.py:70-79,
    generator.py:54-65) with 61 states on a balanced 4 096-tip tree (tests/golden/make_golden.py::case_eigen_k61) --
    marginal pass and joint sweep + back-trace.  fused: the sum sweeps as two matrix-core GEMMs per 16 nodes with the
    constant operands in LDS (pml_kernels_eigen_gemm.h, P(t) is never formed) and the joint sweep with P(t) built and folded
    in registers on the vector units (pml_kernels_eigen_joint.h, a node per wavefront beyond 32 states); not fused: P(t) of
    every branch materialised in HBM (the generic path, what every k > 32 took before round 6).  Both against the reference.
    """
    z = load_golden('synthetic_cr_k61_L12')
    k = 61
    flat = synthetic.balanced_forest(int(z['n_levels']))
    spec, rates = golden_spec(z)
    s = z['sample']
    with hip.Engine(flat, 1, k, tune={} if fused else dict(NO_EIGEN_GEMM=1, NO_EIGEN_JOINT_VALU=1)) as eng:
        eng.set_models([(spec, rates)])
        eng.set_tip_states(z['tip_states'])
        lnl = eng.bottom_up(True)
        np.testing.assert_allclose(lnl[0], z['loglik'], rtol=LNL_RTOL)
        assert_same_scaled(eng.download(hip.BUF_BU)[s], eng.download(hip.BUF_BU_SF)[s], z['bu'], z['bu_sf'][s], what='BU')
        post, lh_sum, lh_sf = eng.top_down_marginals()
        np.testing.assert_allclose(post[0][s], z['posterior'], rtol=POST_RTOL, atol=1e-300)
        np.testing.assert_allclose(post[0].sum(axis=1), 1, rtol=1e-12)
        np.testing.assert_allclose(np.log10(lh_sum[0]) - lh_sf[0], lnl[0] / np.log(10), rtol=1e-11)
        assert_same_scaled(eng.download(hip.BUF_TD)[s], eng.download(hip.BUF_TD_SF)[s], z['td'], z['td_sf'][s], what='TD')
        lnl_j = eng.bottom_up(False)
        np.testing.assert_allclose(lnl_j[0], z['loglik_joint'], rtol=LNL_RTOL)
        nonroot = flat.parent[s] >= 0
        assert np.array_equal(eng.download(hip.BUF_JOINT_TABLE)[s][nonroot], z['joint_table'][nonroot])
        assert np.array_equal(eng.joint_backtrace()[0], z['joint_state'])
        # the marginal sweep again after the joint one (the two share the bottom-up buffers)
        assert np.array_equal(eng.bottom_up(True), lnl)


def test_f81_k300_matches_reference():
    """
    More than 256 states against the reference itself (tests/golden/make_golden.py::case_f81_k300: pastml's F81Model with 300
    states on a balanced 1 024-tip tree, a tenth of the tips unannotated): marginal pass, joint sweep with 16-bit arg-max
    tables + back-trace, MAP / MPPA selection on the device and the restricted likelihoods of the selected masks.
    """
    z = load_golden('synthetic_f81_k300_L10')
    k = 300
    flat = synthetic.balanced_forest(int(z['n_levels']))
    masks = synthetic.one_hot_masks(flat, k, z['tip_states'])
    masks[np.asarray(flat.tips)[~z['tip_observed']]] = 1
    spec, rates = golden_spec(z)
    s = z['sample']
    with hip.Engine(flat, 1, k, keep_td=True) as eng:
        eng.set_models([(spec, rates)])
        eng.set_masks(masks[None])
        lnl = eng.bottom_up(True)
        np.testing.assert_allclose(lnl[0], z['loglik'], rtol=LNL_RTOL)
        assert_same_scaled(eng.download(hip.BUF_BU)[s], eng.download(hip.BUF_BU_SF)[s], z['bu'], z['bu_sf'][s], what='BU')
        post, lh_sum, lh_sf = eng.top_down_marginals()
        np.testing.assert_allclose(post[0][s], z['posterior'], rtol=POST_RTOL, atol=1e-300)
        assert_same_scaled(eng.download(hip.BUF_TD)[s], eng.download(hip.BUF_TD_SF)[s], z['td'], z['td_sf'][s], what='TD')
        assert_same_scaled((post[0] * lh_sum[0][:, None])[s], lh_sf[0][s], z['lh'], z['lh_sf'][s], what='LH')
        lnl_j = eng.bottom_up(False)
        np.testing.assert_allclose(lnl_j[0], z['loglik_joint'], rtol=LNL_RTOL)
        nonroot = flat.parent[s] >= 0
        assert np.array_equal(eng.download(hip.BUF_JOINT_TABLE)[s][nonroot], z['joint_table'][nonroot])
        states = eng.joint_backtrace()
        assert np.array_equal(states[0], z['joint_state'])
        eng.bottom_up(True)
        eng.top_down_marginals()
        sel, nsel = eng.select_states('MAP')
        assert np.array_equal(sel[0][s], z['masks_map'])
        np.testing.assert_allclose(eng.bottom_up(True)[0], z['loglik_restricted_MAP'], rtol=LNL_RTOL)
        eng.set_masks(masks[None])
        eng.bottom_up(True)
        eng.top_down_marginals()
        sel, nsel = eng.select_states('MPPA', force_joint=bool(z['force_joint']))
        assert np.array_equal(sel[0][s], z['masks_mppa'])
        assert int((nsel[0] > 1).sum()) == int(z['mppa_num_unresolved']) and int(nsel[0].sum()) == int(z['mppa_num_states'])
        np.testing.assert_allclose(eng.bottom_up(True)[0], z['loglik_restricted_MPPA'], rtol=LNL_RTOL)


@pytest.mark.parametrize('fused', [True, False])
def test_custom_rates_k100_matches_reference(fused):
    """
    An eigen model beyond 64 states against the reference itself (tests/golden/make_golden.py::case_eigen_k100: pastml's
    CustomRatesModel with 100 states on a balanced 2 048-tip tree, a twentieth of the tips unannotated) -- marginal pass, joint
    sweep + back-trace.  fused: the sum sweeps as two matrix-core GEMMs per 16 nodes with ONE matrix in LDS (the model is
    reversible: pml_kernels_eigen_gemm.h, EigGemm::SYM); not fused: every sweep on P(t) of every branch in HBM, built by the
    matrix-core batch (pml_kernels_pij_wide.h) -- which the joint sweep reads in both.
    """
    z = load_golden('synthetic_cr_k100_L11')
    k = 100
    flat = synthetic.balanced_forest(int(z['n_levels']))
    masks = synthetic.one_hot_masks(flat, k, z['tip_states'])
    masks[np.asarray(flat.tips)[~z['tip_observed']]] = 1
    spec, rates = golden_spec(z)
    s = z['sample']
    with hip.Engine(flat, 1, k, tune={} if fused else dict(NO_EIGEN_GEMM=1), keep_td=True) as eng:
        eng.set_models([(spec, rates)])
        eng.set_masks(masks[None])
        lnl = eng.bottom_up(True)
        np.testing.assert_allclose(lnl[0], z['loglik'], rtol=LNL_RTOL)
        assert_same_scaled(eng.download(hip.BUF_BU)[s], eng.download(hip.BUF_BU_SF)[s], z['bu'], z['bu_sf'][s], what='BU')
        post, lh_sum, lh_sf = eng.top_down_marginals()
        np.testing.assert_allclose(post[0][s], z['posterior'], rtol=POST_RTOL, atol=1e-300)
        np.testing.assert_allclose(post[0].sum(axis=1), 1, rtol=1e-12)
        np.testing.assert_allclose(np.log10(lh_sum[0]) - lh_sf[0], lnl[0] / np.log(10), rtol=1e-11)
        assert_same_scaled(eng.download(hip.BUF_TD)[s], eng.download(hip.BUF_TD_SF)[s], z['td'], z['td_sf'][s], what='TD')
        lnl_j = eng.bottom_up(False)
        np.testing.assert_allclose(lnl_j[0], z['loglik_joint'], rtol=LNL_RTOL)
        nonroot = flat.parent[s] >= 0
        table = eng.download(hip.BUF_JOINT_TABLE)[s]
        # (the reference's P(t) is numpy's product; arg-max rows may differ where two products agree to rounding)
        diff = np.argwhere((table != z['joint_table']) & nonroot[:, None])
        assert len(diff) <= 2, len(diff)
        assert np.array_equal(eng.joint_backtrace()[0], z['joint_state'])
        assert np.array_equal(eng.bottom_up(True), lnl)


def test_cfg3_full_size_matches_reference_sample():
    """
    BASELINE config 3 at full size: 262 144 tips, JTT k=20, one character, joint (Pupko) sweep + back-trace
    (pastml/ml.py:82-148 with is_marginal=False, :598-622) on every joint sweep the library has for it: the default on
    the FP64 vector units (eigen_joint_kernel), the fused FP64 matrix-core sweep (PML_OPT_EIGEN_JOINT_VALU = 0) and the
    sweeps that read materialised P(t) (additionally PML_OPT_EIGEN_FUSED = 0).  Reference: lnL (joint and marginal), the
    joint state of EVERY node, arg-max rows and log10 bottom-up vectors at every 4 099th node.
    """
    z = load_golden('synthetic_cfg3_full')
    flat = synthetic.balanced_forest(int(z['n_levels']))
    spec, rates = golden_spec(z)
    s = z['sample']
    internal = flat.n_children[s] > 0
    results = []
    for valu, fused in ((True, True), (False, True), (False, False)):
        with hip.Engine(flat, 1, 20) as eng:
            eng.set_option(hip.OPT_EIGEN_JOINT_VALU, valu)
            eng.set_option(hip.OPT_EIGEN_FUSED, fused)
            eng.set_models([(spec, rates)])
            eng.set_tip_states(synthetic.tip_states(flat.n_tips, 20, 0))
            lnl = eng.bottom_up(True)
            lnl_j = eng.bottom_up(False)
            states = eng.joint_backtrace()
            table = eng.download(hip.BUF_JOINT_TABLE)
            bu = eng.download(hip.BUF_BU)
            bu_sf = eng.download(hip.BUF_BU_SF)
        np.testing.assert_allclose(lnl[0], z['loglik'], rtol=LNL_RTOL)
        np.testing.assert_allclose(lnl_j[0], z['loglik_joint'], rtol=LNL_RTOL)
        assert np.array_equal(states[0], z['joint_state'])
        nonroot = s > 0
        assert np.array_equal(table[s][nonroot], z['joint_table'][nonroot])
        assert_same_scaled(bu[s][internal], bu_sf[s][internal], z['bu_joint'][internal], z['bu_joint_sf'][internal],
                           what='joint BU, vector units={}, fused={}'.format(valu, fused))
        results.append((lnl[0], lnl_j[0]))
    # the three schedules build P(t) differently (FMA chains, accumulators, HBM): same numbers to rounding
    np.testing.assert_allclose(results[0], results[1], rtol=1e-12)
    np.testing.assert_allclose(results[0], results[2], rtol=1e-12)


def test_cfg4_bench_shape_32_columns():
    """
    BASELINE config 4 exactly as bench.py runs it: 1 048 576 tips, k=64, F81, 32 characters in one context (72 GB).
    Columns are independent and every kernel is deterministic, so column c of the 32-column run must equal, bit for bit,
    the same character run alone: ln L for all 32, posteriors / LH sums on a strided node sample for five of them.
    Plus, for every column, the invariants of bench.py's validation.
    """
    import bench
    L, C, k = 20, 32, 64
    flat = synthetic.balanced_forest(L)
    states = np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)])
    specs = [(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(C)]
    stride = 4099
    with hip.Engine(flat, C, k) as eng:
        eng.set_models(specs)
        eng.set_tip_states(states)
        lnl = eng.bottom_up(True)
        eng.top_down_marginals(posterior=False, lh=False)
        report = bench.validate_columns(eng, flat, k, states, lnl, stride=stride)
        assert report['columns'] == C and report['max_row_sum_error'] < 1e-12
        sample = {c: (eng.download_strided(hip.BUF_POSTERIOR, c, 0, stride),
                      eng.download_strided(hip.BUF_LH_SUM, c, 0, stride),
                      eng.download_strided(hip.BUF_LH_SF, c, 0, stride)) for c in (0, 1, 13, 30, 31)}
    with hip.Engine(flat, 1, k) as eng:
        for c in range(C):
            eng.set_models([specs[c]])
            eng.set_tip_states(states[c])
            alone = eng.bottom_up(True)
            assert alone[0] == lnl[c], 'column {}'.format(c)
            if c in sample:
                eng.top_down_marginals(posterior=False, lh=False)
                assert np.array_equal(eng.download_strided(hip.BUF_POSTERIOR, 0, 0, stride), sample[c][0])
                assert np.array_equal(eng.download_strided(hip.BUF_LH_SUM, 0, 0, stride), sample[c][1])
                assert np.array_equal(eng.download_strided(hip.BUF_LH_SF, 0, 0, stride), sample[c][2])


def test_download_strided_matches_full_download():
    flat = FlatForest.random(300, seed=21, max_arity=4, zero_frac=0.0)
    k = 7
    rng = np.random.default_rng(3)
    with hip.Engine(flat, 2, k) as eng:
        eng.set_models([(random_spec('F81', k, rng), (1.0, 0.0, 1.0)) for _ in range(2)])
        eng.set_masks(np.stack([random_masks(flat, k, rng) for _ in range(2)]))
        eng.bottom_up(True)
        post, lh_sum, lh_sf = eng.top_down_marginals()
        for col in (0, 1):
            for first, stride in ((0, 1), (3, 7), (flat.n_nodes - 1, 5)):
                rows = np.arange(first, flat.n_nodes, stride)
                assert np.array_equal(eng.download_strided(hip.BUF_POSTERIOR, col, first, stride), post[col][rows])
                assert np.array_equal(eng.download_strided(hip.BUF_LH_SUM, col, first, stride), lh_sum[col][rows])
                assert np.array_equal(eng.download_strided(hip.BUF_LH_SF, col, first, stride), lh_sf[col][rows])
        with pytest.raises(hip.HipError):
            eng.download_strided(hip.BUF_POSTERIOR, 0, 0, 1, flat.n_nodes + 1)
        eng.bottom_up(False)
        js = eng.joint_backtrace()
        assert np.array_equal(eng.download_strided(hip.BUF_JOINT_STATE, 1, 2, 3), js[1][2::3])


def test_cfg4_shape_matches_reference_sample():
    """BASELINE config 4's shape (F81, k=64, independent parameters per character) on 16 384 tips, 2 characters."""
    zs = [load_golden('synthetic_cfg4_L14_c{}'.format(c)) for c in (0, 1)]
    flat = synthetic.balanced_forest(int(zs[0]['n_levels']))
    with hip.Engine(flat, 2, 64) as eng:
        eng.set_models([golden_spec(z) for z in zs])
        eng.set_tip_states(np.stack([synthetic.tip_states(flat.n_tips, 64, c) for c in (0, 1)]))
        lnl = eng.bottom_up(True)
        post, lh_sum, lh_sf = eng.top_down_marginals()
    for c, z in enumerate(zs):
        s = z['sample']
        np.testing.assert_allclose(lnl[c], z['loglik'], rtol=LNL_RTOL)
        np.testing.assert_allclose(post[c][s], z['posterior'], rtol=POST_RTOL, atol=1e-300)


def test_cfg4_full_tree_invariants():
    """
    Config 4 at full tree size (1 048 576 tips, k=64, F81), 2 characters: properties that need no oracle --
    posteriors sum to one; every node sees the same total likelihood (pastml/ml.py:468-483), equal to the
    bottom-up log-likelihood; tips keep their observed state; a 4 096-tip subtree's bottom-up vectors equal the
    oracle's on that subtree.
    """
    L = 20
    flat = synthetic.balanced_forest(L)
    C, k = 2, 64
    states = np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)])
    specs = [dict(kind=0, pi=synthetic.f81_frequencies(k, c)) for c in range(C)]
    with hip.Engine(flat, C, k) as eng:
        eng.set_models([(s, (1.0, 0.0, 1.0)) for s in specs])
        eng.set_tip_states(states)
        lnl = eng.bottom_up(True)
        post, lh_sum, lh_sf = eng.top_down_marginals()
        sched = eng.schedule_info()
        # subtree rooted at the first node of depth 8 (id 255): its 4 096 tips
        sub_root = 255
        bu = eng.download(hip.BUF_BU, 0)
        bu_sf = eng.download(hip.BUF_BU_SF, 0)
    for c in range(C):
        np.testing.assert_allclose(post[c].sum(axis=1), 1, rtol=1e-12)
        np.testing.assert_allclose(np.log10(lh_sum[c]) - lh_sf[c], lnl[c] / np.log(10), rtol=1e-11)
        tip_post = post[c][flat.tips]
        assert np.array_equal(tip_post.argmax(axis=1), states[c])
        assert np.all(tip_post.max(axis=1) == 1.0)
    # subtree ids: level d of the subtree = ids [(sub_root+1) * 2^d - 1, ... + 2^d)
    ids = np.concatenate([np.arange((sub_root + 1) * (1 << d) - 1, (sub_root + 1) * (1 << d) - 1 + (1 << d))
                          for d in range(L - 8 + 1)])
    sub = synthetic.balanced_forest(L - 8)
    sub.dist[:] = flat.dist[ids]
    sub.dist[0] = 0
    masks = np.ones((sub.n_nodes, k), dtype=int)
    tip_pos = ids[sub.tips] - flat.tips[0]
    masks[sub.tips] = 0
    masks[sub.tips, states[0][tip_pos]] = 1
    r = orc.bottom_up(sub, masks, specs[0])
    assert_same_scaled(bu[ids], bu_sf[ids], r['bu'], r['bu_sf'], what='subtree BU')
    # ... and the numbers the REFERENCE produced for this very run (tests/golden/make_golden.py::case_cfg4_full: ml.py:82-148,
    # 240-290, 431-502 on the 1 048 576-tip tree, characters 0 and 1): ln L, and at every 4 099th node the posteriors, the
    # marginal likelihoods and the bottom-up vectors.  Default dispatch: the two-level and stacked units are what runs here.
    z = load_golden('synthetic_cfg4_full')
    assert sched[0] == 1 and sched[1] > 0 and sched[2] > 0, 'the two-level / stacked schedule was expected here: {}'.format(sched)
    for c in range(C):
        s = z['c{}_sample'.format(c)]
        np.testing.assert_allclose(lnl[c], z['c{}_loglik'.format(c)], rtol=LNL_RTOL)
        np.testing.assert_allclose(post[c][s], z['c{}_posterior'.format(c)], rtol=POST_RTOL, atol=1e-300)
        with np.errstate(divide='ignore'):
            np.testing.assert_allclose(np.log10(lh_sum[c][s]) - lh_sf[c][s],
                                       np.log10(z['c{}_lh'.format(c)].sum(axis=1)) - z['c{}_lh_sf'.format(c)],
                                       rtol=LNL_RTOL)  # (values of -2.7e6: an ulp is 4.7e-10)
    s = z['c0_sample']
    assert_same_scaled(bu[s], bu_sf[s], z['c0_bu'], z['c0_bu_sf'], what='BU at the sample, reference', ulps=4)


def test_cfg4_bench_characters_match_reference_on_4096_tips():
    """All 32 characters of the bench shard on a 4 096-tip tree against the reference's own run (synthetic_cfg4_full)."""
    z = load_golden('synthetic_cfg4_full')
    flat = synthetic.balanced_forest(int(z['small_n_levels']))
    C, k = 32, 64
    states = np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)])
    with hip.Engine(flat, C, k) as eng:
        eng.set_models([(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(C)])
        eng.set_tip_states(states)
        lnl = eng.bottom_up(True)
        post, lh_sum, lh_sf = eng.top_down_marginals()
    s = z['small_sample']
    np.testing.assert_allclose(lnl, z['small_loglik'], rtol=LNL_RTOL)
    np.testing.assert_allclose(post[:, s], z['small_posterior'], rtol=POST_RTOL, atol=1e-300)
    np.testing.assert_allclose(np.log10(lh_sum) - lh_sf, (lnl / np.log(10))[:, None] * np.ones(flat.n_nodes), rtol=1e-11)


@pytest.mark.parametrize('k', [4, 20, 32, 64, 130])
def test_cherry_fusion_is_bit_identical(k):
    """Recomputing cherries in registers (default) gives exactly the bits of the store-everything schedule."""
    rng = np.random.default_rng(k)
    flat = FlatForest.random(300, seed=k + 1, max_arity=3, zero_frac=0.0, n_trees=2)
    specs = [random_spec('F81', k, rng) for _ in range(2)]
    rates = [(1.3, 0.0, 1.0), (0.8, 0.02, 0.95)]
    masks = np.stack([random_masks(flat, k, rng) for _ in range(2)])
    out = []
    for fusion in (True, False):
        with hip.Engine(flat, 2, k, cherry_fusion=fusion) as eng:
            eng.set_models(list(zip(specs, rates)))
            eng.set_masks(masks)
            lnl = eng.bottom_up(True)
            post, lh_sum, lh_sf = eng.top_down_marginals()
            bu = eng.download(hip.BUF_BU, 1)
            bu_sf = eng.download(hip.BUF_BU_SF, 1)
            td_sf = (eng.download(hip.BUF_TD, 1), eng.download(hip.BUF_TD_SF, 1))
            # the joint sweep fuses cherries too (their tips' arg-max rows are written by the grandparent's unit)
            lnl_j = eng.bottom_up(False)
            tables = np.stack([eng.download(hip.BUF_JOINT_TABLE, c) for c in range(2)])[:, flat.parent >= 0]
            states = eng.joint_backtrace()
            bu_j = eng.download(hip.BUF_BU, 0)
            bu_j_sf = eng.download(hip.BUF_BU_SF, 0)
            out.append((lnl, post, lh_sum, lh_sf, bu, bu_sf, lnl_j, tables, states, bu_j, bu_j_sf, td_sf))
    for a, b in zip(out[0][:11], out[1][:11]):
        assert np.array_equal(a, b)
    # the cherries' top-down vectors are filled in on request in the fused run: same scales to rounding
    assert not np.isnan(out[0][11][0]).any() and not np.isnan(out[0][11][1]).any()
    assert_same_scaled(out[0][11][0], out[0][11][1], out[1][11][0], out[1][11][1], what='TD, fused vs stored cherries')


@pytest.mark.parametrize('k', [2, 5, 12, 64])
def test_single_launch_path_matches_level_path(k):
    """Small forests run a whole sweep in one launch; the level-per-launch schedule must give the same bits."""
    rng = np.random.default_rng(100 + k)
    flat = FlatForest.random(400, seed=k + 7, max_arity=4, zero_frac=0.0, n_trees=2)
    specs = [random_spec('F81', k, rng) for _ in range(3)]
    rates = [(1.1, 0.0, 1.0), (0.7, 0.03, 0.9), (2.0, 0.0, 1.0)]
    masks = np.stack([random_masks(flat, k, rng) for _ in range(3)])
    out = []
    for limit in (1000000, 0):
        with hip.Engine(flat, 3, k, tune=dict(SMALL_MAX_NODES=limit, BLOCK_NODES=0)) as eng:
            eng.set_models(list(zip(specs, rates)))
            eng.set_masks(masks)
            # the path under test is the one that runs: the whole sweep in one launch / level launches
            assert (eng.sweep_schedule()[0] == hip.SCHEDULE_SINGLE_LAUNCH) == bool(limit)
            lnl = eng.bottom_up(True)
            post, lh_sum, lh_sf = eng.top_down_marginals()
            # a second sweep with new parameters re-runs the fused prep
            eng.set_models(list(zip(specs[::-1], rates)))
            lnl2 = eng.bottom_up(True)
            out.append((lnl, post[:, flat.n_roots if False else 0:], lh_sf, lnl2))
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][3], out[1][3])
    assert np.array_equal(out[0][1], out[1][1])  # roots included: both schedules share the root unit
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-14, atol=0)   # roots: different butterfly shape
    assert np.array_equal(out[0][2], out[1][2])


@pytest.mark.parametrize('kind,k', [('F81', 2), ('F81', 5), ('F81', 20), ('F81', 64), ('F81', 130), ('EIGEN', 20)])
def test_device_state_selection_matches_host_rules(kind, k):
    """pml_select_states (MAP / MPPA, with and without force_joint and '.initial' masks) == the host restatement."""
    from pastml_amd import ml
    rng = np.random.default_rng(7 + k)
    flat = FlatForest.random(250, seed=3 * k, max_arity=3, n_trees=2)
    C = 2
    specs = [random_spec(kind, k, rng) for _ in range(C)]
    rates = [(1.0, 0.0, 1.0), (2.5, 0.0, 1.0)]
    masks = np.stack([random_masks(flat, k, rng, missing=0.3, multi=0.3) for _ in range(C)])
    lh_masks = np.ones_like(masks)
    pick = rng.random((C, flat.n_nodes)) < 0.1
    lh_masks[pick] = masks[pick]   # pretend these nodes were altered: their saved masks restrict the likelihoods
    with hip.Engine(flat, C, k) as eng:
        eng.set_models(list(zip(specs, rates)))
        eng.set_masks(masks)
        eng.bottom_up(False)
        js = eng.joint_backtrace()
        eng.bottom_up(True)
        post, lh_sum, _ = eng.top_down_marginals()
        for method, fj, lm in (('MAP', False, None), ('MPPA', False, None), ('MPPA', True, None),
                               ('MPPA', True, lh_masks), ('MAP', False, lh_masks)):
            sel, nsel = eng.select_states(method, force_joint=fj, lh_masks=lm)
            for c in range(C):
                lh = post[c] * (1 if lm is None else lm[c])
                if method == 'MAP':
                    ref, ref_k = ml.select_map(lh), np.ones(flat.n_nodes, dtype=int)
                else:
                    ref, ref_k = ml.select_mppa(lh, js[c].astype(np.int64) if fj else None)
                assert np.array_equal(sel[c], ref), (method, fj, lm is not None, c)
                assert np.array_equal(nsel[c], ref_k)
            # the selection became the columns' masks: the restricted sweep runs on it directly
            lnl = eng.bottom_up(True)
            eng.set_masks(sel)
            np.testing.assert_array_equal(lnl, eng.bottom_up(True))
            eng.set_masks(masks)


@pytest.mark.parametrize('k', [20, 40, 64])
def test_wide_polytomies_vs_oracle(k):
    """Arity up to 9: units with more children / tips per cherry than the lane-parallel gather holds take the
    sequential path inside the same launch; both paths must agree with the oracle."""
    rng = np.random.default_rng(k)
    flat = FlatForest.random(200, seed=50 + k, max_arity=9)
    spec = random_spec('F81', k, rng)
    masks = random_masks(flat, k, rng, internal=0.0)
    assert flat.n_children.max() > 4
    with hip.Engine(flat, 1, k) as eng:
        eng.set_models([(spec, (1.2, 0.0, 1.0))])
        eng.set_masks(masks)
        lnl = eng.bottom_up(True)
        post, lh_sum, lh_sf = eng.top_down_marginals()
        bu, bu_sf = eng.download(hip.BUF_BU), eng.download(hip.BUF_BU_SF)
    r = orc.full_marginal_pass(flat, masks.astype(int), spec, 1.2)
    np.testing.assert_allclose(lnl[0], r['loglik'], rtol=LNL_RTOL)
    assert_same_scaled(bu, bu_sf, r['bu'], r['bu_sf'], what='BU')
    np.testing.assert_allclose(post[0], r['posterior'], rtol=POST_RTOL, atol=1e-300)
    np.testing.assert_allclose(np.log10(lh_sum[0]) - lh_sf[0], r['loglik'] / np.log(10), rtol=1e-11)


@pytest.mark.parametrize('kind,k', [('HKY', 4), ('EIGEN', 5), ('EIGEN', 20)])
def test_matrix_models_observed_tips_on_zero_branches(kind, k):
    """
    Observed tips take a shortcut in the matrix-model bottom-up kernels (one row of P^T, arg-max in closed form).
    Zero-length tip branches are its corner: P(0) is the identity up to entries of +-1e-17 (generator.py:54-65 through
    the eigen-decomposition), so the message holds zeros / negative dust next to one 1 and numpy's first-maximum rule
    (ml.py:134) decides the arg-max table.  Compared with the oracle entry by entry.
    """
    rng = np.random.default_rng(77 + k)
    flat = FlatForest.random(60, seed=5 + k, max_arity=3)
    dist = flat.dist.copy()
    tips = np.flatnonzero(flat.is_tip)
    zero_tips = tips[::3]
    dist[zero_tips] = 0.0
    flat = _with_dist(flat, dist)
    spec = random_spec(kind, k, rng)
    masks = np.ones((flat.n_nodes, k), dtype=np.int8)
    states = rng.integers(0, k, size=flat.n_nodes)
    states[zero_tips[: len(zero_tips) // 2]] = 0  # state 0 has its own branch in the closed form
    for n in tips:
        masks[n] = 0
        masks[n, states[n]] = 1
    # two zero-length tips under one parent must agree, or the likelihood is zero (tested elsewhere)
    for p in np.unique(flat.parent[zero_tips]):
        kids = [c for c in zero_tips if flat.parent[c] == p]
        for c in kids[1:]:
            masks[c] = masks[kids[0]]
    with hip.Engine(flat, 1, k) as eng:
        eng.set_models([(spec, (1.0, 0.0, 1.0))])
        eng.set_masks(masks[None])
        lnl = eng.bottom_up(True)
        bu, bu_sf = eng.download(hip.BUF_BU, 0), eng.download(hip.BUF_BU_SF, 0)
        post, _, _ = eng.top_down_marginals()
        lnl_j = eng.bottom_up(False)
        table = eng.download(hip.BUF_JOINT_TABLE, 0)
        joint = eng.joint_backtrace()
    r = orc.full_marginal_pass(flat, masks.astype(int), spec, 1.0, 0.0, 1.0)
    np.testing.assert_allclose(lnl[0], r['loglik'], rtol=LNL_RTOL, atol=1e-12)
    # entries that are zero in exact arithmetic come out as 0 or +-1e-17 dust, in the reference as here, depending on
    # how P(0) rounds: compare every vector relative to its largest entry (1e-6 relative is the bar for the rest)
    la, lb = log_true(bu, bu_sf), log_true(r['bu'], r['bu_sf'])
    internal = ~flat.is_tip
    ma, mb = la[internal].max(axis=1), lb[internal].max(axis=1)
    np.testing.assert_allclose(ma, mb, rtol=0, atol=LOG10_ATOL)
    np.testing.assert_allclose(10 ** (la[internal] - ma[:, None]), 10 ** (lb[internal] - mb[:, None]), rtol=1e-9,
                               atol=1e-12)
    np.testing.assert_allclose(post[0], r['posterior'], rtol=POST_RTOL, atol=1e-12)
    j = orc.bottom_up(flat, masks.astype(int), spec, 1.0, 0.0, 1.0, is_marginal=False)
    np.testing.assert_allclose(lnl_j[0], j['loglik'], rtol=LNL_RTOL, atol=1e-12)
    nonroot = flat.parent >= 0
    for n, i in np.argwhere(table != j['joint_table']):
        if not nonroot[n]:
            continue
        # the device's P(t) differs from numpy's in the last bits: a flip is allowed only between products that are
        # equal to 1e-12 of the largest (for t = 0 that includes the +-1e-17 dust around the zeros)
        prod = orc.pij(spec, flat.dist[n], 1.0, 0.0, 1.0)[i] * j['bu'][n]
        assert abs(prod[table[n, i]] - prod.max()) <= 1e-12 * max(prod.max(), 1.0), (n, i)
    ref_states = orc.joint_backtrace(flat, j['bu'], j['joint_table'], spec['pi'])
    assert np.mean(joint[0] == ref_states) > 0.95  # ties on zero branches may pick the other of two equal states
    assert np.array_equal(joint[0][flat.is_tip], ref_states[flat.is_tip])


def _with_dist(flat, dist):
    """A copy of a FlatForest with other branch lengths (same topology)."""
    for node, d in zip(flat.nodes, dist):
        node.dist = float(d)
    roots = [n for n in flat.nodes if n.up is None]
    return FlatForest.from_trees(roots)


@pytest.mark.parametrize('k', [4, 20, 64])
def test_device_state_selection_with_tied_probabilities(k):
    """
    JC (equal frequencies) makes posteriors tie exactly: tips without data give uniform posteriors everywhere (every
    candidate of the MPPA walk ties, the best number of states is k), a tip pair with two different observed states
    gives a two-way tie at their parent.  The device selection must break the ties as the sort / argmin of
    ml.py:505-574 do; compared with the host restatement of those rules, with and without the joint state forced.
    """
    from pastml_amd import ml
    rng = np.random.default_rng(11 + k)
    flat = FlatForest.random(90, seed=k, max_arity=3)
    spec = dict(kind=0, pi=np.ones(k) / k)
    tips = np.flatnonzero(flat.is_tip)
    masks = np.ones((3, flat.n_nodes, k), dtype=np.int8)
    # column 0: no data at all; column 1: every tip observed with one of two states; column 2: half of the tips observed
    for col, frac in ((1, 1.0), (2, 0.5)):
        for n in tips:
            if rng.random() < frac:
                masks[col, n] = 0
                masks[col, n, rng.integers(2) if col == 1 else rng.integers(k)] = 1
    with hip.Engine(flat, 3, k) as eng:
        eng.set_models([(spec, (1.0, 0.0, 1.0))] * 3)
        eng.set_masks(masks)
        eng.bottom_up(False)
        js = eng.joint_backtrace()
        eng.bottom_up(True)
        post, _, _ = eng.top_down_marginals()
        assert np.all(post[0] == post[0][:, :1])  # uniform rows, exactly
        for fj in (False, True):
            sel, nsel = eng.select_states('MPPA', force_joint=fj)
            for c in range(3):
                ref, ref_k = ml.select_mppa(post[c], js[c].astype(np.int64) if fj else None)
                assert np.array_equal(nsel[c], ref_k), (fj, c)
                assert np.array_equal(sel[c], ref), (fj, c)
            assert np.all(nsel[0] == k)
            eng.set_masks(masks)
            eng.bottom_up(True)
            eng.top_down_marginals()


def test_keep_td_option_survives_tree_upload_and_is_not_sticky():
    """Engine(keep_td=True) stores the TD vectors in the first sweep (no second sweep on download); a download on an
    engine without the option repeats the sweep once and leaves the option off (pooled contexts stay lean)."""
    flat = synthetic.balanced_forest(12)   # above the single-launch size: level launches are counted
    k = 6
    spec = (dict(kind=0, pi=synthetic.f81_frequencies(k, 0)), (1.0, 0.0, 1.0))
    tds = []
    for keep in (True, False):
        with hip.Engine(flat, 1, k, keep_td=keep) as eng:
            eng.set_models([spec])
            eng.set_tip_states(synthetic.tip_states(flat.n_tips, k, 0))
            eng.profile_enable(True)
            eng.bottom_up(True)
            eng.top_down_marginals(posterior=False, lh=False)
            _, launches = eng.profile_read(1)
            tds.append(eng.download(hip.BUF_TD))
            _, after = eng.profile_read(1)
            assert (after == launches) == keep
            eng.download(hip.BUF_TD_SF)
            assert eng.profile_read(1)[1] == after          # materialised once
            eng.bottom_up(True)
            eng.top_down_marginals(posterior=False, lh=False)
            _, again = eng.profile_read(1)
            eng.download(hip.BUF_TD)
            assert (eng.profile_read(1)[1] == again) == keep  # without the option the next sweep is lean again
    assert np.array_equal(tds[0], tds[1], equal_nan=True)


def test_marginal_pass_equals_the_two_calls():
    """pml_marginal_pass = pml_bottom_up + pml_top_down_marginals with one host round trip: same bits, same errors."""
    rng = np.random.default_rng(11)
    for kind, k, tips in (('F81', 5, 300), ('F81', 64, 3000), ('HKY', 4, 200), ('EIGEN', 20, 500)):
        flat = FlatForest.random(tips, seed=k, max_arity=3, zero_frac=0.0)
        specs = [(random_spec(kind, k, rng), (1.2, 0.0, 1.0)) for _ in range(2)]
        masks = np.stack([random_masks(flat, k, rng) for _ in range(2)])
        with hip.Engine(flat, 2, k) as eng:
            eng.set_models(specs)
            eng.set_masks(masks)
            lnl = eng.bottom_up(True)
            post, lh_sum, lh_sf = eng.top_down_marginals()
            lnl2, post2, lh_sum2, lh_sf2 = eng.marginal_pass()
            assert np.array_equal(lnl, lnl2) and np.array_equal(post, post2)
            assert np.array_equal(lh_sum, lh_sum2) and np.array_equal(lh_sf, lh_sf2)
            # results stay valid for what follows a marginal pass
            assert eng.download(hip.BUF_POSTERIOR, 1).shape == (flat.n_nodes, k)
    # a column without likelihood: the error of pml_bottom_up, and no top-down results
    z = load_golden('edge_zero_likelihood')
    flat = golden_forest(z)
    spec, rates = golden_spec(z)
    with hip.Engine(flat, 1, 3) as eng:
        eng.set_models([(spec, rates)])
        eng.set_masks(z['masks'])
        with pytest.raises(hip.ZeroLikelihoodError):
            eng.marginal_pass()
        with pytest.raises(hip.HipError):
            eng.download(hip.BUF_POSTERIOR)


def test_joint_pass_equals_the_two_calls():
    """pml_joint_pass = joint pml_bottom_up + pml_joint_backtrace with one host round trip (back-trace replayed as a
    graph): same bits; on altered forests too."""
    rng = np.random.default_rng(12)
    for kind, k, tips, zero in (('F81', 5, 3000, 0.0), ('F81', 64, 3000, 0.0), ('EIGEN', 20, 3000, 0.0), ('HKY', 4, 500, 0.0)):
        flat = FlatForest.random(tips, seed=k + 3, max_arity=3, zero_frac=zero)
        specs = [(random_spec(kind, k, rng), (1.2, 0.0, 1.0)) for _ in range(2)]
        masks = np.stack([random_masks(flat, k, rng) for _ in range(2)])
        with hip.Engine(flat, 2, k) as eng:
            eng.set_models(specs)
            eng.set_masks(masks)
            lnl = eng.bottom_up(False)
            states = eng.joint_backtrace()
            for _ in range(2):   # the second call replays the captured graphs
                lnl2, states2 = eng.joint_pass()
                assert np.array_equal(lnl, lnl2) and np.array_equal(states, states2)
            assert np.array_equal(eng.download(hip.BUF_JOINT_STATE, 1), states[1])
    z = load_golden('edge_zero')
    flat = golden_forest(z)
    spec, rates = golden_spec(z)
    with hip.Engine(flat, 1, len(spec['pi'])) as eng:
        eng.set_models([(spec, rates)])
        eng.set_masks(z['masks_altered'])
        eng.set_initial_masks(z['masks_initial'])
        lnl, states = eng.joint_pass()
        np.testing.assert_allclose(lnl[0], z['loglik_joint'], rtol=LNL_RTOL)
        assert np.array_equal(states[0], z['joint_state'])


def test_many_columns_take_the_single_launch_kernels_with_the_same_bits():
    """Contexts of 64+ columns on forests of up to 16 384 nodes sweep in one launch (one workgroup per column walks all
    levels); a one-column context on the same forest (above the 2 048-node limit) takes the level kernels: same bits."""
    rng = np.random.default_rng(31)
    flat = FlatForest.random(1500, seed=9, max_arity=3, zero_frac=0.0)
    assert flat.n_nodes > 2048
    for k in (2, 12, 40):
        C = 66
        specs = [(random_spec('F81', k, rng), (float(rng.uniform(0.5, 3)), 0.0, 1.0)) for _ in range(C)]
        masks = np.stack([random_masks(flat, k, rng) for _ in range(C)])
        with hip.Engine(flat, C, k) as eng:
            eng.set_models(specs)
            eng.set_masks(masks)
            lnl, post, lh_sum, lh_sf = eng.marginal_pass()
        with hip.Engine(flat, 1, k) as eng:
            for c in (0, 17, 65):
                eng.set_models([specs[c]])
                eng.set_masks(masks[c])
                lnl1, post1, lh_sum1, lh_sf1 = eng.marginal_pass()
                assert lnl1[0] == lnl[c]
                assert np.array_equal(post1[0], post[c]) and np.array_equal(lh_sum1[0], lh_sum[c])
                assert np.array_equal(lh_sf1[0], lh_sf[c])


@pytest.mark.parametrize('n_tips,cols,k', [(300, 70, 2), (6000, 80, 2), (6000, 12, 4), (3000, 5, 12), (40000, 3, 64)])
def test_sweep_of_some_columns_leaves_the_others_alone(n_tips, cols, k):
    """pml_bottom_up_submit_columns: the columns named take part in the sweep and get the values of a sweep of all columns,
    the others keep the value of the last sweep that computed them (single-launch sweeps, subtree blocks, level kernels);
    a zero likelihood in a column that sits the sweep out is not reported; the next ordinary sweep computes everything."""
    rng = np.random.default_rng(n_tips + cols)
    flat = FlatForest.random(n_tips, seed=n_tips + 1, max_arity=3, zero_frac=0.0, n_trees=2)
    masks = np.stack([random_masks(flat, k, rng, internal=0.0) for _ in range(cols)])
    first = [(random_spec('F81', k, rng), (float(rng.uniform(0.5, 3)), 0.0, 1.0)) for _ in range(cols)]
    second = [(random_spec('F81', k, rng), (float(rng.uniform(0.5, 3)), 0.0, 1.0)) for _ in range(cols)]
    active = (rng.random(cols) < (0.25 if cols > 64 else 0.4)).astype(np.uint8)   # (80 columns: at most 32 active --
    active[0], active[-1] = 1, 0                                                   #  the few-column schedule of a wide context)
    with hip.Engine(flat, cols, k) as eng:
        eng.set_masks(masks)
        eng.set_models(first)
        before = eng.bottom_up(True)
        for rep in range(3):   # (the second and third time as replayed launch sequences)
            eng.set_models(first)
            eng.bottom_up(True)
            for c in np.flatnonzero(active):
                eng.set_models([second[c]], col_begin=int(c))
            eng.bottom_up_submit(True, active=active)
            mixed = eng.bottom_up_collect(True)
            bu_mixed = [eng.download(hip.BUF_BU, c) for c in (0, cols - 1)]
        eng.set_models(second)
        after = eng.bottom_up(True)
        bu_after = eng.download(hip.BUF_BU, 0)
        eng.set_models(first)
        assert np.array_equal(eng.bottom_up(True), before)
        bu_before = eng.download(hip.BUF_BU, cols - 1)
    if cols == 80:
        assert active.sum() <= 32
    assert np.array_equal(mixed, np.where(active == 1, after, before))
    assert np.array_equal(bu_mixed[0], bu_after) and np.array_equal(bu_mixed[1], bu_before)
    assert not np.array_equal(before, after)


@pytest.mark.parametrize('kind,k,n_tips,arity,n_trees', [('F81', 4, 3000, 2, 1), ('F81', 12, 900, 4, 3), ('F81', 64, 5000, 2, 1),
                                                         ('F81', 70, 600, 3, 2), ('HKY', 4, 700, 3, 1), ('EIGEN', 7, 500, 3, 2),
                                                         ('EIGEN', 20, 2500, 2, 1)])
def test_height_order_is_invisible_at_the_boundary(kind, k, n_tips, arity, n_trees):
    """
    pml_tree_upload renumbers a ragged forest into height order (pml_tree_order); every per-node array of the interface
    stays in the caller's numbering.  Against a context that keeps the caller's numbering (NO_HEIGHT_ORDER): the same bits
    in everything that crosses the boundary -- masks in (words and tip states), ln L, the whole-table outputs, every
    download (whole and strided), joint states and tables, the device selection with likelihood masks, the P(t) batch, the
    sampled scenario counts of pml_marginal_counts for a given seed.
    """
    rng = np.random.default_rng(k * 1000 + n_tips)
    flat = FlatForest.random(n_tips, seed=k + n_tips, max_arity=arity, zero_frac=0.0, n_trees=n_trees)
    cols = 3
    masks = np.stack([random_masks(flat, k, rng) for _ in range(cols)])
    lh_masks = (rng.random((cols, flat.n_nodes, k)) < 0.8).astype(np.int8)
    lh_masks[..., 0] = 1
    specs = [(random_spec(kind, k, rng), (float(rng.uniform(0.5, 3)), 0.0, 1.0)) for _ in range(cols)]
    tips = np.stack([rng.integers(0, k, size=flat.n_tips) for _ in range(cols)])
    out = {}
    # ('shaped': the numbering that also follows the shape-sorted lists -- large forests take it by themselves, round 6)
    orders = {}
    for name, tune in (('plain', dict(NO_HEIGHT_ORDER=1)), ('ordered', dict(SHAPE_ORDER=0)), ('shaped', dict(SHAPE_ORDER=1))):
        got = {}
        with hip.Engine(flat, cols, k, tune=tune) as eng:
            order = eng.node_order()
            orders[name] = order
            got_identity = np.array_equal(order, np.arange(flat.n_nodes))
            assert got_identity == (name == 'plain')
            assert np.array_equal(np.sort(order), np.arange(flat.n_nodes))
            eng.set_models(specs)
            eng.set_tip_states(tips)
            got['lnl_tips'] = eng.bottom_up(True)
            got['bu_tips'] = eng.download(hip.BUF_BU, 1)
            eng.set_masks(masks)
            lnl, post, lh_sum, lh_sf = eng.marginal_pass()
            got.update(lnl=lnl, post=post, lh_sum=lh_sum, lh_sf=lh_sf)
            # (the sampler's draws are keyed by the caller's node ids: the same scenarios whatever the library's numbering)
            got['counts'] = eng.marginal_counts(500, seed=17, col=1)
            for what in (hip.BUF_BU, hip.BUF_BU_SF, hip.BUF_TD, hip.BUF_TD_SF, hip.BUF_POSTERIOR, hip.BUF_LH_SUM, hip.BUF_LH_SF):
                got['dl%d' % what] = eng.download(what, 2)
            if kind == 'F81':
                got['exp'] = eng.download(hip.BUF_BRANCH_EXP, 0)
            for what in (hip.BUF_POSTERIOR, hip.BUF_LH_SUM, hip.BUF_LH_SF):
                got['st%d' % what] = eng.download_strided(what, 1, 3, 7)
                assert np.array_equal(got['st%d' % what], eng.download(what, 1)[3::7])
            jl, js = eng.joint_pass()
            got.update(jl=jl, js=js, jt=eng.download(hip.BUF_JOINT_TABLE, 0)[len(flat.roots):],   # (the roots' tables are undefined)
                       js_dl=eng.download(hip.BUF_JOINT_STATE, 2),
                       js_st=eng.download_strided(hip.BUF_JOINT_STATE, 2, 1, 5))
            eng.bottom_up(True)
            eng.top_down_marginals(posterior=False, lh=False)
            sel, nsel = eng.select_states('MPPA', force_joint=True, lh_masks=lh_masks)
            got.update(sel=sel, nsel=nsel)
            if kind != 'F81' and k <= 20:
                eng.set_models(specs)
                got['pij'] = eng.pij_batch(copy_out=True)
        out[name] = got
    for key, want in out['plain'].items():
        assert np.array_equal(want, out['ordered'][key], equal_nan=True), key
        assert np.array_equal(want, out['shaped'][key], equal_nan=True), key
    assert not np.array_equal(orders['ordered'], orders['shaped'])   # (they ARE two numberings)
    assert np.array_equal(out['ordered']['js'][2], out['ordered']['js_dl'])


def test_zero_likelihood_report_names_the_callers_nodes():
    """The ids of a PML_ZERO_LIKELIHOOD report are the caller's, whatever numbering the library works in."""
    flat = FlatForest.random(400, seed=5, max_arity=3, zero_frac=0.0, n_trees=1)
    k = 4
    # a zero-length branch between two nodes with different single states: the reference's zero-likelihood case
    child = int(flat.tips[37])
    par = int(flat.parent[child])
    dist = np.array(flat.dist)
    dist[child] = 0.0
    bad = FlatForest(flat.parent, flat.n_children, flat.first_child, dist, flat.roots)
    masks = np.ones((bad.n_nodes, k), dtype=np.int8)
    masks[child] = [1, 0, 0, 0]
    masks[par] = [0, 1, 0, 0]
    seen = []
    for tune in (dict(NO_HEIGHT_ORDER=1), {}):
        with hip.Engine(bad, 1, k, tune=tune) as eng:
            eng.set_models([(dict(kind=0, pi=np.ones(k) / k), (1.0, 0.0, 1.0))])
            eng.set_masks(masks)
            with pytest.raises(hip.ZeroLikelihoodError) as e:
                eng.bottom_up(True)
            seen.append((int(e.value.err_parent[0]), int(e.value.err_child[0])))
    assert seen[0] == seen[1] == (par, child)


@pytest.mark.parametrize('k,cols', [(20, 48), (64, 40)])
def test_partial_sweep_keeps_the_idle_columns_downloadable(k, cols):
    """
    A context whose full sweeps run the level schedule with two-level units (their children's vectors are never written)
    and whose few-column sweeps run the subtree blocks: after a sweep of a few columns, the bottom-up vectors of a column
    that sat it out must still be rebuilt for a download -- what the last FULL sweep left out of memory is still missing
    for that column (the "children absorbed" flag is sticky across partial sweeps).
    """
    flat = synthetic.balanced_forest(13)   # 8 192 tips: 4 095 stored nodes x 48 columns > 160 000 >= 4 095 x 32
    rng = np.random.default_rng(k)
    specs = [(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0 + 0.01 * c, 0.0, 1.0)) for c in range(cols)]
    other = [(dict(kind=0, pi=synthetic.f81_frequencies(k, 100 + c)), (1.3, 0.0, 1.0)) for c in range(cols)]
    states = np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(cols)])
    active = np.zeros(cols, dtype=np.uint8)
    active[rng.choice(cols - 1, size=5, replace=False)] = 1
    idle = cols - 1
    with hip.Engine(flat, cols, k) as eng:
        eng.set_tip_states(states)
        eng.set_models(specs)
        assert eng.sweep_schedule()[0] == hip.SCHEDULE_TWO_LEVEL
        before = eng.bottom_up(True)                    # a full sweep: the units' children are not in memory
        for c in np.flatnonzero(active):
            eng.set_models([other[c]], col_begin=int(c))
        eng.bottom_up_submit(True, active=active)
        mixed = eng.bottom_up_collect(True)
        got = eng.download(hip.BUF_BU, idle)            # (nothing has ever materialised the idle column's rows)
        got_sf = eng.download(hip.BUF_BU_SF, idle)
        first_active = int(np.flatnonzero(active)[0])
        got_active = eng.download(hip.BUF_BU, first_active)
        eng.set_models(other)
        after = eng.bottom_up(True)
        want_active = eng.download(hip.BUF_BU, first_active)
        eng.set_models(specs)
        assert np.array_equal(eng.bottom_up(True), before)
        want = eng.download(hip.BUF_BU, idle)
        want_sf = eng.download(hip.BUF_BU_SF, idle)
    assert np.array_equal(mixed, np.where(active == 1, after, before))
    assert np.array_equal(got, want) and np.array_equal(got_sf, want_sf)
    assert np.array_equal(got_active, want_active)
    assert np.isfinite(got).all() and (got.max(axis=1) > 0).all()


@pytest.mark.parametrize('kind,k,marginal', [('HKY', 4, True), ('EIG', 6, True), ('F81', 5, False)])
def test_column_flags_are_ignored_where_no_kernel_reads_them(kind, k, marginal):
    """pml_bottom_up_submit_columns outside the F81 marginal sweep (matrix / eigen models, the joint sweep): every column is
    computed, as the header says."""
    rng = np.random.default_rng(k)
    flat = FlatForest.random(500, seed=k, max_arity=3, zero_frac=0.0, n_trees=1)
    cols = 3
    masks = np.stack([random_masks(flat, k, rng, internal=0.0) for _ in range(cols)])
    first = [(random_spec(kind, k, rng), (float(rng.uniform(0.5, 3)), 0.0, 1.0)) for _ in range(cols)]
    second = [(random_spec(kind, k, rng), (float(rng.uniform(0.5, 3)), 0.0, 1.0)) for _ in range(cols)]
    with hip.Engine(flat, cols, k) as eng:
        eng.set_masks(masks)
        eng.set_models(first)
        eng.bottom_up(marginal)
        eng.set_models(second)
        eng.bottom_up_submit(marginal, active=np.array([1, 0, 0], dtype=np.uint8))
        flagged = eng.bottom_up_collect(marginal)
        assert np.array_equal(flagged, eng.bottom_up(marginal))


@pytest.mark.parametrize('n_tips,cols', [(300, 3), (6000, 2), (300, 70)])
def test_completion_word_wait_returns_what_the_stream_wait_returns(n_tips, cols):
    """Short sweeps of few columns end in a kernel that raises a word in pinned memory, and the host spins on it instead of
    synchronising the stream (pml_bottom_up / _collect, pml_marginal_pass without copies; NO_SPIN_WAIT = the stream): the
    results of a series of sweeps with changing models, of a sweep submitted twice before it is collected, and of passes
    with and without copies are the same either way (70 columns: no word, the stream)."""
    k = 4
    rng = np.random.default_rng(n_tips + cols)
    flat = FlatForest.random(n_tips, seed=n_tips, max_arity=3, zero_frac=0.0, n_trees=2)
    masks = np.stack([random_masks(flat, k, rng, internal=0.0) for _ in range(cols)])
    series = [[(random_spec('F81', k, rng), (float(rng.uniform(0.5, 3)), 0.0, 1.0)) for _ in range(cols)] for _ in range(6)]
    results = {}
    for name, tune in (('word', {}), ('stream', dict(NO_SPIN_WAIT=1))):
        out = []
        with hip.Engine(flat, cols, k, tune=tune) as eng:
            eng.set_masks(masks)
            for specs in series:
                eng.set_models(specs)
                out.append(eng.bottom_up(True))
                out.append(eng.marginal_pass(posterior=False, lh=False)[0])
                lnl, post, lh_sum, lh_sf = eng.marginal_pass()
                out += [lnl, post, lh_sum]
            # submitted twice, collected once: the second sweep's results
            eng.set_models(series[0])
            eng.bottom_up_submit(True)
            eng.set_models(series[1])
            eng.bottom_up_submit(True)
            out.append(eng.bottom_up_collect(True))
            eng.set_models(series[1])
            assert np.array_equal(out[-1], eng.bottom_up(True))
            # a pass without copies leaves the table where a later download finds it
            eng.set_models(series[2])
            eng.marginal_pass(posterior=False, lh=False)
            out.append(eng.download(hip.BUF_POSTERIOR, cols - 1))
        results[name] = out
    for x, y in zip(results['word'], results['stream']):
        assert np.array_equal(x, y)


@pytest.mark.parametrize('k', [2, 4, 12, 64])
def test_block_schedule_gives_the_bits_of_the_level_schedule(k):
    """Mid-size forests: subtree blocks walked by one workgroup each + the top above the cuts (a handful of launches)
    against one launch per level; blocks of 256 (default), 1024, 64 and 7 stored nodes; balanced and ragged forests."""
    rng = np.random.default_rng(40 + k)
    forests = [synthetic.balanced_forest(13),
               FlatForest.random(6000, seed=k + 2, max_arity=4, zero_frac=0.0, n_trees=3)]
    for flat in forests:
        C = 3
        specs = [(random_spec('F81', k, rng), (float(rng.uniform(0.5, 3)), 0.0, 1.0)) for _ in range(C)]
        masks = np.stack([random_masks(flat, k, rng, internal=0.0) for _ in range(C)])
        results = []
        n_blocks = []
        for block_nodes in (0, 256, 1024, 64, 7):
            with hip.Engine(flat, C, k, tune=dict(BLOCK_NODES=block_nodes, BLOCK_MAX_WORK=1 << 40)) as eng:
                eng.set_models(specs)
                eng.set_masks(masks)
                kind, nb, _ = eng.sweep_schedule()   # the schedule under test is the one that runs
                assert (kind == hip.SCHEDULE_BLOCKS) == (block_nodes != 0), (block_nodes, kind)
                n_blocks.append(nb)
                lnl, post, lh_sum, lh_sf = eng.marginal_pass()
                again = eng.bottom_up(True)            # graph replay of the same schedule
                bu = eng.download(hip.BUF_BU, 1)
                assert np.array_equal(lnl, again)
            results.append((lnl, post, lh_sum, lh_sf, bu))
        assert n_blocks[0] == 0 and n_blocks[4] > n_blocks[3] > n_blocks[1] >= n_blocks[2] > 0, n_blocks
        for other in results[1:]:
            for a, b in zip(results[0], other):
                assert np.array_equal(a, b)


@pytest.mark.parametrize('k', [3, 11, 20, 32])
def test_eigen_joint_sweep_wide_polytomies_and_deep_trees(k):
    """
    The vector-unit joint sweep of the eigen models (pml_kernels_eigen_joint.h) where its shortcuts end: nodes with
    more than 14 children (the unit descriptor counts up to 14, the kernel reads the true number), unobserved and
    ambiguous tips (the general pass instead of the one-column closed form), several columns, and a caterpillar deep
    enough to drive the products out of the rescaling band.  ln L, tables and states against the oracle.
    """
    rng = np.random.default_rng(300 + k)
    wide = FlatForest.random(150, seed=70 + k, max_arity=24)
    assert wide.n_children.max() > 14
    from pastml_amd.tree import TreeNode
    root = TreeNode(name='r', dist=0.0)
    cur = root
    for d in range(700):   # every level multiplies by a tip's message (<= max P ~ 0.1-0.5): far below 2^-200 in total
        cur.add_child(name='t{}'.format(d), dist=0.8)
        cur = cur.add_child(name='i{}'.format(d), dist=0.05)
    cur.add_child(name='ta', dist=0.1)
    cur.add_child(name='tb', dist=0.1)
    deep = FlatForest.from_trees([root])
    for flat in (wide, deep):
        C = 2
        specs = [random_spec('EIGEN', k, rng) for _ in range(C)]
        rates = [(float(rng.uniform(0.5, 3)), 0.0, 1.0), (1.0, 0.02, 0.9)]
        masks = np.stack([random_masks(flat, k, rng, missing=0.1, multi=0.1, internal=0.03) for _ in range(C)])
        with hip.Engine(flat, C, k) as eng:
            eng.set_models(list(zip(specs, rates)))
            eng.set_masks(masks)
            lnl_j = eng.bottom_up(False)
            tables = [eng.download(hip.BUF_JOINT_TABLE, c) for c in range(C)]
            states = eng.joint_backtrace()
        nonroot = flat.parent >= 0
        for c in range(C):
            j = orc.bottom_up(flat, masks[c].astype(int), specs[c], *rates[c], is_marginal=False)
            np.testing.assert_allclose(lnl_j[c], j['loglik'], rtol=LNL_RTOL, atol=1e-12)
            flips = 0
            for n, i in np.argwhere(tables[c] != j['joint_table']):
                if not nonroot[n]:
                    continue
                prod = orc.pij(specs[c], flat.dist[n], *rates[c])[i] * j['bu'][n]
                assert abs(prod[tables[c][n, i]] - prod.max()) <= 1e-12 * prod.max()
                flips += 1
            if flips == 0:
                assert np.array_equal(states[c], orc.joint_backtrace(flat, j['bu'], j['joint_table'], specs[c]['pi']))


@pytest.mark.parametrize('k', [5, 20])
def test_eigen_joint_vector_and_matrix_core_sweeps_agree(k, monkeypatch):
    """Both joint sweeps of the eigen models (vector FMAs; materialised / matrix-core P) give the same ln L to rounding
    and the same reconstruction on a random forest."""
    import subprocess
    import sys
    import json
    code = '''
import json, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from pastml_amd import hip
from pastml_amd.tree import FlatForest
from test_gpu_parity import random_spec, random_masks
k = {k}
rng = np.random.default_rng(77)
flat = FlatForest.random(300, seed=9, max_arity=5, n_trees=2)
spec = random_spec('EIGEN', k, rng)
masks = random_masks(flat, k, rng)
with hip.Engine(flat, 1, k) as eng:
    eng.set_models([(spec, (1.3, 0.0, 1.0))])
    eng.set_masks(masks[None])
    lnl = eng.bottom_up(False)
    states = eng.joint_backtrace()
print(json.dumps(dict(lnl=float(lnl[0]), states=states[0].tolist())))
'''.format(root=REPO, tests=os.path.dirname(os.path.abspath(__file__)), k=k)
    out = []
    for off in (False, True):
        env = dict(os.environ)
        if off:
            env['PASTML_HIP_NO_EIGEN_JOINT_VALU'] = '1'
        res = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        out.append(json.loads(res.stdout.strip().splitlines()[-1]))
    np.testing.assert_allclose(out[0]['lnl'], out[1]['lnl'], rtol=1e-12)
    assert out[0]['states'] == out[1]['states']


@pytest.mark.parametrize('k', [2, 4, 7])
def test_staged_posterior_stores_give_the_same_bits(k):
    """Top-down LEVEL kernels of narrow units (k <= 8) write their posteriors through LDS slots (observed tips as
    (id, state) pairs); the subtree-block kernels of the same lane shape write them straight from the units.  Same unit
    functions, same tables, bit for bit -- on a forest with polytomies, unobserved and ambiguous tips and restricted
    internal nodes, big enough for level launches."""
    rng = np.random.default_rng(11)
    flat = FlatForest.random(40000, seed=4, max_arity=5, n_trees=2)
    C = 2
    specs = [(random_spec('F81', k, rng), (1.3, 0.0, 1.0)) for _ in range(C)]
    masks = np.stack([random_masks(flat, k, rng) for _ in range(C)])
    out = []
    for tune, want in ((dict(BLOCK_NODES=0, SMALL_MANY_NODES=0, NO_THIN=1), hip.SCHEDULE_LEVELS),
                       (dict(SMALL_MANY_NODES=0, BLOCK_MAX_WORK=1 << 30), hip.SCHEDULE_BLOCKS)):
        with hip.Engine(flat, C, k, tune=tune) as eng:
            eng.set_models(specs)
            eng.set_masks(masks)
            assert eng.sweep_schedule()[0] == want
            out.append(eng.marginal_pass())
    assert np.isfinite(out[0][1]).all()
    for x, y in zip(out[0], out[1]):
        assert np.array_equal(x, y)


def test_hky_sweeps_with_p_in_registers_give_the_bits_of_the_materialised_batch():
    """HKY: the sweeps build P(t) of a branch from the closed form in registers (PML_P_HKY); PASTML_HIP_NO_HKY_FUSED=1
    reads the materialised batch.  Same closed form, same order of additions: identical posteriors, ln L, tables."""
    import json
    import subprocess
    import sys
    code = '''
import hashlib, json, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from pastml_amd import hip
from pastml_amd.tree import FlatForest
from test_gpu_parity import random_spec, random_masks
rng = np.random.default_rng(21)
flat = FlatForest.random(5000, seed=8, max_arity=4, n_trees=2)
C = 3
specs = [(random_spec('HKY', 4, rng), (float(rng.uniform(0.5, 2)), 0.0, 1.0)) for _ in range(C)]
masks = np.stack([random_masks(flat, 4, rng) for _ in range(C)])
with hip.Engine(flat, C, 4) as eng:
    eng.set_models(specs)
    eng.set_masks(masks)
    lnl, post, lh_sum, lh_sf = eng.marginal_pass()
    lnl_j, states = eng.joint_pass()
    tables = [eng.download(hip.BUF_JOINT_TABLE, c) for c in range(C)]
h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
print(json.dumps(dict(lnl=h(lnl), post=h(post), lh_sum=h(lh_sum), lh_sf=h(lh_sf), lnl_j=h(lnl_j), states=h(states),
                      tables=h(np.stack(tables)), finite=bool(np.isfinite(post).all()))))
'''.format(root=REPO, tests=os.path.dirname(os.path.abspath(__file__)))
    out = []
    for off in (False, True):
        env = dict(os.environ)
        if off:
            env['PASTML_HIP_NO_HKY_FUSED'] = '1'
        res = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        out.append(json.loads(res.stdout.strip().splitlines()[-1]))
    assert out[0]['finite']
    assert out[0] == out[1]


@pytest.mark.parametrize('k', [2, 4, 12, 64])
def test_implicit_tip_posteriors_give_the_same_table(k):
    """
    PML_OPT_IMPLICIT_TIP_POSTERIORS: the top-down sweep leaves the unit-vector rows of observed tips unwritten (a third
    of its traffic on a binary tree) and they are written when somebody reads the table, or before the masks that define
    them change.  Every reader must see exactly the table of the default sweep: full and strided downloads, the
    posterior_out of the sweep call, the state selection that follows.
    """
    rng = np.random.default_rng(300 + k)
    flat = FlatForest.random(500, seed=k + 11, max_arity=3, zero_frac=0.0, n_trees=2)
    specs = [random_spec('F81', k, rng) for _ in range(3)]
    rates = [(1.2, 0.0, 1.0), (0.6, 0.02, 0.9), (2.5, 0.0, 1.0)]
    masks = np.stack([random_masks(flat, k, rng) for _ in range(3)])   # observed, ambiguous and unknown tips
    out = []
    for implicit in (False, True):
        with hip.Engine(flat, 3, k) as eng:
            eng.set_option(hip.OPT_IMPLICIT_TIP_POSTERIORS, implicit)
            eng.set_models(list(zip(specs, rates)))
            eng.set_masks(masks)
            lnl, _, lh_sum, lh_sf = eng.marginal_pass(posterior=False, lh=True)
            strided = eng.download_strided(hip.BUF_POSTERIOR, 1, 3, 7)
            full = np.stack([eng.download(hip.BUF_POSTERIOR, c) for c in range(3)])
            # a second pass, its table through the call itself
            eng.set_models(list(zip(specs[::-1], rates)))
            lnl2, post2, _, _ = eng.marginal_pass(posterior=True, lh=False)
            # ... and one read only by the selection (which rewrites the masks the implicit rows are defined by)
            eng.marginal_pass(posterior=False, lh=False)
            sel, nsel = eng.select_states('MAP')
            after = np.stack([eng.download(hip.BUF_POSTERIOR, c) for c in range(3)]) if False else None
            out.append((lnl, lh_sum, lh_sf, strided, full, lnl2, post2, sel, nsel))
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b, equal_nan=True)
    assert np.array_equal(out[0][4][1][3::7], out[0][3])


@pytest.mark.parametrize('k', [32, 64])
def test_full_mask_pair_waves_match_the_oracle(k):
    """
    The cherry level's specialised bodies (bu_f81_marg_body FULL / PAIR: k = lanes x states per lane, every mask of the
    wave's units and cherries full, every unit two cherries of two tips) run on whole waves of a balanced tree with
    observed tips -- and must not run when one unit of the wave has a restricted internal node or a tip without data.
    Both against the oracle, and the second forest against itself with the restrictions lifted one at a time.
    """
    flat = synthetic.balanced_forest(9)
    rng = np.random.default_rng(k)
    spec = random_spec('F81', k, rng)
    states = synthetic.tip_states(flat.n_tips, k, 3)
    plain = synthetic.one_hot_masks(flat, k, states).astype(int)
    mixed = plain.copy()
    tips = np.flatnonzero(flat.is_tip)
    internal = np.flatnonzero(~flat.is_tip)
    mixed[tips[5]] = 1                                  # a tip without data
    mixed[tips[40], :] = 0
    mixed[tips[40], [1, 3]] = 1                         # an ambiguous tip
    mixed[internal[-7], : k // 2] = 0                   # a restricted cherry
    mixed[internal[len(internal) // 2], 1::2] = 0       # a restricted node higher up
    for masks in (plain, mixed):
        with hip.Engine(flat, 1, k) as eng:
            eng.set_models([(spec, (1.4, 0.0, 1.0))])
            eng.set_masks(masks[None])
            lnl = eng.bottom_up(True)
            post, lh_sum, lh_sf = eng.top_down_marginals()
            bu = eng.download(hip.BUF_BU, 0)
            bu_sf = eng.download(hip.BUF_BU_SF, 0)
        ref = orc.full_marginal_pass(flat, masks, spec, sf=1.4)
        np.testing.assert_allclose(lnl[0], ref['loglik'], rtol=LNL_RTOL)
        np.testing.assert_allclose(post[0], ref['posterior'], rtol=1e-9, atol=1e-300)
        assert_same_scaled(bu[internal], bu_sf[internal], ref['bu'][internal], ref['bu_sf'][internal], what='BU')


def _forest_with_balanced_clumps(n_leaves, seed, clump_frac=0.5):
    """A random binary forest where a share of the leaves is replaced by perfectly balanced 8-tip subtrees (the shape
    the two-level units take: a node with two children that each carry two cherries of two tips)."""
    from pastml_amd.tree import TreeNode
    rng = np.random.default_rng(seed)
    roots = []
    for _ in range(2):
        root = TreeNode(name='', dist=0.0)
        leaves = [root]
        while len(leaves) < n_leaves // 2:
            leaf = leaves.pop(int(rng.integers(len(leaves))))
            for _c in range(2 if rng.random() < 0.9 else 3):
                leaves.append(leaf.add_child(dist=float(rng.uniform(0.001, 0.3))))
        for leaf in leaves:
            if rng.random() < clump_frac:
                level = [leaf]
                for _d in range(3):
                    level = [n.add_child(dist=float(rng.uniform(0.001, 0.3))) for n in level for _c in range(2)]
        roots.append(root)
    for ti, root in enumerate(roots):
        for i, n in enumerate(root.traverse('preorder')):
            n.name = 't{}_{}'.format(ti, i) if n.is_leaf() else 'n{}_{}'.format(ti, i)
    return FlatForest.from_trees(roots)


@pytest.mark.parametrize('k', [17, 20, 29, 32, 40, 64])
def test_two_level_units_give_the_bits_of_the_level_schedule(k):
    """
    Level schedule of large forests, lane groups of 8 and more: nodes with two stored children that each carry two
    cherries of two tips run as two-level units (their children's bottom-up vectors are never written, their posterior
    rows are not read back).  Against the same library with PASTML_HIP_NO_SUPER=1 -- bit for bit: ln L, posteriors,
    sums, scales, the bottom-up vectors a download materialises, the top-down vectors of PML_OPT_KEEP_TD -- on a
    balanced tree (every node of the two levels is taken over) and on a ragged forest with balanced clumps (both kinds
    of units, thin rest levels), with unobserved and ambiguous tips and restricted internal nodes (the bodies with masks)
    and with every state allowed (the straight-line bodies); k = 17, 20, 29, 40 have padding states.
    """
    rng = np.random.default_rng(900 + k)
    forests = [synthetic.balanced_forest(12), _forest_with_balanced_clumps(700, seed=k)]
    for fi, flat in enumerate(forests):
        C = 3
        specs = [(random_spec('F81', k, rng), (float(rng.uniform(0.5, 3)), 0.0, 1.0)) for _ in range(C)]
        masks = np.stack([random_masks(flat, k, rng, missing=0.05, multi=0.05, internal=0.02) for _ in range(C)])
        # column 0: observed tips, every internal state allowed (the full-mask bodies wherever a wave is uniform)
        masks[0] = synthetic.one_hot_masks(flat, k, rng.integers(0, k, size=flat.n_tips))
        results = []
        for no_super in (True, False):
            with hip.Engine(flat, C, k, tune=dict(LEVEL_SCHEDULE, NO_SUPER=1 if no_super else None)) as eng:
                eng.set_models(specs)
                eng.set_masks(masks)
                # the schedule under test is the one that runs: two-level / stacked units scheduled and launched, or none
                on, n_two, n_stack = eng.schedule_info()
                assert on == (not no_super) and (n_two > 0 and n_stack > 0) == (not no_super), (no_super, on, n_two, n_stack)
                eng.profile_enable(True)
                lnl, post, lh_sum, lh_sf = eng.marginal_pass()
                assert (eng.profile_read(3)[1] > 0) == (not no_super) and (eng.profile_read(4)[1] > 0) == (not no_super)
                eng.profile_enable(False)
                again = eng.bottom_up(True)
                assert np.array_equal(lnl, again)
                bu = [eng.download(hip.BUF_BU, c) for c in range(C)]
                bu_sf = [eng.download(hip.BUF_BU_SF, c) for c in range(C)]
                post2, lh_sum2, lh_sf2 = eng.top_down_marginals()    # after the download's materialisation
                assert np.array_equal(post, post2) and np.array_equal(lh_sum, lh_sum2)
                td = eng.download(hip.BUF_TD, 1)
                td_sf = eng.download(hip.BUF_TD_SF, 1)
            results.append((lnl, post, lh_sum, lh_sf, np.stack(bu), np.stack(bu_sf), td, td_sf))
        assert np.isfinite(results[0][1]).all()
        for a, b in zip(results[0], results[1]):
            assert np.array_equal(a, b), 'forest {}'.format(fi)
        # and the numbers are right: column 1 against the oracle
        ref = orc.bottom_up(flat, masks[1].astype(int), specs[1][0], *specs[1][1])
        np.testing.assert_allclose(results[1][0][1], ref['loglik'], rtol=LNL_RTOL)


def test_two_level_units_report_the_reference_pair_on_zero_likelihood():
    """A child of a two-level node whose vector comes out all zero: the unit falls back to the sequential path, which
    names the pair the reference would (the same pair as without two-level units)."""
    k = 64
    flat = synthetic.balanced_forest(12)
    rng = np.random.default_rng(5)
    spec = (random_spec('F81', k, rng), (1.0, 0.0, 1.0))
    masks = synthetic.one_hot_masks(flat, k, rng.integers(0, k, size=flat.n_tips))
    bad = masks.copy()
    # an internal node two levels above the tips that allows a single state none of whose ... no: make it impossible
    # outright: its mask is empty
    node = int(flat.parent[flat.parent[flat.tips[777]]])
    bad[node] = 0
    out = []
    for no_super in (True, False):
        with hip.Engine(flat, 2, k, tune=dict(LEVEL_SCHEDULE, NO_SUPER=1 if no_super else None)) as eng:
            eng.set_models([spec, spec])
            eng.set_masks(np.stack([masks, bad]))
            assert eng.schedule_info()[0] == (not no_super)
            with pytest.raises(hip.ZeroLikelihoodError) as e:
                eng.bottom_up(True)
            out.append((int(e.value.err_parent[0]), int(e.value.err_child[0]), int(e.value.err_parent[1]),
                        int(e.value.err_child[1])))
    assert out[0][:2] == (-1, -1)
    assert out[0][2] == node
    assert out[0] == out[1]


def test_eigen_joint_sweep_follows_the_masks_through_graph_replay():
    """The joint sweep of the eigen models leaves out the launch for unobserved tips when every column's masks came from
    tip states; its launch sequence is replayed as a graph.  Masks that change between calls -- observed tips, then some
    unobserved and ambiguous ones, then observed again -- must change the sequence with them: every result equals a fresh
    engine's."""
    k = 20
    rng = np.random.default_rng(77)
    flat = FlatForest.random(3000, seed=5, max_arity=3, n_trees=2)
    spec = (random_spec('EIGEN', k, rng), (1.2, 0.0, 1.0))
    states = rng.integers(0, k, size=flat.n_tips).astype(np.int32)
    loose = random_masks(flat, k, rng, missing=0.2, multi=0.2, internal=0.0)

    def fresh(setter):
        with hip.Engine(flat, 1, k) as e:
            e.set_models([spec])
            setter(e)
            lnl = e.bottom_up(False)
            return lnl, e.download(hip.BUF_JOINT_TABLE, 0), e.joint_backtrace()

    ref_obs = fresh(lambda e: e.set_tip_states(states))
    ref_loose = fresh(lambda e: e.set_masks(loose[None]))
    with hip.Engine(flat, 1, k) as eng:
        eng.set_models([spec])
        for setter, ref in ((lambda: eng.set_tip_states(states), ref_obs), (lambda: eng.set_masks(loose[None]), ref_loose),
                            (lambda: eng.set_tip_states(states), ref_obs)):
            setter()
            for _ in range(3):   # capture, then replays
                lnl = eng.bottom_up(False)
                assert np.array_equal(lnl, ref[0])
                assert np.array_equal(eng.download(hip.BUF_JOINT_TABLE, 0), ref[1])
                assert np.array_equal(eng.joint_backtrace(), ref[2])


@pytest.mark.parametrize('k', [5, 20, 32])
def test_eigen_tiers_give_the_bits_of_one_launch_per_level(k):
    """Eigen models: the thin levels of the joint sweep and of the marginal bottom-up sweep run in tiers of subtree blocks
    (one launch per tier); PASTML_HIP_NO_EIGJ_TIERS=1 gives every level its own launch.  Same passes over the same nodes:
    ln L, arg-max tables, joint states, posteriors bit for bit -- balanced tree, ragged forest, several columns, tips of
    every kind."""
    rng = np.random.default_rng(600 + k)
    for flat in (synthetic.balanced_forest(11), FlatForest.random(5000, seed=k, max_arity=4, n_trees=3)):
        C = 2
        specs = [(random_spec('EIGEN', k, rng), (float(rng.uniform(0.5, 2)), 0.0, 1.0)) for _ in range(C)]
        masks = np.stack([random_masks(flat, k, rng, missing=0.1, multi=0.1, internal=0.02) for _ in range(C)])
        out = []
        launches = []
        for off in (True, False):
            with hip.Engine(flat, C, k, tune=dict(NO_EIGJ_TIERS=1 if off else None)) as eng:
                eng.set_models(specs)
                eng.set_masks(masks)
                eng.profile_enable(True)
                eng.bottom_up(False)
                launches.append(eng.profile_read(0)[1])
                eng.profile_enable(False)
                lnl_j = eng.bottom_up(False)
                tables = [eng.download(hip.BUF_JOINT_TABLE, c) for c in range(C)]
                states = eng.joint_backtrace()
                lnl, post, lh_sum, lh_sf = eng.marginal_pass()
            out.append((lnl_j, np.stack(tables), states, lnl, post, lh_sum, lh_sf))
        assert launches[1] < launches[0], launches   # the tiers ran: fewer launches than one per level
        for a, b in zip(out[0], out[1]):
            assert np.array_equal(a, b)


@pytest.mark.parametrize('k', [29, 33, 64])
def test_two_level_units_when_roots_are_two_level_nodes(k):
    """Forests of tiny balanced trees: a root with two children of two cherries each IS a two-level node -- its row comes from
    the roots' launch, its own vector from the two-level launch, and the rest lists are empty or nearly so.  Against the
    oracle."""
    from pastml_amd.tree import TreeNode
    rng = np.random.default_rng(k)

    def forest_of(levels):
        roots = []
        for n_levels in levels:
            root = TreeNode(name='', dist=0.0)
            level = [root]
            for _ in range(n_levels):
                level = [n.add_child(dist=float(rng.uniform(0.01, 0.3))) for n in level for _c in range(2)]
            roots.append(root)
        for ti, root in enumerate(roots):
            for i, n in enumerate(root.traverse('preorder')):
                n.name = 't{}_{}'.format(ti, i) if n.is_leaf() else 'n{}_{}'.format(ti, i)
        return FlatForest.from_trees(roots)

    for levels in ([3], [3, 3, 3], [3, 4, 2], [5]):
        flat = forest_of(levels)
        C = 2
        specs = [(random_spec('F81', k, rng), (1.3, 0.0, 1.0)) for _ in range(C)]
        masks = np.stack([random_masks(flat, k, rng, missing=0.1, multi=0.1, internal=0.0) for _ in range(C)])
        with hip.Engine(flat, C, k, tune=dict(LEVEL_SCHEDULE, SMALL_MAX_NODES=0)) as eng:
            eng.set_models(specs)
            eng.set_masks(masks)
            eng.profile_enable(True)
            lnl, post, lh_sum, lh_sf = eng.marginal_pass()
            assert eng.profile_read(3)[1] > 0 and eng.profile_read(4)[1] > 0
        for c in range(C):
            r = orc.full_marginal_pass(flat, masks[c].astype(int), specs[c][0], *specs[c][1])
            np.testing.assert_allclose(lnl[c], r['loglik'], rtol=LNL_RTOL)
            np.testing.assert_allclose(post[c], r['posterior'], rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize('k', [2, 4, 7, 12, 16, 20, 33, 64])
def test_thin_ends_as_subtree_blocks_have_the_bits_of_the_level_launches(k):
    """
    Thin ends of a large ragged forest (pml_tree_upload, "thin ends"): the high fused levels in tiers of subtree blocks (a
    launch per tier, several small subtrees per workgroup) + the narrow end's launch, the deep depths as bins of subtrees
    in one launch -- against the same library with NO_THIN (a launch per level), bit for bit: ln L, posteriors, sums,
    scales, the bottom-up vectors a download materialises, a partial sweep; fewer launches; and against the oracle.
    THIN_UNITS / THIN_BLOCK_NODES are scaled down so that forests of a few thousand tips have thin ends of several tiers
    and bins; k > 16: units of 8 lanes (the level schedule without two-level units, which has no thin ends of its own).
    """
    rng = np.random.default_rng(2300 + k)
    forests = [FlatForest.random(6000, seed=k, max_arity=2, n_trees=1), FlatForest.random(5000, seed=k + 1, max_arity=4, n_trees=3)]
    base = dict(BLOCK_NODES=0, SMALL_MAX_NODES=0, SMALL_MANY_NODES=0, THIN_UNITS=400, THIN_BLOCK_NODES=24, NARROW_UNITS=8,
                NO_SUPER=1 if k > 16 else None)
    for fi, flat in enumerate(forests):
        C = 3
        specs = [(random_spec('F81', k, rng), (float(rng.uniform(0.5, 3)), 0.0, 1.0)) for _ in range(C)]
        masks = np.stack([random_masks(flat, k, rng, missing=0.05, multi=0.05, internal=0.02) for _ in range(C)])
        masks[0] = synthetic.one_hot_masks(flat, k, rng.integers(0, k, size=flat.n_tips))
        results, launches = [], []
        for off in (True, False):
            with hip.Engine(flat, C, k, tune=dict(base, NO_THIN=1 if off else None), keep_td=(k % 2 == 0)) as eng:
                eng.set_models(specs)
                eng.set_masks(masks)
                eng.profile_enable(True)
                lnl, post, lh_sum, lh_sf = eng.marginal_pass()
                launches.append((eng.profile_read(0)[1], eng.profile_read(1)[1]))
                eng.profile_enable(False)
                assert np.array_equal(lnl, eng.bottom_up(True))
                lnl2, post2, lh_sum2, lh_sf2 = eng.marginal_pass()    # (the captured graphs)
                assert np.array_equal(lnl, lnl2) and np.array_equal(post, post2)
                bu = [eng.download(hip.BUF_BU, c) for c in range(C)]
                bu_sf = [eng.download(hip.BUF_BU_SF, c) for c in range(C)]
                if k % 2 == 0:   # (the top-down vectors of PML_OPT_KEEP_TD, written by the deep end's launch as well)
                    bu.append(eng.download(hip.BUF_TD, 1))
                    bu_sf.append(eng.download(hip.BUF_TD_SF, 1))
                # a sweep of one column only
                specs2 = list(specs)
                specs2[1] = (specs[1][0], (specs[1][1][0] * 1.25, 0.0, 1.0))
                eng.set_models(specs2)
                eng.bottom_up_submit(True, active=np.array([0, 1, 0], dtype=np.uint8))
                part = np.asarray(eng.bottom_up_collect(True))
            results.append((lnl, post, lh_sum, lh_sf, np.stack(bu), np.stack(bu_sf), part))
        assert launches[1][0] < launches[0][0] and launches[1][1] < launches[0][1], launches
        assert np.isfinite(results[0][1]).all()
        for a, b in zip(results[0], results[1]):
            assert np.array_equal(a, b), 'forest {}'.format(fi)
        ref = orc.bottom_up(flat, masks[1].astype(int), specs[1][0], *specs[1][1])
        np.testing.assert_allclose(results[1][0][1], ref['loglik'], rtol=LNL_RTOL)


@pytest.mark.parametrize('k', [17, 20, 32, 33, 48, 64])
def test_lane_shapes_follow_the_forests_arity(k):
    """
    The lane shape of the F81 kernels is a function of k and the forest (pml_chars_alloc), never of the columns.  32 < k <= 64:
    the top-down kernels take 8 states per lane on (mostly) binary forests and 4 where many nodes have three or four children
    (16 lanes gather four children in parallel, 8 lanes two); the bottom-up levels take 8 only on forests with balanced parts.
    16 < k <= 32: 8 lanes x 4 states, on forests with polytomies 16 x 2.  The default's bits are those of the explicit shape
    (F81_R / F81_TD_R / BU_WIDE) it should have taken, whatever the number of columns; the other shape rounds differently
    somewhere but agrees to 1e-11; and the numbers are right against the oracle.
    """
    rng = np.random.default_rng(3100 + k)
    wide = k > 32
    for flat, poly, balanced in ((FlatForest.random(1500, seed=k, max_arity=2, n_trees=1), False, False),
                                 (synthetic.balanced_forest(9), False, True),
                                 (FlatForest.random(1500, seed=k + 1, max_arity=3, n_trees=2), True, False),
                                 (FlatForest.random(1500, seed=k + 2, max_arity=6, n_trees=1), True, False)):
        if wide:
            explicit = dict(F81_TD_R=4 if poly else 8, BU_WIDE=1 if balanced else 0)
            other = dict(F81_TD_R=8 if poly else 4, BU_WIDE=0 if balanced else 1)   # (both sweeps in the shape not taken)
        else:
            explicit = dict(F81_R=2, F81_TD_R=2) if poly else {}
            other = {} if poly else dict(F81_R=2, F81_TD_R=2)
        specs = [(random_spec('F81', k, rng), (float(rng.uniform(0.5, 3)), 0.0, 1.0)) for _ in range(3)]
        masks = np.stack([random_masks(flat, k, rng, missing=0.05, multi=0.05, internal=0.02) for _ in range(3)])
        out = {}
        for name, C, tune in (('default', 3, {}), ('one column', 1, {}), ('explicit', 3, explicit), ('other', 3, other)):
            if name == 'other' and not wide and poly:
                tune = dict(F81_R=4, F81_TD_R=4)
            with hip.Engine(flat, C, k, tune=tune) as eng:
                eng.set_models(specs[:C])
                eng.set_masks(masks[:C])
                out[name] = eng.marginal_pass()
        for a, b in zip(out['default'], out['explicit']):
            assert np.array_equal(a, b)
        for a, b in zip(out['default'], out['one column']):
            assert np.array_equal(a[:1], b)
        assert not np.array_equal(out['default'][1], out['other'][1])   # (the other shape rounds differently somewhere)
        np.testing.assert_allclose(out['default'][1], out['other'][1], rtol=1e-11, atol=1e-300)
        ref = orc.bottom_up(flat, masks[1].astype(int), specs[1][0], *specs[1][1])
        np.testing.assert_allclose(out['default'][0][1], ref['loglik'], rtol=LNL_RTOL)


@pytest.mark.parametrize('k', [65, 67, 128, 130, 256])
def test_lean_units_with_masks_of_several_words_have_the_sequential_paths_bits(k):
    """
    More than 64 states: masks of several words.  Units of at most two children (cherries of at most two tips) issue every
    load up front and take the mask bits of the lane's own state pairs (bu_f81_unit_lean_w / td_f81_unit_lean_w) instead of
    walking children and tips one round trip at a time -- against the sequential path (NO_WIDE_LEAN), bit for bit: ln L,
    posteriors, sums, scales, the bottom-up and top-down vectors; binary, balanced and polytomy forests (there most units
    stay on the sequential path), observed, ambiguous and unobserved tips, restricted internal nodes; and against the oracle.
    """
    rng = np.random.default_rng(4100 + k)
    forests = [FlatForest.random(1200, seed=k, max_arity=2, n_trees=2), synthetic.balanced_forest(8),
               FlatForest.random(900, seed=k + 1, max_arity=4, n_trees=1)]
    for fi, flat in enumerate(forests):
        C = 3
        specs = [(random_spec('F81', k, rng), (float(rng.uniform(0.5, 3)), 0.0, 1.0)) for _ in range(C)]
        masks = np.stack([random_masks(flat, k, rng, missing=0.05, multi=0.05, internal=0.02) for _ in range(C)])
        masks[0] = synthetic.one_hot_masks(flat, k, rng.integers(0, k, size=flat.n_tips))
        results = []
        for off in (True, False):
            with hip.Engine(flat, C, k, tune=dict(NO_WIDE_LEAN=1 if off else None), keep_td=True) as eng:
                eng.set_models(specs)
                eng.set_masks(masks)
                lnl, post, lh_sum, lh_sf = eng.marginal_pass()
                bu = np.stack([eng.download(hip.BUF_BU, c) for c in range(C)])
                bu_sf = np.stack([eng.download(hip.BUF_BU_SF, c) for c in range(C)])
                td = eng.download(hip.BUF_TD, 1)
                td_sf = eng.download(hip.BUF_TD_SF, 1)
                assert np.array_equal(lnl, eng.bottom_up(True))
            results.append((lnl, post, lh_sum, lh_sf, bu, bu_sf, td, td_sf))
        assert np.isfinite(results[0][1]).all()
        for a, b in zip(results[0], results[1]):
            assert np.array_equal(a, b), 'forest {}'.format(fi)
        ref = orc.bottom_up(flat, masks[1].astype(int), specs[1][0], *specs[1][1])
        np.testing.assert_allclose(results[1][0][1], ref['loglik'], rtol=LNL_RTOL)

@pytest.mark.parametrize('k,n_tips,arity', [(257, 300, 2), (300, 900, 4), (400, 2500, 2), (512, 600, 3)])
def test_more_than_256_states_match_the_oracle(k, n_tips, arity):
    """
    257 - 512 states (round 6; the F81 family): one wavefront per unit with 8 states per lane, masks of five to eight words,
    16-bit arg-max tables, the plain level schedule -- against the oracle (the reference has no bound on k, pastml/ml.py:134):
    ln L, the bottom-up and top-down vectors, the marginal pass, the joint sweep with its tables and back-trace; MAP / MPPA
    selection against the host rules; the fused pass against the separate calls, bit for bit.
    """
    from pastml_amd import ml
    rng = np.random.default_rng(k)
    flat = FlatForest.random(n_tips, seed=k, max_arity=arity, n_trees=2)
    C = 2
    specs = [random_spec('F81', k, rng) for _ in range(C)]
    rates = [(float(rng.uniform(0.5, 2)), 0.0, 1.0), (float(rng.uniform(0.5, 2)), 0.02, 0.9)]
    masks = np.stack([random_masks(flat, k, rng, missing=0.1, multi=0.1, internal=0.02) for _ in range(C)])
    with hip.Engine(flat, C, k, keep_td=True) as eng:
        eng.set_models(list(zip(specs, rates)))
        assert eng.sweep_schedule()[0] == hip.SCHEDULE_LEVELS
        eng.set_masks(masks)
        lnl = eng.bottom_up(True)
        post, lh_sum, lh_sf = eng.top_down_marginals()
        bus = [(eng.download(hip.BUF_BU, c), eng.download(hip.BUF_BU_SF, c)) for c in range(C)]
        tds = [(eng.download(hip.BUF_TD, c), eng.download(hip.BUF_TD_SF, c)) for c in range(C)]
        fused = eng.marginal_pass()
        for a, b in zip(fused, (lnl, post, lh_sum, lh_sf)):
            assert np.array_equal(a, b)
        lnl_j = eng.bottom_up(False)
        tables = [eng.download(hip.BUF_JOINT_TABLE, c) for c in range(C)]
        states = eng.joint_backtrace()
        lnl_j2, states2 = eng.joint_pass()
        assert np.array_equal(lnl_j, lnl_j2) and np.array_equal(states, states2)
        eng.bottom_up(True)
        eng.top_down_marginals()
        for method, fj in (('MAP', False), ('MPPA', False), ('MPPA', True)):
            sel, nsel = eng.select_states(method, force_joint=fj)
            for c in range(C):
                if method == 'MAP':
                    ref, ref_k = ml.select_map(post[c]), np.ones(flat.n_nodes, dtype=int)
                else:
                    ref, ref_k = ml.select_mppa(post[c], states[c].astype(np.int64) if fj else None)
                assert np.array_equal(sel[c], ref), (method, fj, c)
                assert np.array_equal(nsel[c], ref_k)
            eng.set_masks(masks)
    nonroot = flat.parent >= 0
    for c in range(C):
        r = orc.full_marginal_pass(flat, masks[c].astype(int), specs[c], *rates[c])
        np.testing.assert_allclose(lnl[c], r['loglik'], rtol=LNL_RTOL, atol=1e-12)
        assert_same_scaled(bus[c][0], bus[c][1], r['bu'], r['bu_sf'], what='BU col {}'.format(c))
        assert_same_scaled(tds[c][0], tds[c][1], r['td'], r['td_sf'], rows=nonroot, what='TD col {}'.format(c))
        np.testing.assert_allclose(post[c], r['posterior'], rtol=POST_RTOL, atol=1e-300)
        tot = np.log10(lh_sum[c]) - lh_sf[c]
        np.testing.assert_allclose(tot, r['loglik_per_tree'][flat.tree_id] / np.log(10), rtol=1e-11, atol=1e-12)
        j = orc.bottom_up(flat, masks[c].astype(int), specs[c], *rates[c], is_marginal=False)
        np.testing.assert_allclose(lnl_j[c], j['loglik'], rtol=LNL_RTOL, atol=1e-12)
        assert np.array_equal(tables[c][nonroot], j['joint_table'][nonroot])
        assert np.array_equal(states[c], orc.joint_backtrace(flat, j['bu'], j['joint_table'], specs[c]['pi']))


@pytest.mark.parametrize('k', [2, 4, 7, 12, 16])
def test_lean_units_for_polytomies_have_the_sequential_paths_bits(k):
    """
    The kernels that walk several levels in one launch (small forests, subtree blocks, the thin ends of large ones), units of
    fewer than 8 lanes: a unit of three or four children (cherries of up to four tips) takes its children two at a time, every
    load of a pair before any of its values (bu_f81_unit_lean_poly / td_f81_unit_lean_poly), where no unit of its wavefront
    needs the sequential path -- against that path (NO_WIDE_LEAN), bit for bit: ln L, posteriors, sums, scales, the bottom-up
    and top-down vectors; single-launch and block schedules, at most 3 / 4 / 6 children per node; and against the oracle.
    """
    rng = np.random.default_rng(4300 + k)
    forests = [FlatForest.random(400, seed=k, max_arity=3, n_trees=1), FlatForest.random(3000, seed=k + 1, max_arity=4, n_trees=2),
               FlatForest.random(2500, seed=k + 2, max_arity=6, n_trees=1), FlatForest.random(2000, seed=k + 3, max_arity=3, n_trees=1)]
    for fi, flat in enumerate(forests):
        C = 3
        specs = [(random_spec('F81', k, rng), (float(rng.uniform(0.5, 3)), 0.0, 1.0)) for _ in range(C)]
        masks = np.stack([random_masks(flat, k, rng, missing=0.05, multi=0.05, internal=0.02) for _ in range(C)])
        masks[0] = synthetic.one_hot_masks(flat, k, rng.integers(0, k, size=flat.n_tips))
        results = []
        for off in (True, False):
            with hip.Engine(flat, C, k, tune=dict(NO_WIDE_LEAN=1 if off else None), keep_td=(fi % 2 == 0)) as eng:
                eng.set_models(specs)
                eng.set_masks(masks)
                assert eng.sweep_schedule()[0] in (hip.SCHEDULE_SINGLE_LAUNCH, hip.SCHEDULE_BLOCKS)
                lnl, post, lh_sum, lh_sf = eng.marginal_pass()
                bu = np.stack([eng.download(hip.BUF_BU, c) for c in range(C)])
                bu_sf = np.stack([eng.download(hip.BUF_BU_SF, c) for c in range(C)])
                extra = (eng.download(hip.BUF_TD, 1), eng.download(hip.BUF_TD_SF, 1)) if fi % 2 == 0 else ()
                assert np.array_equal(lnl, eng.bottom_up(True))
            results.append((lnl, post, lh_sum, lh_sf, bu, bu_sf) + extra)
        assert np.isfinite(results[0][1]).all()
        for a, b in zip(results[0], results[1]):
            assert np.array_equal(a, b), 'forest {}'.format(fi)
        ref = orc.bottom_up(flat, masks[1].astype(int), specs[1][0], *specs[1][1])
        np.testing.assert_allclose(results[1][0][1], ref['loglik'], rtol=LNL_RTOL)
