"""
CPU-only tests of the host-side logic of the boundary (no sweep is computed here): forest statistics, observed
frequencies, masks and their zero-branch alteration, MAP / MPPA selection including the reference's stateful
'.initial' behaviour -- all against golden vectors produced by the real reference.
"""
import os

import numpy as np
import pandas as pd
import pytest

from conftest import load_golden, golden_forest, GOLDEN
from pastml_amd import ml
from pastml_amd.acr import calculate_observed_freqs
from pastml_amd.annotation import ForestStats, preannotate_forest
from pastml_amd.tree import read_tree, FlatForest, TreeNode, name_tree

DATA = os.path.join(GOLDEN, 'data')


def albania():
    tree = read_tree(os.path.join(DATA, 'Albanian.tree.152tax.tre'))
    df = pd.read_csv(os.path.join(DATA, 'data.txt'), index_col=0, header=0)[['Country']]
    return tree, df


def test_newick_and_flat_forest_match_reference_tree():
    tree, _ = albania()
    z = load_golden('albania_F81')
    flat = FlatForest.from_trees([tree])
    assert flat.n_nodes == 305 and flat.n_tips == 154 and flat.n_bu_levels == 25
    assert np.array_equal(flat.parent, z['parent'])
    assert np.array_equal(flat.first_child[~flat.is_tip], z['first_child'][~flat.is_tip])
    assert np.array_equal(flat.dist, z['dist'])
    assert [n.name for n in flat.nodes] == list(z['node_names'])
    # traversal orders of the container
    assert [n.name for n in tree.traverse()] == list(z['node_names'])
    post = [n for n in tree.traverse('postorder')]
    ids = {id(n): i for i, n in enumerate(flat.nodes)}
    assert [ids[id(n)] for n in post] == list(flat.postorder_ids())
    # newick round trip
    again = read_tree(tree.write())
    assert [n.name for n in again.traverse()] == list(z['node_names'])
    assert np.array_equal(FlatForest.from_trees([again]).dist, flat.dist)


def test_forest_stats_and_observed_frequencies():
    tree, df = albania()
    z = load_golden('albania_F81')
    preannotate_forest([tree], df=df)
    fs = ForestStats([tree])
    assert fs.avg_nonzero_brlen == float(z['fs_avg_nonzero_brlen'])
    assert fs.forest_length == float(z['fs_forest_length'])
    assert (fs.num_nodes, fs.num_tips) == (int(z['fs_num_nodes']), int(z['fs_num_tips']))
    states = z['opt_states']
    _, obs, _ = calculate_observed_freqs('Country', [tree], states)
    assert np.array_equal(obs, z['observed_frequencies'])


def test_initialize_and_alter_albania():
    tree, df = albania()
    z = load_golden('albania_F81')
    preannotate_forest([tree], df=df)
    p = ml.ForestProblem([tree], 'Country', z['opt_states'])
    p.initialize_allowed_states()
    assert np.array_equal(p.masks, z['fix_masks_initial'])
    assert np.array_equal(p.annotated, z['annotation'].any(axis=1))
    altered = p.alter_zero_node_allowed_states()
    assert sorted(altered.tolist()) == list(z['fix_altered_nodes'])
    assert np.array_equal(p.masks, z['fix_masks_altered'])
    p.unalter_zero_node_allowed_states(altered)
    assert np.array_equal(p.masks, z['fix_masks_initial'])
    # the reference-style feature API agrees
    ml.initialize_allowed_states(tree, 'Country', z['opt_states'])
    got = np.array([getattr(n, 'Country_ALLOWED_STATES') for n in p.nodes])
    assert np.array_equal(got, z['fix_masks_initial'])


CASES = [('albania_F81', 'fix_'), ('albania_JC', 'fix_'), ('albania_EFT', 'fix_'), ('albania_F81', 'tau_'),
         ('edge_poly', ''), ('edge_zero', ''), ('edge_zero_tau', ''), ('edge_forest', ''),
         ('synthetic_f81_k5_L9', ''), ('synthetic_jtt_k20_L8', ''), ('synthetic_f81_k64_L8', '')]


@pytest.mark.parametrize('name,prefix', CASES)
def test_selection_pipeline_matches_reference(name, prefix):
    """
    Replays the mask bookkeeping of ml_acr (ml.py:697-748) on the reference's likelihood arrays: alteration, MAP,
    the alteration performed by the restricted-MAP sweep, MPPA and its statistics.
    """
    z = load_golden(name)
    g = lambda key: z[prefix + key]
    flat = golden_forest(z)
    k = g('masks_initial').shape[1]
    p = ml.ForestProblem([], 'ch', np.arange(k), flat=flat)
    p.masks = g('masks_initial').copy()
    p.annotated = z['annotation'].any(axis=1) if 'annotation' in z else ~np.all(g('masks_initial') == 1, axis=1)
    tau = float(z[{'fix_': 'opt_', 'tau_': 'tau_', '': ''}[prefix] + 'tau'])
    altered = p.alter_zero_node_allowed_states() if tau == 0 else np.zeros(0, dtype=int)
    assert sorted(altered.tolist()) == list(g('altered_nodes'))
    assert np.array_equal(p.masks, g('masks_altered'))
    p.unalter_zero_node_allowed_states(altered)
    lh = g('lh').copy()
    lh[p.has_init] *= p.init_masks[p.has_init]
    p.masks = ml.select_map(lh)
    assert np.array_equal(p.masks, g('masks_map'))
    # restricted-MAP sweep with alter=True
    if tau == 0:
        a2 = p.alter_zero_node_allowed_states()
        p.unalter_zero_node_allowed_states(a2)
        assert np.array_equal(p.masks, g('masks_map'))
    lh[p.has_init] *= p.init_masks[p.has_init]
    masks, best_k = ml.select_mppa(lh, g('joint_state') if bool(g('force_joint')) else None)
    assert np.array_equal(masks, g('masks_mppa'))
    assert int((best_k > 1).sum()) == int(g('mppa_num_unresolved'))
    assert int(best_k.sum()) == int(g('mppa_num_states'))
    np.testing.assert_allclose(np.log(best_k.astype(float)).sum(), float(g('mppa_log_num_scenarios')), rtol=1e-12)


def test_select_mppa_agrees_with_per_node_rule():
    """Vectorised MPPA == the per-node restatement in the oracle on random likelihoods with ties."""
    from oracle import pastml_oracle as orc
    rng = np.random.default_rng(3)
    for k in (2, 3, 7, 20):
        lh = rng.dirichlet(np.ones(k) * 0.3, size=400)
        lh[::7] = np.round(lh[::7], 1) + 1e-3  # ties
        lh[::11, 0] = 0
        js = rng.integers(0, k, size=len(lh))
        for joint in (None, js):
            a, ak = ml.select_mppa(lh, joint)
            b, bk = orc.choose_mppa(lh, joint)
            assert np.array_equal(a, b) and np.array_equal(ak, bk)


def test_models_parameter_vectors():
    """Parameter vector / bounds layout of the host models (models/__init__.py:145-181, 311-363; HKYModel.py:84-130)."""
    from pastml_amd.models.F81Model import F81Model
    from pastml_amd.models.JCModel import JCModel
    from pastml_amd.models.HKYModel import HKYModel
    from pastml_amd.models.JTTModel import JTTModel, JTT_RATE_MATRIX

    class FS:
        forest_length, num_nodes, num_tips, avg_nonzero_brlen = 10., 21, 11, 0.5

    m = F81Model(states=np.array(list('cab')), forest_stats=FS(), frequencies=np.array([.2, .3, .5]))
    assert list(m.states) == ['a', 'b', 'c']
    assert m.sf == 2.0 and m.tau == 0
    assert m.get_num_params() == 3
    np.testing.assert_allclose(m.get_optimised_parameters(), [2.0, .4, .6])
    np.testing.assert_allclose(m.get_bounds(), [[0.002, 20.], [1e-6, 1e7], [1e-6, 1e7]])
    m.fix_extra_params()
    assert m.get_num_params() == 3 and len(m.get_optimised_parameters()) == 1 and m.extra_params_fixed()
    m.unfix_extra_params()
    m.set_params_from_optimised(np.array([3., 1., 2.]))
    np.testing.assert_allclose(m.frequencies, [.25, .5, .25])
    assert m.sf == 3.
    np.testing.assert_allclose(m.get_mu(), 1 / (1 - .375))
    m.freeze()
    assert m.get_num_params() == 0
    with pytest.raises(NotImplementedError):
        m.sf = 1.
    jc = JCModel(states=np.array(list('abcd')), forest_stats=FS(), parameter_file={'scaling_factor': 1.5})
    assert jc.sf == 1.5 and jc.get_num_params() == 0 and jc.basic_params_fixed()
    jc2 = JCModel(states=np.array(list('abcd')), forest_stats=FS(), tau=0.1, optimise_tau=True)
    assert jc2.get_num_params() == 2
    np.testing.assert_allclose(jc2._tau_factor, 10. / (10. + 0.1 * 20))
    hky = HKYModel(forest_stats=FS(), parameter_file={'kappa': 2., 'A': .1, 'C': .2, 'G': .3, 'T': .4})
    assert hky.kappa == 2. and not hky._optimise_kappa and not hky._optimise_frequencies
    assert hky.get_num_params() == 1
    hky2 = HKYModel(forest_stats=FS())
    assert hky2.get_num_params() == 5
    np.testing.assert_allclose(hky2.get_bounds()[-1], [1e-6, 20.])
    jtt = JTTModel(forest_stats=FS())
    assert jtt.get_num_params() == 1 and np.array_equal(JTT_RATE_MATRIX, JTT_RATE_MATRIX.T)
    spec = jtt.kernel_spec()
    np.testing.assert_allclose(spec['A'].dot(np.diag(spec['d'])).dot(spec['Ainv']).sum(axis=1), 0, atol=1e-12)


def test_name_tree():
    t = read_tree('((a:1,b:1):1,(c:1,:1):1);')
    name_tree(t)
    names = [n.name for n in t.traverse('preorder')]
    assert len(set(names)) == len(names) and names[0] == 'root' and all(names)


def _leaf_distances(tree):
    """{leaf name: {other leaf name: path length}} by walking to the root (small trees only)."""
    depth = {}
    for n in tree.traverse('preorder'):
        depth[id(n)] = (depth[id(n.up)] + n.dist) if n.up is not None else 0.0
    leaves = list(tree)
    anc = {id(l): [] for l in leaves}
    for l in leaves:
        n = l
        while n is not None:
            anc[id(l)].append(n)
            n = n.up
    out = {}
    for a in sorted(leaves, key=lambda l: l.name)[:12]:
        on_path = {id(x) for x in anc[id(a)]}
        for b in leaves:
            lca = next(x for x in anc[id(b)] if id(x) in on_path)
            out[(a.name, b.name)] = depth[id(a)] + depth[id(b)] - 2 * depth[id(lca)]
    return out


@pytest.mark.parametrize('seed', range(6))
def test_set_outgroup_keeps_the_unrooted_tree(seed):
    """Re-rooting (used by the re-rooting invariance test of the reference) changes the root, not the tree."""
    tree, _ = albania()
    before = _leaf_distances(tree)
    total = sum(n.dist for n in tree.traverse() if n.up is not None)
    rng = np.random.default_rng(seed)
    candidates = [n for n in tree.traverse() if n.up is not None]
    target = candidates[rng.integers(len(candidates))]
    half = target.dist / 2
    assert tree.set_outgroup(target) is tree
    assert tree.children[0] is target and len(tree.children) == 2
    assert target.dist == pytest.approx(half) and tree.children[1].dist == pytest.approx(half)
    for n in tree.traverse():
        for c in n.children:
            assert c.up is n
    after = _leaf_distances(tree)
    assert set(before) == set(after)
    for key, d in before.items():
        assert after[key] == pytest.approx(d, abs=1e-12)
    assert sum(n.dist for n in tree.traverse() if n.up is not None) == pytest.approx(total, abs=1e-12)


def test_generator_lbfgsb_reproduces_scipy_minimize():
    """
    batch.lbfgsb_steps (scipy's reverse-communication routine driven from a generator, so that all characters of a
    group advance in one loop) asks for the points scipy.optimize.minimize(method='L-BFGS-B', jac=True) asks for, in
    its order, and ends at its result.
    """
    from scipy.optimize import minimize
    from pastml_amd import batch as B
    if B._setulb is None:
        pytest.skip('this SciPy build has another reverse-communication routine: the thread driver is used')

    def fg(x):
        f = np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2) + 0.1 * np.sum(np.sin(3 * x))
        g = 0.3 * np.cos(3 * x)
        g[:-1] += -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
        g[1:] += 200 * (x[1:] - x[:-1] ** 2)
        return f, g

    rng = np.random.default_rng(0)
    for trial in range(12):
        n = int(rng.integers(2, 9))
        bounds = np.stack([rng.uniform(-2, 0, n), rng.uniform(0.5, 3, n)], axis=1)
        x0 = rng.uniform(-3, 3, n)   # (also outside the bounds: both clip it)
        asked_scipy = []

        def recorded(x):
            asked_scipy.append(x.copy())
            return fg(x)
        ref = minimize(recorded, x0, method='L-BFGS-B', bounds=bounds, jac=True)
        asked = []
        steps = B.lbfgsb_steps(x0, bounds)
        try:
            point = next(steps)
            while True:
                asked.append(point.copy())
                point = steps.send(fg(point))
        except StopIteration as stop:
            found = stop.value
        assert len(asked) == len(asked_scipy) and all(np.array_equal(a, b) for a, b in zip(asked, asked_scipy))
        assert np.array_equal(found.x, ref.x) and found.fun == ref.fun and found.success == ref.success
        assert found.nit == ref.nit and found.nfev == ref.nfev


def test_continued_lbfgsb_run_is_the_run_with_the_tighter_tolerance():
    """
    A search that ends on L-BFGS-B's relative-reduction test is continued once at a tenfold tighter ftol
    (batch.CONTINUE_FTOL_FACTOR; searches of at least batch.CONTINUE_MIN_PARAMETERS parameters): the continued run asks for
    the points scipy's minimize asks for when it is given the tighter ftol from the start, and ends where that run ends --
    never above where the default run ends.
    """
    from scipy.optimize import minimize
    from pastml_amd import batch as B
    if B._setulb is None:
        pytest.skip('this SciPy build has another reverse-communication routine')

    def fg(x):
        f = np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
        g = np.zeros_like(x)
        g[:-1] += -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
        g[1:] += 200 * (x[1:] - x[:-1] ** 2)
        return f, g

    n = 30
    bounds = np.stack([np.full(n, -5.0), np.full(n, 5.0)], axis=1)
    seen = 0
    for seed in range(6):
        x0 = np.random.default_rng(seed).uniform(-2, 2, n)
        plain = minimize(fg, x0, method='L-BFGS-B', bounds=bounds, jac=True)
        asked_tight = []

        def recorded(x):
            asked_tight.append(x.copy())
            return fg(x)
        tight = minimize(recorded, x0, method='L-BFGS-B', bounds=bounds, jac=True,
                         options=dict(ftol=2.220446049250313e-09 * B.CONTINUE_FTOL_FACTOR))
        asked = []
        steps = B.lbfgsb_steps(x0, bounds, continue_factor=B.CONTINUE_FTOL_FACTOR)
        try:
            point = next(steps)
            while True:
                asked.append(point.copy())
                point = steps.send(fg(point))
        except StopIteration as stop:
            found = stop.value
        if 'RELATIVE REDUCTION' not in plain.message:
            assert found.continued_at is None and np.array_equal(found.x, plain.x)
            continue
        seen += 1
        assert found.continued_at == (plain.nit, plain.fun)
        assert len(asked) == len(asked_tight) and all(np.array_equal(a, b) for a, b in zip(asked, asked_tight))
        assert np.array_equal(found.x, tight.x) and found.fun == tight.fun and found.fun <= plain.fun
    assert seen >= 2


def test_tau_none_frees_tau_for_the_first_ml_character_only(monkeypatch):
    """
    acr(tau=None) -- the pipeline's smoothing=True -- as in the reference (pastml/acr.py:185-187, inside its loop over
    the characters): the first maximum-likelihood character optimises tau, the argument is then 0 for the later ones.
    """
    from pastml_amd import acr as acr_module

    class Planned(Exception):
        pass

    seen = {}

    def fake_run_tasks(forest, tasks, **kwargs):
        seen['tau_free'] = [bool(t.model._optimise_tau) for t in tasks]
        seen['tau'] = [t.model.tau for t in tasks]
        raise Planned()

    from pastml_amd import batch as batch_module
    monkeypatch.setattr(batch_module, 'run_tasks', fake_run_tasks)   # (acr imports it when called)
    tree = read_tree(os.path.join(GOLDEN, 'data', 'Albanian.tree.152tax.tre'))
    tips = [t.name for t in tree]
    df = pd.DataFrame({c: ['x' if i % 3 else 'y' for i in range(len(tips))] for c in ('a', 'b', 'c')}, index=tips)
    for tau, reoptimise, expect in ((None, False, [True, False, False]), (0, False, [False] * 3),
                                    (None, True, [True] * 3), (0.01, False, [False] * 3)):
        with pytest.raises(Planned):
            acr_module.acr(read_tree(os.path.join(GOLDEN, 'data', 'Albanian.tree.152tax.tre')), df.copy(),
                           prediction_method=[ml.MPPA, ml.MAP, ml.JOINT], model='F81', tau=tau, reoptimise=reoptimise)
        assert seen['tau_free'] == expect, (tau, reoptimise)
        assert seen['tau'] == [0 if tau is None else tau] * 3


def test_two_point_scheme_is_scipys_forward_difference():
    """
    batch.two_point_scheme: the points of scipy's 2-point gradient (absolute step 1e-8, mirrored at the bounds, relative
    fall-back where the step vanishes) and the steps to divide by -- (f(point_i) - f(x0)) / step_i must be
    approx_derivative's gradient bit for bit, or the batched optimiser would not walk scipy's iterates.
    """
    from scipy.optimize._numdiff import approx_derivative
    from pastml_amd import batch as B
    if B._adjust_scheme_to_bounds is None:
        pytest.skip('this SciPy build has no _adjust_scheme_to_bounds: the helper is called twice instead')
    rng = np.random.default_rng(1)
    for t in range(200):
        n = int(rng.integers(1, 14))
        lo = rng.uniform(-1, 0.5, n)
        up = lo + rng.uniform(1e-9, 2, n)
        x0 = lo + (up - lo) * rng.uniform(0, 1, n)
        if t % 4 == 0:
            x0[0] = up[0]            # the step is mirrored at an upper bound
        if t % 5 == 0:
            x0[-1] = lo[-1]
        if t % 7 == 0:               # 1e-8 is below the spacing of x0: scipy's relative fall-back
            x0, lo, up = x0 * 1e9, lo * 1e9 - 1, up * 1e9 + 1
        w = rng.normal(size=n)

        def f(x):
            return float(np.sin(x @ w) + np.sum(x ** 2))
        ref = approx_derivative(f, x0, method='2-point', abs_step=1e-8, f0=f(x0), bounds=(lo, up))
        points, steps = B.two_point_scheme(x0, lo, up)
        mine = (np.array([f(p) for p in points]) - f(x0)) / steps
        assert np.array_equal(np.atleast_1d(ref), mine), t


def test_kernel_points_decodes_a_batch_like_one_vector_at_a_time():
    """F81Model.kernel_points (all optimiser vectors of a batch decoded array-wise) == set_params_from_optimised +
    kernel_spec + rate_params per vector, bit for bit, and leaves the model at the last vector."""
    from pastml_amd.models import Model, ModelWithFrequencies
    from pastml_amd.models._closed_form import F81Model, JCModel
    fs = ForestStats([read_tree(os.path.join(GOLDEN, 'data', 'Albanian.tree.152tax.tre'))])
    rng = np.random.default_rng(0)
    for k in (2, 3, 5, 12, 36, 67):
        for optimise_tau in (False, True):
            for fixed in (False, True):
                states = ['s%02d' % i for i in range(k)]
                freqs = np.random.default_rng(k).dirichlet(np.ones(k))
                a, b = (F81Model(states=states, forest_stats=fs, optimise_tau=optimise_tau, frequencies=freqs.copy())
                        for _ in range(2))
                if fixed:
                    a.fix_extra_params()
                    b.fix_extra_params()
                bounds = a.get_bounds()
                lo, up = bounds[:, 0], bounds[:, 1]
                vectors = [lo + (up - lo) * rng.uniform(0, 1, len(lo)) ** 3 for _ in range(k + 2)]
                if optimise_tau:
                    vectors[1][1] = 0.0
                pa, pb = a.kernel_points(vectors), ModelWithFrequencies.kernel_points(b, vectors)
                for (sa, ra), (sb, rb) in zip(pa, pb):
                    assert np.array_equal(sa['pi'], sb['pi']) and ra == rb
                assert np.array_equal(a.frequencies, b.frequencies) and (a.sf, a.tau, a._tau_factor) == (b.sf, b.tau, b._tau_factor)
    jc = JCModel(states=['a', 'b', 'c'], forest_stats=fs)
    assert [p[1][0] for p in jc.kernel_points([np.array([v]) for v in (1., 2., 3.)])] == [1., 2., 3.] and jc.sf == 3.


def test_batched_diagonalisation_gives_the_single_calls_bits():
    """
    CustomRatesModel.kernel_points: the distinct frequency vectors of a batch of optimiser points are diagonalised by one
    stacked numpy.linalg.eig / inv call.  Every (d, A, A^-1) must be the array the reference's route gives for that vector --
    one get_diagonalisation per assignment of the frequencies (generator.py:16-30, CustomRatesModel.py:62-68) -- bit for bit:
    the optimiser's iterates depend on them.  With frequency smoothing a point's frequencies depend on the previous point's
    (models/__init__.py:331-335), which the deferred decoding must keep.
    """
    from pastml_amd.models import ModelWithFrequencies
    from pastml_amd.models._eigen import CustomRatesModel, get_diagonalisation, get_diagonalisation_batch
    fs = ForestStats([read_tree(os.path.join(GOLDEN, 'data', 'Albanian.tree.152tax.tre'))])
    rng = np.random.default_rng(1)
    for k in (2, 5, 20, 36):
        rates = np.triu(rng.uniform(0.1, 3, size=(k, k)), 1)
        rates = rates + rates.T
        freqs = rng.dirichlet(np.ones(k) * 2, size=7)
        d, a, ai = get_diagonalisation_batch(freqs, rates)
        for i in range(len(freqs)):
            d1, a1, ai1 = get_diagonalisation(freqs[i], rates)
            assert np.array_equal(d[i], d1) and np.array_equal(a[i], a1) and np.array_equal(ai[i], ai1)
        states = np.array(['s%02d' % i for i in range(k)])
        for smoothing in (False, True):
            m1, m2 = (CustomRatesModel(states=states, forest_stats=fs, rate_matrix=rates, frequencies=freqs[0].copy(),
                                       frequency_smoothing=smoothing) for _ in range(2))
            for fixed in (True, False):   # (the first stage of the search moves the scaling factor only)
                if fixed:
                    m1.fix_extra_params()
                    m2.fix_extra_params()
                else:
                    m1.unfix_extra_params()
                    m2.unfix_extra_params()
                bounds = m1.get_bounds()
                lo, up = bounds[:, 0], bounds[:, 1]
                x0 = m1.get_optimised_parameters()
                vectors = [x0.copy()] + [np.clip(x0 + 1e-8 * (np.arange(len(x0)) == i), lo, up) for i in range(len(x0))] + [x0.copy()]
                pa, pb = m1.kernel_points(vectors), ModelWithFrequencies.kernel_points(m2, vectors)
                assert len(pa) == len(pb) == len(vectors)
                for (sa, ra), (sb, rb) in zip(pa, pb):
                    assert ra == rb
                    for key in ('pi', 'd', 'A', 'Ainv'):
                        assert np.array_equal(sa[key], sb[key]), (k, smoothing, fixed, key)
                for attr in ('D_DIAGONAL', 'A', 'A_INV', 'frequencies'):
                    assert np.array_equal(getattr(m1, attr), getattr(m2, attr))
                # ... and a plain assignment afterwards still re-diagonalises
                f2 = rng.dirichlet(np.ones(k))
                if not fixed:
                    m1.unfix_extra_params()
                    m1._optimise_frequencies, keep = True, m1._optimise_frequencies
                    m1.frequencies = f2
                    m1._optimise_frequencies = keep
                    assert np.array_equal(m1.D_DIAGONAL, get_diagonalisation(f2, rates)[0])
                    m1.frequencies = m2.frequencies


def test_visible_devices(monkeypatch):
    """Which GPUs one process uses: its own as a rank of a multi-process launch, an explicit list, else all visible."""
    from pastml_amd import batch as B, hip
    for var in ('PASTML_AMD_DEVICES', 'LOCAL_RANK', 'PASTML_HIP_DEVICE', 'WORLD_SIZE'):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(hip, 'device_count', lambda: 4)
    assert B.visible_devices() == [0, 1, 2, 3]
    assert B.visible_devices(2) == [2]
    monkeypatch.setenv('PASTML_AMD_DEVICES', '1,3,3')
    assert B.visible_devices() == [1, 3, 3]
    monkeypatch.delenv('PASTML_AMD_DEVICES')
    monkeypatch.setenv('LOCAL_RANK', '2')
    assert B.visible_devices() == [2]
    monkeypatch.delenv('LOCAL_RANK')
    monkeypatch.setattr(hip, 'device_count', lambda: 0)
    assert B.visible_devices() == [0]


def test_bench_byte_model_knows_the_two_level_schedule():
    """bench.schedule_bytes restates, from the tree, which nodes the library's level schedule runs as two-level units
    (pml_tree_upload) and what the sweeps then move; the GPU side of this is test_gpu_api's profile-slot check."""
    import bench
    from pastml_amd import synthetic
    flat = synthetic.balanced_forest(12)
    # level launches (stored nodes x columns beyond the subtree blocks' reach): every node of height 3 is a two-level node
    sb = bench.schedule_bytes(flat, 64, 128)
    assert sb['n_two_level'] == flat.n_tips // 8
    plain = bench.schedule_bytes(flat, 64, 3)          # subtree blocks: no two-level units
    assert plain['n_two_level'] == 0 and plain['bottom_up_two_level'] == 0
    vec = 8 * 64
    n2 = sb['n_two_level']
    # bottom-up, per two-level node.  Plain: its two children as units (descriptor 32, vector + pi.v + exponent written
    # 528, two cherries gathered 64 and their pi.v + exponent written 32, four tips gathered 96 = 752 each) and its own unit
    # (32 + 528 + two children gathered 64 + their vectors read 1024 = 1648).  Two-level: 920.
    assert plain['bottom_up'] - sb['bottom_up'] == n2 * (2 * 752 + 1648 - 920)
    # top-down.  Plain: its own unit (descriptor 32, own row 528, two children: gather 32 + row written 528 each, their
    # vectors read 1024 = 2704) and its children's units (32 + 528 + two cherries (32 + 528) + four tips (24 + 528) = 3888
    # each).  Two-level: descriptor and own row once (560), per child 32 + 48 + 96 + 7 rows of 528.
    assert plain['top_down'] - sb['top_down'] == n2 * (2704 + 2 * 3888 - 560 - 2 * (176 + 7 * 528))
    assert vec == 512
    assert bench.schedule_bytes(flat, 16, 128)['n_two_level'] == 0      # units of fewer than 8 lanes (k <= 16)
    assert bench.schedule_bytes(flat, 20, 128)['n_two_level'] == flat.n_tips // 8
    # a caterpillar has no such nodes
    root = TreeNode(name='r', dist=0.0)
    cur = root
    for d in range(3000):
        cur.add_child(name='t{}'.format(d), dist=0.1)
        cur = cur.add_child(name='i{}'.format(d), dist=0.1)
    cur.add_child(name='ta', dist=0.1)
    cur.add_child(name='tb', dist=0.1)
    assert bench.schedule_bytes(FlatForest.from_trees([root]), 64, 128)['n_two_level'] == 0


def test_library_fd_points_are_numpys():
    """
    pml_host_f81_fd_points (the points of one forward-difference gradient, decoded, in one call into the library) ==
    two_point_scheme + F81Model.kernel_points bit for bit -- frequencies (numpy's pairwise summation, k beyond 128 included),
    scaling and smoothing factors, tau factors, steps -- and the model is left where kernel_points leaves it; a step that
    would leave the bounds is handed back to the numpy path.
    """
    from pastml_amd.batch import two_point_scheme
    from pastml_amd.models._closed_form import F81Model, JCModel
    fs = ForestStats([read_tree(os.path.join(GOLDEN, 'data', 'Albanian.tree.152tax.tre'))])
    rng = np.random.default_rng(3)
    checked = 0
    for k in (2, 3, 5, 9, 12, 30, 67, 130, 200):
        for optimise_tau in (False, True):
            for fixed in (False, True):
                states = ['s%03d' % i for i in range(k)]
                freqs = np.random.default_rng(k).dirichlet(np.ones(k))
                a, b = (F81Model(states=states, forest_stats=fs, optimise_tau=optimise_tau, frequencies=freqs.copy())
                        for _ in range(2))
                if fixed:
                    a.fix_extra_params()
                    b.fix_extra_params()
                bounds = a.get_bounds()
                lo, up = np.ascontiguousarray(bounds[:, 0]), np.ascontiguousarray(bounds[:, 1])
                for trial in range(4):
                    x = lo + (up - lo) * rng.uniform(0, 1, len(lo)) ** 3
                    if optimise_tau and trial == 1:
                        x[1] = 0.0
                    step = 1e-8 if trial < 3 else 1e-6   # (scipy's step; the polish run's)
                    made = a.fd_block(x, lo, up, step)
                    assert made is not None
                    block, steps = made
                    points, steps_np = two_point_scheme(x, lo, up, step)
                    ref = b.kernel_points(np.vstack((x[None, :], points)))
                    assert np.array_equal(steps, steps_np)
                    assert len(block) == len(ref) == len(x) + 1
                    for name in ('pi', 'sf', 'tau', 'tf'):
                        assert np.array_equal(getattr(block, name), getattr(ref, name)), (k, optimise_tau, fixed, name)
                    assert np.array_equal(a.frequencies, b.frequencies)
                    assert (a.sf, a.tau, a._tau_factor) == (b.sf, b.tau, b._tau_factor)
                    checked += 1
                # at the upper bound the step must be mirrored: scipy's helper, not the library
                x = up.copy()
                assert a.fd_block(x, lo, up) is None
    assert checked == 9 * 2 * 2 * 4
    jc = JCModel(states=['a', 'b', 'c'], forest_stats=fs)
    b0 = jc.get_bounds()
    block, steps = jc.fd_block(np.array([1.5]), np.ascontiguousarray(b0[:, 0]), np.ascontiguousarray(b0[:, 1]))
    assert len(block) == 2 and np.array_equal(block.pi, np.full((2, 3), 1 / 3)) and block.sf[0] == 1.5


def test_acr_names_a_character_with_too_many_states():
    """Boundary difference to the reference (no bound on k there): said up front, by name, before any device work -- 512 states
    for the F81 family, 256 for the models with a transition matrix per branch."""
    import pandas as pd
    from pastml_amd import acr as acr_module, hip
    from pastml_amd.tree import read_tree as read_newick
    assert acr_module.MAX_STATES == hip.MAX_STATES == 512 and acr_module.MAX_STATES_MATRIX == hip.MAX_STATES_MATRIX == 256
    n = 600
    tree = read_newick('(' + ','.join('t{}:1'.format(i) for i in range(n)) + ');')
    df = pd.DataFrame({'wide': ['s{:03d}'.format(i) for i in range(n)], 'ok': ['a', 'b'] * (n // 2)},
                      index=['t{}'.format(i) for i in range(n)])
    with pytest.raises(ValueError, match='wide has 600 states.*at most 512'):
        acr_module.acr(tree, df, prediction_method='MPPA', model='F81')
    df300 = pd.DataFrame({'wide': ['s{:03d}'.format(i % 300) for i in range(n)]}, index=['t{}'.format(i) for i in range(n)])
    with pytest.raises(ValueError, match='wide has 300 states.*at most 256.*CUSTOM_RATES'):
        acr_module.acr(tree, df300, prediction_method='MPPA', model='CUSTOM_RATES', column2rates={'wide': os.devnull})


def test_memory_plan_counts_the_transition_matrices_of_wide_eigen_models():
    """run_tasks cuts a group of characters to what the device holds (batch._column_bytes): eigen models beyond 64 states keep
    P(t) of every branch in HBM -- for the joint sweep of a character's own column (65 - 128 states: the optimiser's sum sweeps
    are fused), for every column beyond 128 states."""
    from pastml_amd import batch, hip
    from pastml_amd.tree import FlatForest
    flat = FlatForest.balanced(8)
    for k in (20, 64):
        assert batch._column_bytes(flat, k, [k + 1], kind=hip.KIND_EIGEN) == batch._column_bytes(flat, k, [k + 1])
    base = batch._column_bytes(flat, 100, [101])
    assert batch._column_bytes(flat, 100, [101], kind=hip.KIND_EIGEN) == base + flat.n_nodes * 8 * 100 * 100
    base = batch._column_bytes(flat, 200, [201])
    assert batch._column_bytes(flat, 200, [201], kind=hip.KIND_EIGEN) == base + flat.n_nodes * 8 * 200 * 200 * 202
    assert batch._column_bytes(flat, 300, [2], kind=hip.KIND_F81) == batch._column_bytes(flat, 300, [2])


def test_value2list_broadcasts_like_the_reference():
    from pastml_amd import value2list
    given = ['a', 'b']
    assert value2list(4, given, 'D') == ['a', 'b', 'D', 'D'] and given == ['a', 'b']   # (the caller's list is left alone)
    assert value2list(3, None, 'D') == ['D'] * 3 and value2list(3, 'x', 'D') == ['x'] * 3
    assert value2list(3, ['x'], 'D') == ['x'] * 3 and value2list(2, [], 'D') == ['D'] * 2
    assert value2list(2, ['a', 'b', 'c'], 'D') == ['a', 'b', 'c']
