"""
GPU tests of the drop-in boundary: ``pastml_amd.acr.acr()`` / ``ml.ml_acr()`` and the stand-alone functions, written
after the reference's own unit tests (tests/ACRParameterOptimisationMPPA*Test.py, ACRState*Test.py, HKYF81Test.py,
CUSTOM_RATESTest.py, PijTest.py) with the reference's pinned numbers and the golden outputs of the real reference.
"""
import os
from collections import Counter

import numpy as np
import pandas as pd
import pytest

from conftest import load_golden, GOLDEN
from pastml_amd import get_personalized_feature_name, STATES
from pastml_amd.acr import acr
from pastml_amd.annotation import ForestStats
from pastml_amd.ml import LH, LH_SF, MPPA, MAP, JOINT, ML, LOG_LIKELIHOOD, RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR, \
    MARGINAL_PROBABILITIES, MODEL, PastMLLikelihoodError
from pastml_amd import ml
from pastml_amd import hip
from pastml_amd.models.CustomRatesModel import CustomRatesModel, CUSTOM_RATES
from pastml_amd.models.F81Model import F81Model, F81
from pastml_amd.models.HKYModel import HKYModel, HKY, KAPPA, HKY_STATES, A, C, G, T
from pastml_amd.models.JCModel import JCModel, JC
from pastml_amd.models.EFTModel import EFT
from pastml_amd.models.JTTModel import JTTModel, JTT, JTT_STATES, JTT_RATE_MATRIX
from pastml_amd.models.generator import save_matrix
from pastml_amd.tree import read_tree, FlatForest

pytestmark = pytest.mark.gpu

DATA = os.path.join(GOLDEN, 'data')
TREE_NWK = os.path.join(DATA, 'Albanian.tree.152tax.tre')
STATES_INPUT = os.path.join(DATA, 'data.txt')
feature = 'Country'


def albania_df():
    return pd.read_csv(STATES_INPUT, index_col=0, header=0)[[feature]]


_cache = {}


def albania_result(model, method=MPPA):
    key = (model, method)
    if key not in _cache:
        tree = read_tree(TREE_NWK)
        _cache[key] = (tree, acr(tree, albania_df(), prediction_method=method, model=model))
    return _cache[key]


# pinned by the reference's tests
PINNED = {F81: dict(lnl=-110.178, restricted=-111.662, sf=3.841,
                    root={'Africa': 0.952, 'Albania': 0.001, 'EastEurope': 0.011, 'Greece': 0.011, 'WestEurope': 0.025},
                    node_4={'Africa': 0.944, 'Albania': 0.000, 'EastEurope': 0.000, 'Greece': 0.001, 'WestEurope': 0.054},
                    freqs={'Africa': 0.082, 'Albania': 0.028, 'EastEurope': 0.081, 'Greece': 0.365, 'WestEurope': 0.444}),
          JC: dict(lnl=-121.873, restricted=-123.421, sf=4.951),
          EFT: dict(lnl=-123.173, restricted=-125.359, sf=5.38)}


@pytest.mark.parametrize('model', [F81, JC, EFT])
def test_acr_mppa_albania_pinned_values(model):
    tree, results = albania_result(model)
    res = results[0]
    pin = PINNED[model]
    assert abs(res[LOG_LIKELIHOOD] - pin['lnl']) < 5e-4
    assert abs(res[RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(MPPA)] - pin['restricted']) < 5e-4
    assert abs(res[MODEL].sf - pin['sf']) < 5e-3
    assert abs(res[MODEL].frequencies.sum() - 1) < 1e-9
    mps = res[MARGINAL_PROBABILITIES]
    if 'root' in pin:
        for loc, v in pin['root'].items():
            assert abs(mps.loc['ROOT', loc] - v) < 5e-4
        for loc, v in pin['node_4'].items():
            assert abs(mps.loc['node_4', loc] - v) < 5e-4
        for loc, v in pin['freqs'].items():
            assert abs(res[MODEL].frequencies[np.where(res[STATES] == loc)][0] - v) < 5e-4
    for loc in res[STATES]:
        assert abs(mps.loc['02ALAY1660', loc] - (1 if loc == 'Albania' else 0)) < 1e-12


@pytest.mark.parametrize('model', [F81, JC, EFT])
def test_acr_mppa_albania_matches_reference_run(model):
    """Against the golden end-to-end output of the real reference (same scipy), far tighter than the pinned digits."""
    tree, results = albania_result(model)
    res = results[0]
    z = load_golden('albania_' + model)
    assert res[MODEL].name == str(z['opt_model_name'])
    np.testing.assert_allclose(res[LOG_LIKELIHOOD], z['opt_loglik'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(res[MODEL].sf, z['opt_sf'], rtol=2e-4)
    np.testing.assert_allclose(res[MODEL].frequencies, z['opt_frequencies'], atol=2e-5)
    for m in (JOINT, MAP, MPPA):
        np.testing.assert_allclose(res[RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(m)],
                                   z['opt_loglik_restricted_' + m], rtol=0, atol=1e-4)
    flat = FlatForest.from_trees([tree])
    nodes = flat.nodes
    mps = res[MARGINAL_PROBABILITIES]
    assert list(mps.index) == [n.name for n in nodes]
    assert list(mps.columns) == list(z['opt_states'])
    np.testing.assert_allclose(mps.values, z['opt_posterior'], rtol=0, atol=2e-5)
    # selected states, joint states, scenario statistics
    s2i = {s: i for i, s in enumerate(res[STATES])}
    sel = np.zeros((len(nodes), len(s2i)), dtype=np.int8)
    for i, n in enumerate(nodes):
        for s in getattr(n, feature):
            sel[i, s2i[s]] = 1
    assert np.array_equal(sel, z['opt_selected_mppa'])
    assert np.array_equal([getattr(n, feature + '_JOINT_STATE') for n in nodes], z['opt_joint_state'])
    assert res['num_unresolved_nodes'] == int(z['opt_num_unresolved_nodes'])
    assert float(res['num_scenarios']) == float(z['opt_num_scenarios'])
    np.testing.assert_allclose(res['num_states_per_node_avg'], z['opt_num_states_per_node_avg'])
    assert abs(res['percentage_of_unresolved_nodes'] - 100 * int(z['opt_num_unresolved_nodes']) / 305) < 1e-12


def test_likelihood_same_for_all_nodes():
    """tests/ACRParameterOptimisationMPPAF81Test.py:108-122."""
    tree, results = albania_result(F81)
    lh_feature = get_personalized_feature_name(feature, LH)
    lh_sf_feature = get_personalized_feature_name(feature, LH_SF)
    for node in tree.traverse():
        if not node.is_root() and not (node.is_leaf() and node.dist == 0):
            node_loglh = np.log10(getattr(node, lh_feature).sum()) - getattr(node, lh_sf_feature)
            parent_loglh = np.log10(getattr(node.up, lh_feature).sum()) - getattr(node.up, lh_sf_feature)
            assert round(abs(node_loglh - parent_loglh), 2) == 0
    root_loglh = np.log10(getattr(tree, lh_feature).sum()) - getattr(tree, lh_sf_feature)
    assert abs(root_loglh - results[0][LOG_LIKELIHOOD] / np.log(10)) < 1e-9


def reroot_tree_randomly(rng):
    """tests/ACRParameterOptimisationMPPAF81Test.py:20-37: drop the old root, re-root on a random branch."""
    rerooted_tree = read_tree(TREE_NWK)
    candidates = [_ for _ in rerooted_tree.traverse() if not _.is_root() and not _.up.is_root() and _.dist]
    new_root = candidates[rng.integers(len(candidates))]
    old_root_child = rerooted_tree.children[0]
    old_root_child_dist = old_root_child.dist
    other_children = list(rerooted_tree.children[1:])
    old_root_child.up = None
    for child in other_children:
        child.up = None
        old_root_child.add_child(child, dist=old_root_child_dist + child.dist)
    old_root_child.set_outgroup(new_root)
    new_root = new_root.up
    for _ in new_root.traverse():
        if not _.name:
            _.name = 'unknown'
    return new_root


def test_rerooted_values_are_the_same():
    """
    tests/ACRParameterOptimisationMPPAF81Test.py:51-83: optimised likelihood, scaling factor, frequencies and marginal
    probabilities do not depend on where the root is (2 decimals, as in the reference) -- and, stricter, at fixed
    parameters the likelihood of a reversible model is the same for every root (Felsenstein's pulley principle).
    """
    tree, results = albania_result(F81)
    acr_result = results[0]
    rng = np.random.default_rng(20)
    for _ in range(5):
        rerooted_tree = reroot_tree_randomly(rng)
        rerooted_acr_result = acr(rerooted_tree, albania_df(), prediction_method=MPPA, model=F81)[0]
        for freq, refreq in zip(acr_result[MODEL].frequencies, rerooted_acr_result[MODEL].frequencies):
            assert round(abs(freq - refreq), 2) == 0
        assert round(abs(acr_result[LOG_LIKELIHOOD] - rerooted_acr_result[LOG_LIKELIHOOD]), 2) == 0
        assert round(abs(acr_result[MODEL].sf - rerooted_acr_result[MODEL].sf), 2) == 0
        mps = acr_result[MARGINAL_PROBABILITIES]
        remps = rerooted_acr_result[MARGINAL_PROBABILITIES]
        for node_name in ('node_4', '02ALAY1660'):
            for loc in acr_result[STATES]:
                assert round(abs(mps.loc[node_name, loc] - remps.loc[node_name, loc]), 2) == 0

        # fixed parameters: same tree length, so the same sf means the same process on the unrooted tree
        fixed = {feature: {'scaling_factor': acr_result[MODEL].sf,
                           **dict(zip(acr_result[STATES], acr_result[MODEL].frequencies))}}
        fixed_result = acr(reroot_tree_randomly(rng), albania_df(), prediction_method=MPPA, model=F81,
                           column2parameters=fixed)[0]
        assert abs(fixed_result[LOG_LIKELIHOOD] - acr_result[LOG_LIKELIHOOD]) < 1e-9 * abs(acr_result[LOG_LIKELIHOOD])
        remps = fixed_result[MARGINAL_PROBABILITIES]
        for node_name in ('node_4', '02ALAY1660', 'node_10'):
            np.testing.assert_allclose(remps.loc[node_name].values, mps.loc[node_name].values, rtol=1e-8, atol=1e-12)


def test_state_selection_albania_mppa_f81():
    """tests/ACRStateMPPAF81Test.py:40-104 (on the uncollapsed tree: named nodes only)."""
    tree, _ = albania_result(F81)
    by_name = {n.name: n for n in tree.traverse()}
    assert getattr(tree, feature) == {'Africa'}
    assert getattr(by_name['node_48'], feature) == {'Africa', 'Greece', 'WestEurope'}
    assert getattr(by_name['node_32'], feature) == {'WestEurope', 'Greece'}
    assert getattr(by_name['node_80'], feature) == {'Greece'}
    assert getattr(by_name['01ALAY1715'], feature) == {'Albania'}
    assert getattr(by_name['94SEAF9671'], feature) == {'WestEurope'}


@pytest.mark.parametrize('method', [JOINT, MAP])
def test_state_selection_other_methods(method):
    """JOINT / MAP runs end to end and agree with the golden joint states / MAP masks at the reference optimum."""
    tree, results = albania_result(F81, method)
    res = results[0]
    assert res['method'] == method and res['character'] == feature
    z = load_golden('albania_F81')
    np.testing.assert_allclose(res[LOG_LIKELIHOOD], z['opt_loglik'], atol=2e-6)
    nodes = FlatForest.from_trees([tree]).nodes
    s2i = {s: i for i, s in enumerate(res[STATES])}
    chosen = np.array([s2i[next(iter(getattr(n, feature)))] for n in nodes])
    assert all(len(getattr(n, feature)) == 1 for n in nodes)
    if method == JOINT:
        assert np.array_equal(chosen, z['fix_joint_state'])
        assert MARGINAL_PROBABILITIES not in res
    else:
        assert np.array_equal(chosen, z['fix_masks_map'].argmax(axis=1))
        assert RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(JOINT) not in res


def test_meta_method_ml_returns_three_results():
    tree = read_tree(TREE_NWK)
    results = acr(tree, albania_df(), prediction_method=ML, model=JC)
    assert [r['method'] for r in results] == [JOINT, MAP, MPPA]
    assert [r['character'] for r in results] == [feature + '_' + m for m in (JOINT, MAP, MPPA)]
    for m in (JOINT, MAP, MPPA):
        assert all(hasattr(n, feature + '_' + m) for n in tree.traverse())


def test_fixed_parameters_skip_the_optimiser():
    z = load_golden('albania_F81')
    params = {'scaling_factor': float(z['opt_sf'])}
    params.update({s: f for s, f in zip(z['opt_states'], z['opt_frequencies'])})
    tree = read_tree(TREE_NWK)
    res = acr(tree, albania_df(), prediction_method=MPPA, model=F81, column2parameters={feature: params})[0]
    assert res[MODEL].get_num_params() == 0
    np.testing.assert_allclose(res[LOG_LIKELIHOOD], z['opt_loglik'], rtol=1e-11)
    np.testing.assert_allclose(res[MARGINAL_PROBABILITIES].values, z['opt_posterior'], rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(res[RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(MPPA)],
                               z['opt_loglik_restricted_MPPA'], rtol=1e-11)


def test_tau_smoothing_run():
    """tau > 0: no state alteration, branch lengths smoothed (models/__init__.py:39-42)."""
    z = load_golden('albania_F81')
    params = {'scaling_factor': float(z['tau_sf']), 'smoothing_factor': float(z['tau_tau'])}
    params.update({s: f for s, f in zip(z['tau_states'], z['tau_frequencies'])})
    tree = read_tree(TREE_NWK)
    res = acr(tree, albania_df(), prediction_method=MPPA, model=F81, column2parameters={feature: params},
              force_joint=False)[0]
    np.testing.assert_allclose(res[MODEL]._tau_factor, z['tau_tau_factor'], rtol=1e-14)
    np.testing.assert_allclose(res[LOG_LIKELIHOOD], z['tau_loglik'], rtol=1e-11)
    np.testing.assert_allclose(res[MARGINAL_PROBABILITIES].values, z['tau_posterior'], rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(res[RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(MPPA)],
                               z['tau_loglik_restricted_MPPA'], rtol=1e-11)
    assert res['num_unresolved_nodes'] == int(z['tau_mppa_num_unresolved'])


# ---------------------------------------------------------------------------------------------------------------------
def test_pij_model_identities():
    """tests/PijTest.py: the eigen-decomposed CUSTOM_RATES P(t) equals the closed forms of F81 / JC / HKY."""
    rng = np.random.default_rng(17)
    states10 = np.array(list('ABCDEFGHIJ'))
    for _ in range(5):
        t = 10 * rng.random()
        freqs = rng.random(10)
        freqs /= freqs.sum()
        ones = np.ones((10, 10)) - np.eye(10)
        p_cr = CustomRatesModel(sf=1, states=states10, forest_stats=None, frequencies=freqs, rate_matrix=ones).get_Pij_t(t)
        p_f81 = F81Model(sf=1, forest_stats=None, frequencies=freqs, states=states10).get_Pij_t(t)
        assert p_cr.shape == (10, 10) and np.allclose(p_cr, p_f81)
        eq = np.ones(10) / 10
        p_cr = CustomRatesModel(sf=1, states=states10, forest_stats=None, frequencies=eq, rate_matrix=ones).get_Pij_t(t)
        assert np.allclose(p_cr, JCModel(sf=1, forest_stats=None, states=states10).get_Pij_t(t))
        kappa = 20 * rng.random()
        f4 = rng.random(4)
        f4 /= f4.sum()
        rm = np.ones((4, 4)) - np.eye(4)
        rm[A, G] = rm[G, A] = rm[C, T] = rm[T, C] = kappa
        p_cr = CustomRatesModel(sf=1, states=HKY_STATES, forest_stats=None, frequencies=f4, rate_matrix=rm).get_Pij_t(t)
        assert np.allclose(p_cr, HKYModel(sf=1, forest_stats=None, kappa=kappa, frequencies=f4).get_Pij_t(t))


def test_hky_vs_f81_on_nucleotide_tree():
    """tests/HKYF81Test.py: HKY with kappa fixed to 1 reproduces F81; free kappa fits better."""
    tab = 'tree.152taxa.sf_0.5.A_0.6.C_0.15.G_0.2.T_0.05'
    df = pd.read_csv(os.path.join(DATA, tab + '.pastml.tab'), index_col=0, header=0, sep='\t')[['ACR']]
    nwk = os.path.join(DATA, tab + '.nwk')
    r_f81 = acr(read_tree(nwk), df.copy(), prediction_method=MPPA, model=F81)[0]
    tree = read_tree(nwk)
    r_hky1 = acr(tree, df.copy(), prediction_method=MPPA, model=HKY, column2parameters={'ACR': {KAPPA: 1}})[0]
    r_hky = acr(read_tree(nwk), df.copy(), prediction_method=MPPA, model=HKY)[0]
    z = load_golden('nucleotide_hky_f81')
    for label, r in (('f81', r_f81), ('hky_k1', r_hky1), ('hky', r_hky)):
        np.testing.assert_allclose(r[LOG_LIKELIHOOD], z[label + '_loglik'], atol=5e-5)
        np.testing.assert_allclose(r[MODEL].sf, z[label + '_sf'], rtol=2e-3)
        np.testing.assert_allclose(r[MARGINAL_PROBABILITIES].values, z[label + '_posterior'], atol=2e-4)
    np.testing.assert_allclose(r_hky[MODEL].kappa, z['hky_kappa'], rtol=2e-3)
    for param in (LOG_LIKELIHOOD, RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(MPPA)):
        assert abs(r_hky1[param] - r_f81[param]) < 5e-4
    assert abs(r_hky1[MODEL].sf - r_f81[MODEL].sf) < 5e-3
    assert r_hky[LOG_LIKELIHOOD] > r_hky1[LOG_LIKELIHOOD]
    np.testing.assert_allclose(r_hky1[MODEL].frequencies, r_f81[MODEL].frequencies, atol=5e-4)
    for name in ('ROOT', 'node_4'):
        np.testing.assert_allclose(r_hky1[MARGINAL_PROBABILITIES].loc[name].values,
                                   r_f81[MARGINAL_PROBABILITIES].loc[name].values, atol=5e-4)


def test_jtt_equals_custom_rates_with_jtt_matrix(tmp_path):
    """
    tests/CUSTOM_RATESTest.py:40-93: JTT and CUSTOM_RATES fed with the JTT matrix and the JTT run's parameter file give
    *identical* log-likelihoods (assertEqual) and marginal probabilities (np.all(==)): determinism, bit for bit.
    """
    rng = np.random.default_rng(4)
    tree = read_tree(TREE_NWK)
    for tip in tree:
        s = {JTT_STATES[int(rng.integers(20))]}
        tip.add_feature('state1', s)
        tip.add_feature('state2', set(s))
    r_jtt = acr(tree, columns=['state1'], column2states={'state1': JTT_STATES}, prediction_method=MPPA, model=JTT)[0]
    params = str(tmp_path / 'params.tab')
    with open(params, 'w') as f:
        f.write('parameter\tvalue\n')
        r_jtt[MODEL].save_parameters(f)
    rm = str(tmp_path / 'rate_matrix.txt')
    save_matrix(JTT_STATES, JTT_RATE_MATRIX, rm)
    r_cr = acr(tree, columns=['state2'], prediction_method=MPPA, model=CUSTOM_RATES,
               column2parameters={'state2': params}, column2rates={'state2': rm},
               column2states={'state2': JTT_STATES})[0]
    assert r_cr[MODEL].name == CUSTOM_RATES and r_cr[MODEL].get_num_params() == 0
    assert r_jtt[LOG_LIKELIHOOD] == r_cr[LOG_LIKELIHOOD]
    assert np.all(r_jtt[MARGINAL_PROBABILITIES].values == r_cr[MARGINAL_PROBABILITIES].values)


def test_zero_likelihood_raises_pastml_error():
    z = load_golden('edge_zero_likelihood')
    t = read_tree('((a:0.1,b:0.2)i1:0,(c:0.1,d:0.3)i2:0.2)r:0;')
    states = z['states']
    by = {n.name: n for n in t.traverse()}
    by['a'].add_feature('ch', {states[0]})
    by['b'].add_feature('ch', {states[0]})
    by['c'].add_feature('ch', {states[1]})
    by['d'].add_feature('ch', {states[2]})
    model = F81Model(states=states, forest_stats=ForestStats([t]), frequencies=z['frequencies'])
    model.freeze()
    ml.initialize_allowed_states(t, 'ch', states)
    by['r'].add_feature('ch_ALLOWED_STATES', np.array([0, 1, 0]))
    by['i1'].add_feature('ch_ALLOWED_STATES', np.array([1, 0, 0]))
    with pytest.raises(PastMLLikelihoodError) as e:
        ml.get_bottom_up_loglikelihood(t, 'ch', model, is_marginal=True, alter=True)
    assert str(e.value) == str(z['error_message'])


def test_standalone_sweep_functions_on_tree_features():
    """The importable functions of ml.py used one by one, as utilities/transition_counter.py does."""
    z = load_golden('albania_F81')
    tree = read_tree(TREE_NWK)
    from pastml_amd.annotation import preannotate_forest
    preannotate_forest([tree], df=albania_df())
    model = F81Model(states=z['opt_states'], forest_stats=ForestStats([tree]), sf=float(z['opt_sf']),
                     frequencies=z['opt_frequencies'])
    model.freeze()
    ml.initialize_allowed_states(tree, feature, model.states)
    lnl = ml.get_bottom_up_loglikelihood(tree, feature, model, is_marginal=True, alter=True)
    np.testing.assert_allclose(lnl, z['fix_loglik'], rtol=1e-11)
    # explicit alteration + alter=False, as ml_acr / marginal_counts do (ml.py:700-706)
    flat = FlatForest.from_trees([tree])
    problem = ml._problem_of(tree, feature, model)
    ml._pull_masks_from_features(problem, tree, feature)
    altered = problem.alter_zero_node_allowed_states()
    ml._push_masks_to_features(problem, feature)
    ml.get_bottom_up_loglikelihood(tree, feature, model, is_marginal=True, alter=False)
    bu = np.array([getattr(n, feature + '_BOTTOM_UP_LIKELIHOOD') for n in flat.nodes])
    bu_sf = np.array([getattr(n, feature + '_BOTTOM_UP_LIKELIHOOD_SF') for n in flat.nodes])
    with np.errstate(divide='ignore'):
        ours, ref = np.log10(bu) - bu_sf[:, None], np.log10(z['fix_bu']) - z['fix_bu_sf'][:, None]
    fin = np.isfinite(ref)
    np.testing.assert_allclose(ours[fin], ref[fin], atol=2e-10)
    ml.calculate_top_down_likelihood(tree, feature, model)
    ml.calculate_marginal_likelihoods(tree, feature, model.frequencies)
    assert not hasattr(tree, feature + '_BOTTOM_UP_LIKELIHOOD')
    mp = ml.convert_likelihoods_to_probabilities(tree, feature, model.states)
    np.testing.assert_allclose(mp.values, z['fix_posterior'], rtol=1e-9, atol=1e-300)
    assert list(mp.index) == list(z['node_names'])
    assert sorted(altered.tolist()) == list(z['fix_altered_nodes'])


def test_forest_of_trees_and_threads():
    """Several trees and several characters in one acr() call (thread pool over characters, acr.py:226-231)."""
    z = load_golden('edge_forest')
    flat = FlatForest(z['parent'], z['n_children'], z['first_child'], z['dist'], np.arange(int(z['n_roots'])))
    roots = flat.to_tree_nodes(names=list(z['node_names']))
    states = z['states']
    ann = z['annotation']
    for i, n in enumerate(flat.nodes):
        if ann[i].any():
            n.add_feature('ch', set(states[ann[i].astype(bool)]))
            n.add_feature('ch2', set(states[ann[i].astype(bool)]))
    params = {'scaling_factor': float(z['sf'])}
    params.update({s: f for s, f in zip(states, z['frequencies'])})
    res = acr(roots, columns=['ch', 'ch2'], column2states={'ch': states, 'ch2': states}, prediction_method=MPPA,
              model=F81, column2parameters={'ch': params, 'ch2': params}, threads=3)
    assert [r['character'] for r in res] == ['ch', 'ch2']
    for r in res:
        np.testing.assert_allclose(r[LOG_LIKELIHOOD], z['loglik'], rtol=1e-11)
        np.testing.assert_allclose(r[RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(MPPA)], z['loglik_restricted_MPPA'],
                                   rtol=1e-11)
        mp = r[MARGINAL_PROBABILITIES]
        # rows tree by tree (pd.concat of the per-tree tables, ml.py:713)
        order = np.lexsort((np.arange(flat.n_nodes), flat.tree_id))
        assert list(mp.index) == [z['node_names'][i] for i in order]
        np.testing.assert_allclose(mp.values, z['posterior'][order], rtol=1e-9, atol=1e-300)
        assert r['num_unresolved_nodes'] == int(z['mppa_num_unresolved'])
    sel = np.array([[s in getattr(n, 'ch') for s in states] for n in flat.nodes], dtype=np.int8)
    assert np.array_equal(sel, z['masks_mppa'])


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config 5: examples/HIV1C (3 619 tips)
# ---------------------------------------------------------------------------------------------------------------------
HIV = os.path.join(DATA, 'hiv1c')


def hiv_inputs(column):
    tree = read_tree(os.path.join(HIV, 'pastml_phyml_tree.nwk'))
    df = pd.read_csv(os.path.join(HIV, 'metadata_subset.tab'), sep='\t', index_col=0, header=0)
    df.index = df.index.map(str)
    return tree, df[[column]].copy()


def test_hiv1c_loc_at_stored_parameters():
    """Loc (k=12) at the parameters of examples/HIV1C/data/pastml_params: pinned log-likelihood -3692.227."""
    z = load_golden('hiv1c')
    tree, df = hiv_inputs('Loc')
    params = {'scaling_factor': float(z['locfix_sf'])}
    params.update({s: f for s, f in zip(z['loc_states'], z['locfix_frequencies'])})
    res = acr(tree, df, prediction_method=MPPA, model=F81, column2parameters={'Loc': params})[0]
    fs = res[MODEL].forest_stats
    np.testing.assert_array_equal([fs.avg_nonzero_brlen, fs.num_nodes, fs.num_tips, fs.forest_length],
                                  z['forest_stats'])
    assert abs(res[LOG_LIKELIHOOD] - (-3692.227)) < 5e-4
    np.testing.assert_allclose(res[LOG_LIKELIHOOD], z['locfix_loglik'], rtol=1e-11)
    for m in (JOINT, MAP, MPPA):
        np.testing.assert_allclose(res[RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(m)],
                                   z['locfix_loglik_restricted_' + m], rtol=1e-11)
    flat = FlatForest.from_trees([tree])
    mp = res[MARGINAL_PROBABILITIES]
    np.testing.assert_allclose(mp.values[z['sample']], z['locfix_posterior_sample'], rtol=1e-8, atol=1e-300)
    assert np.array_equal([getattr(n, 'Loc_JOINT_STATE') for n in flat.nodes], z['locfix_joint_state'])
    s2i = {s: i for i, s in enumerate(res[STATES])}
    sel = np.zeros((flat.n_nodes, len(s2i)), dtype=np.int8)
    for i, n in enumerate(flat.nodes):
        for s in getattr(n, 'Loc'):
            sel[i, s2i[s]] = 1
    assert np.array_equal(sel, z['locfix_selected_mppa'])
    assert res['num_unresolved_nodes'] == int(z['locfix_num_unresolved_nodes'])
    np.testing.assert_allclose(res['num_states_per_node_avg'], z['locfix_num_states_per_node_avg'])


@pytest.mark.parametrize('column,label', [('RT:K103N', 'k103n'), ('PR:L90M', 'l90m'), ('Loc', 'locopt')])
def test_hiv1c_optimised_columns(column, label):
    """Full ml_acr with parameter optimisation (reference: 10 s per binary column, 245 s for Loc)."""
    z = load_golden('hiv1c')
    tree, df = hiv_inputs(column)
    res = acr(tree, df, prediction_method=MPPA, model=F81)[0]
    assert list(res[STATES]) == list(z[label + '_states'])
    np.testing.assert_allclose(res[LOG_LIKELIHOOD], z[label + '_loglik'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(res[MODEL].sf, z[label + '_sf'], rtol=1e-3)
    np.testing.assert_allclose(res[MODEL].frequencies, z[label + '_frequencies'], atol=1e-4)
    np.testing.assert_allclose(res[MARGINAL_PROBABILITIES].values[z['sample']], z[label + '_posterior_sample'],
                               atol=2e-4)
    np.testing.assert_allclose(res[RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(MPPA)],
                               z[label + '_loglik_restricted_MPPA'], rtol=0, atol=5e-3)


@pytest.mark.parametrize('model', [F81, JC])
def test_batched_optimiser_reproduces_sequential_iterates(model, monkeypatch):
    """
    The optimiser evaluates the n_params + 1 finite-difference points of every L-BFGS-B gradient in one batched sweep;
    since every column is computed independently, the optimum must be the one of the point-by-point run, bit for bit.
    """
    out = []
    for flag in ('1', '0'):
        monkeypatch.setenv('PASTML_AMD_BATCHED_OPTIMISER', flag)
        tree = read_tree(TREE_NWK)
        res = acr(tree, albania_df(), prediction_method=MPPA, model=model)[0]
        out.append(res)
    a, b = out
    assert a[LOG_LIKELIHOOD] == b[LOG_LIKELIHOOD]
    assert a[MODEL].sf == b[MODEL].sf
    assert np.array_equal(a[MODEL].frequencies, b[MODEL].frequencies)
    assert np.array_equal(a[MARGINAL_PROBABILITIES].values, b[MARGINAL_PROBABILITIES].values)
    assert a[RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(MPPA)] == b[RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(MPPA)]


@pytest.mark.parametrize('model', [F81, JC])
def test_fallback_optimiser_driver_finds_the_same_optima(model, monkeypatch):
    """
    The single-loop driver calls scipy's PRIVATE reverse-communication routine (``_lbfgsb.setulb``, probed by its doc string,
    pastml_amd/batch.py); a SciPy build without it falls back to scipy.optimize.minimize with one thread per character.
    Forced here: same optima -- the iterates are L-BFGS-B's in both drivers -- and the pinned Albania values.
    """
    from pastml_amd import batch
    tree, results = albania_result(model)
    assert batch.single_loop_optimiser_available()
    monkeypatch.setattr(batch, '_setulb', None)
    assert not batch.single_loop_optimiser_available()
    tree2 = read_tree(TREE_NWK)
    df = albania_df()
    df['copy'] = df[feature]   # (two characters: the fallback's thread-per-character rendezvous is exercised too)
    res = acr(tree2, df, prediction_method=MPPA, model=model)
    want = results[0]
    for got in res:
        assert got[LOG_LIKELIHOOD] == want[LOG_LIKELIHOOD]
        assert got[MODEL].sf == want[MODEL].sf
        assert np.array_equal(got[MODEL].frequencies, want[MODEL].frequencies)
        assert np.array_equal(got[MARGINAL_PROBABILITIES].values, want[MARGINAL_PROBABILITIES].values)
    assert abs(res[0][LOG_LIKELIHOOD] - PINNED[model]['lnl']) < 5e-4 and abs(res[0][MODEL].sf - PINNED[model]['sf']) < 5e-3


def test_fallback_driver_follows_the_main_drivers_rules_for_many_parameters(monkeypatch):
    """
    A search of 20 and more free parameters (F81, 24 states on the Albania tree: 24 parameters) takes the extra steps of
    pastml_amd/batch.py -- the polish run with the 1e-6 step, continued once when it stops on the relative-reduction test --
    in BOTH drivers, under the same switches (the scipy fallback re-ran every start at ftol / 10 and never
    polished).  Same procedure, same routine: the same optimum; and with the switches off both give the reference's plain
    procedure.
    """
    from pastml_amd import batch
    k = 24
    states = np.array(['s{:02d}'.format(i) for i in range(k)])
    draw = np.random.default_rng(4)
    weights = draw.dirichlet(np.ones(k))
    assignment = {}

    def run():
        tree = read_tree(TREE_NWK)
        for tip in tree:
            assignment.setdefault(tip.name, states[draw.choice(k, p=weights)])
            tip.add_feature('many', {assignment[tip.name]})
        np.random.seed(239)
        return acr(tree, columns=['many'], column2states={'many': states}, prediction_method=MPPA, model=F81)[0]

    for polish in (1e-6, 0.0):
        monkeypatch.setattr(batch, 'POLISH_STEP', polish)
        main = run()
        with monkeypatch.context() as m:
            m.setattr(batch, '_setulb', None)
            assert not batch.single_loop_optimiser_available()
            fallback = run()
        assert abs(main[LOG_LIKELIHOOD] - fallback[LOG_LIKELIHOOD]) <= 1e-9 * abs(main[LOG_LIKELIHOOD]), polish
        np.testing.assert_allclose(main[MODEL].frequencies, fallback[MODEL].frequencies, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(main[MODEL].sf, fallback[MODEL].sf, rtol=1e-6)
        if polish:
            polished = main[LOG_LIKELIHOOD]
        else:
            assert polished >= main[LOG_LIKELIHOOD] - 1e-9   # (the polish can only raise ln L)


def test_too_many_states_are_refused_by_name():
    """The one bound the reference does not have (Limits): the library answers PML_ERR_UNSUPPORTED -- beyond 512
    states for the F81 family, beyond 256 for an eigen model --, acr() says which character it is before doing any work."""
    flat = FlatForest.balanced(4)
    with pytest.raises(hip.HipError) as e:
        hip.Engine(flat, 1, hip.MAX_STATES + 1)
    assert e.value.status == hip.PML_ERR_UNSUPPORTED and 'at most 512' in str(e.value)
    with hip.Engine(flat, 1, hip.MAX_STATES) as eng:   # the bound itself works
        eng.set_models([(dict(kind=0, pi=np.ones(hip.MAX_STATES) / hip.MAX_STATES), (1.0, 0.0, 1.0))])
        eng.set_tip_states(np.arange(flat.n_tips, dtype=np.int32) * 37 % hip.MAX_STATES)
        assert np.isfinite(eng.bottom_up(True)[0])
    k = hip.MAX_STATES_MATRIX + 1
    with hip.Engine(flat, 1, k) as eng:
        with pytest.raises(hip.HipError) as e:
            eng.set_models([(dict(kind=2, pi=np.ones(k) / k, d=np.zeros(k), A=np.eye(k), Ainv=np.eye(k)), (1.0, 0.0, 1.0))])
        assert e.value.status == hip.PML_ERR_UNSUPPORTED and 'at most 256' in str(e.value)


@pytest.mark.parametrize('model', [JC, EFT])
def test_acr_with_300_states_matches_reference_run(model):
    """
    acr() end to end beyond 256 states (the F81 family: 64 lanes x 8 states, 16-bit arg-max tables) against the reference's
    own run (tests/golden/make_golden.py::case_f81_k300: balanced 1 024-tip tree, tip states simulated down the tree -- some 300
    of them at the tips --, a tenth of the tips unannotated, MPPA with the scaling factor optimised).
    """
    from pastml_amd import synthetic
    z = load_golden('synthetic_f81_k300_L10')
    flat = synthetic.balanced_forest(int(z['n_levels']))
    tree = flat.to_tree_nodes()[0]
    tips = [flat.nodes[t] for t in flat.tips]
    names = synthetic.state_names(int(z['acr_n_candidates']))
    df = pd.DataFrame({'c0': [names[z['acr_tip_states'][j]] if z['tip_observed'][j] else None for j in range(len(tips))]},
                      index=[t.name for t in tips])
    res = acr(tree, df, prediction_method=MPPA, model=model, threads=1)[0]
    pre = 'acr_{}_'.format(model)
    assert list(res[STATES]) == list(z[pre + 'states']) and len(res[STATES]) > 256
    np.testing.assert_allclose(res[LOG_LIKELIHOOD], z[pre + 'loglik'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(res[MODEL].sf, z[pre + 'sf'], rtol=2e-4)
    for m in (JOINT, MAP, MPPA):
        np.testing.assert_allclose(res[RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(m)], z[pre + 'loglik_restricted_' + m],
                                   rtol=0, atol=1e-4)
    nodes = FlatForest.from_trees([tree]).nodes
    s = z['sample']
    mps = res[MARGINAL_PROBABILITIES]
    assert list(mps.index) == [n.name for n in nodes]
    np.testing.assert_allclose(mps.values[s], z[pre + 'posterior'], rtol=0, atol=2e-5)
    s2i = {st: i for i, st in enumerate(res[STATES])}
    sel = np.zeros((len(nodes), len(s2i)), dtype=np.int8)
    for i, n in enumerate(nodes):
        for st in getattr(n, 'c0'):
            sel[i, s2i[st]] = 1
    assert np.array_equal(sel[s], z[pre + 'selected_mppa'])
    assert np.array_equal(sel.sum(axis=1), z[pre + 'n_selected'])
    assert np.array_equal([getattr(n, 'c0_JOINT_STATE') for n in nodes], z[pre + 'joint_state'])
    assert res['num_unresolved_nodes'] == int(z[pre + 'num_unresolved_nodes'])
    np.testing.assert_allclose(res['num_states_per_node_avg'], z[pre + 'num_states_per_node_avg'])


def test_serialised_tables_round_trip(tmp_path):
    """
    pastml/acr.py:45-73: parameter + marginal-probability tables in the reference's format; the parameter file fed back
    through column2parameters reproduces the run without optimisation (as tests/CUSTOM_RATESTest.py does), and the
    tables of the reference's stored Albania run (examples/Albania/data/pastml/MPPA/F81, v1.9.15) parse the same way.
    """
    from pastml_amd.acr import _serialize_acr
    from pastml_amd.file import get_pastml_parameter_file, get_pastml_marginal_prob_file
    tree, results = albania_result(F81)
    res = results[0]
    _serialize_acr((res, str(tmp_path)))
    pfile = tmp_path / get_pastml_parameter_file(MPPA, F81, feature)
    mfile = tmp_path / get_pastml_marginal_prob_file(MPPA, F81, feature)
    assert pfile.name == 'params.character_Country.method_MPPA.model_F81.tab'
    assert mfile.name == 'marginal_probabilities.character_Country.model_F81.tab'
    params = pd.read_csv(pfile, sep='\t', index_col=0, header=0)['value']
    assert float(params['log_likelihood']) == res[LOG_LIKELIHOOD]
    assert float(params['scaling_factor']) == res[MODEL].sf
    assert int(params['num_nodes']) == 305 and int(params['num_tips']) == 154
    for s, f in zip(res[STATES], res[MODEL].frequencies):
        assert float(params[s]) == f
    mp = pd.read_csv(mfile, sep='\t', index_col=0, header=0)
    assert mp.index.name == 'node' and list(mp.columns) == list(res[STATES])
    np.testing.assert_allclose(mp.values, res[MARGINAL_PROBABILITIES].values, rtol=1e-11)  # pandas' fast float parser
    again = acr(read_tree(TREE_NWK), albania_df(), prediction_method=MPPA, model=F81,
                column2parameters={feature: str(pfile)})[0]
    assert again[MODEL].get_num_params() == 0
    np.testing.assert_allclose(again[LOG_LIKELIHOOD], res[LOG_LIKELIHOOD], rtol=1e-12)
    np.testing.assert_allclose(again[MARGINAL_PROBABILITIES].values, res[MARGINAL_PROBABILITIES].values, rtol=1e-9,
                               atol=1e-300)


def test_marginal_counts_statistical_parity():
    """
    ml.marginal_counts against the reference's estimate (40 000 repetitions each): the two are independent Monte-Carlo
    estimates of the same expectation, so they must agree within sampling noise (tests/MRANDJCTest.py compares to 2
    decimals in the same spirit).
    """
    from pastml_amd import synthetic
    z = load_golden('marginal_counts')
    n_rep = int(z['n_repetitions'])
    np.random.seed(7)
    flat = synthetic.balanced_forest(6)
    roots = flat.to_tree_nodes()
    states = synthetic.state_names(4)
    for j, t in enumerate(flat.tips):
        flat.nodes[t].add_feature('c', {states[z['jc_tip_states'][j]]})
    model = JCModel(states=states, forest_stats=ForestStats(roots), sf=float(z['jc_sf']))
    model.freeze()
    ours = ml.marginal_counts(roots, 'c', model, n_repetitions=n_rep)
    ref = z['jc_counts']
    # every entry is a mean of n_rep scenario counts; a generous bound on its standard error
    tol = 6 * np.sqrt(np.maximum(ref, 0.05) / n_rep) * 3 + 0.02
    assert np.all(np.abs(ours - ref) < tol), np.abs(ours - ref).max()
    assert abs(ours.sum() - ref.sum()) < 0.3

    zz = load_golden('albania_F81')
    tree = read_tree(TREE_NWK)
    from pastml_amd.annotation import preannotate_forest
    preannotate_forest([tree], df=albania_df())
    model = F81Model(states=zz['opt_states'], forest_stats=ForestStats([tree]), sf=float(zz['opt_sf']),
                     frequencies=zz['opt_frequencies'])
    model.freeze()
    ours = ml.marginal_counts([tree], feature, model, n_repetitions=n_rep)
    ref = z['albania_counts']
    tol = 6 * np.sqrt(np.maximum(ref, 0.05) / n_rep) * 3 + 0.03
    assert np.all(np.abs(ours - ref) < tol), (np.abs(ours - ref).max(), ours.round(3), ref.round(3))


def test_eigen_models_optimised_match_reference():
    """
    acr() with JTT (sf free) and CUSTOM_RATES (sf + 4 frequency ratios free, the model is re-diagonalised at every
    optimiser point) against the reference's end-to-end run; random tip states drive sf to its upper bound, which
    exercises the bound handling of the batched finite differences.
    """
    z = load_golden('eigen_optimised')
    tree = read_tree(TREE_NWK)
    tips = {t.name: t for t in tree}
    for name, s in zip(z['jtt_tip_names'], z['jtt_tip_states']):
        tips[str(name)].add_feature('aa', {JTT_STATES[s]})
    res = acr(tree, columns=['aa'], column2states={'aa': JTT_STATES}, prediction_method=MPPA, model=JTT)[0]
    np.testing.assert_allclose(res[LOG_LIKELIHOOD], z['jtt_loglik'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(res[MODEL].sf, z['jtt_sf'], rtol=1e-6)
    np.testing.assert_allclose(res[MARGINAL_PROBABILITIES].values, z['jtt_posterior'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(res[RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(MPPA)],
                               z['jtt_loglik_restricted_MPPA'], rtol=0, atol=1e-5)
    assert res['num_unresolved_nodes'] == int(z['jtt_num_unresolved_nodes'])

    tree = read_tree(TREE_NWK)
    tips = {t.name: t for t in tree}
    states = z['cr_states']
    for name, s in zip(z['jtt_tip_names'], z['cr_tip_states']):
        tips[str(name)].add_feature('cr', {states[s]})
    res = acr(tree, columns=['cr'], column2states={'cr': states}, prediction_method=MPPA, model=CUSTOM_RATES,
              column2rates={'cr': os.path.join(DATA, 'custom_rates_k5.txt')})[0]
    np.testing.assert_array_equal(res[MODEL].rate_matrix, z['cr_rate_matrix'])
    np.testing.assert_allclose(res[LOG_LIKELIHOOD], z['cr_loglik'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(res[MODEL].frequencies, z['cr_frequencies'], atol=2e-4)
    np.testing.assert_allclose(res[MARGINAL_PROBABILITIES].values, z['cr_posterior'], rtol=0, atol=5e-4)


def test_acr_custom_rates_with_72_states_fused_and_materialised_agree(tmp_path, monkeypatch):
    """
    acr() end to end with a CUSTOM_RATES model of 72 states (beyond 64: the sum sweeps with one matrix in LDS, the joint sweep on
    P(t) built by the matrix-core batch): frequencies given, scaling factor optimised -- the same matrices evaluation after
    evaluation -- and then frequencies free as well (every point of every gradient its own eigendecomposition and its own
    re-orthonormalisation on the device).  Against the same call with the fused sweeps switched off (P(t) of every branch in
    HBM in every evaluation): the optimum, the posteriors, the selected states.
    """
    from pastml_amd import synthetic
    k, n_tips = 76, 300   # (72 of the 76 candidate states are observed at the tips)
    rng = np.random.default_rng(7070)
    flat = FlatForest.random(n_tips, seed=70, max_arity=3, lo=0.02, hi=0.4)
    tree = flat.to_tree_nodes()[0]
    names = synthetic.state_names(k)
    st = np.zeros(flat.n_nodes, dtype=np.int64)
    for n in range(flat.n_nodes):   # ids are in level order: parents first
        p = flat.parent[n]
        st[n] = rng.integers(k) if p < 0 or rng.random() > np.exp(-4.0 * flat.dist[n]) else st[p]
    tips = [flat.nodes[t] for t in flat.tips]
    df = pd.DataFrame({'c': [names[st[t]] for t in flat.tips]}, index=[t.name for t in tips])
    states = np.array(sorted(set(df['c'])))
    kk = len(states)
    assert 64 < kk <= 128
    rates = np.triu(rng.uniform(0.2, 2.0, size=(kk, kk)), 1)
    rates = rates + rates.T
    rate_file = str(tmp_path / 'rates.txt')
    save_matrix(states, rates, rate_file)
    freqs = rng.dirichlet(np.ones(kk) * 5)
    params = {s: f for s, f in zip(states, freqs)}

    def run(fused, fixed_frequencies):
        hip.drain_engine_pool()   # (the switches of a context are read when it is created)
        if fused:
            monkeypatch.delenv('PASTML_HIP_NO_EIGEN_GEMM', raising=False)
        else:
            monkeypatch.setenv('PASTML_HIP_NO_EIGEN_GEMM', '1')
        t = flat.to_tree_nodes()[0]
        return t, acr(t, df, prediction_method=MPPA, model=CUSTOM_RATES, column2rates={'c': rate_file},
                      column2parameters={'c': params} if fixed_frequencies else None)[0]

    try:
        for fixed in (True, False):
            (ta, a), (tb, b) = run(True, fixed), run(False, fixed)
            np.testing.assert_allclose(a[LOG_LIKELIHOOD], b[LOG_LIKELIHOOD], rtol=0, atol=1e-5 if fixed else 2e-3)
            np.testing.assert_allclose(a[MODEL].sf, b[MODEL].sf, rtol=1e-5 if fixed else 2e-2)
            np.testing.assert_allclose(a[MARGINAL_PROBABILITIES].values, b[MARGINAL_PROBABILITIES].values, rtol=0,
                                       atol=1e-6 if fixed else 2e-2)
            if fixed:
                np.testing.assert_array_equal(a[MODEL].frequencies, b[MODEL].frequencies)
                for na, nb in zip(FlatForest.from_trees([ta]).nodes, FlatForest.from_trees([tb]).nodes):
                    assert getattr(na, 'c') == getattr(nb, 'c') and getattr(na, 'c_JOINT_STATE') == getattr(nb, 'c_JOINT_STATE')
                assert a['num_unresolved_nodes'] == b['num_unresolved_nodes']
    finally:
        monkeypatch.delenv('PASTML_HIP_NO_EIGEN_GEMM', raising=False)
        hip.drain_engine_pool()


@pytest.mark.parametrize('model_name,k', [('F81', 5), ('HKY', 4), ('JTT', 20)])
def test_marginal_counts_device_sampler_agrees_with_host_sampler(model_name, k):
    """
    pml_marginal_counts (scenarios drawn on the device, Philox) against the host sampler that follows ml.py:753-862 line
    by line with numpy's generator: two independent Monte-Carlo estimates of the same k x k expectation.  F81 uses the
    closed-form P(t) inside the sampler, HKY the materialised one, JTT (k = 20: fused matrix-core sweeps that never
    build P) materialises it for the sampler.
    """
    from pastml_amd.models.HKYModel import HKYModel
    from pastml_amd.models.JTTModel import JTTModel, JTT_STATES
    from pastml_amd.tree import FlatForest
    rng = np.random.default_rng(5 + k)
    flat = FlatForest.random(70, seed=17 + k, max_arity=3)
    roots = flat.to_tree_nodes()
    if model_name == 'JTT':
        states = np.array(JTT_STATES)
    elif model_name == 'HKY':
        states = np.array(['A', 'C', 'G', 'T'])
    else:
        states = np.array(['s{}'.format(i) for i in range(k)])
    for t in flat.tips:
        if rng.random() < 0.85:
            flat.nodes[t].add_feature('c', {states[rng.integers(len(states))]})
    fs = ForestStats(roots)
    if model_name == 'F81':
        model = F81Model(states=states, forest_stats=fs, sf=1.2 / fs.avg_nonzero_brlen,
                         frequencies=rng.dirichlet(np.ones(k) * 3))
    elif model_name == 'HKY':
        model = HKYModel(states=states, forest_stats=fs, sf=1.0 / fs.avg_nonzero_brlen,
                         frequencies=rng.dirichlet(np.ones(4) * 3), kappa=3.0)
    else:
        model = JTTModel(states=states, forest_stats=fs, sf=0.8 / fs.avg_nonzero_brlen)
    model.freeze()
    n_rep = 20000
    np.random.seed(3)
    dev = ml.marginal_counts(roots, 'c', model, n_repetitions=n_rep)
    host = ml.marginal_counts(roots, 'c', model, n_repetitions=n_rep, device_sampling=False)
    assert dev.shape == host.shape == (len(states), len(states))
    # entries are means of n_rep scenario counts with variance of the order of their value
    tol = 8 * np.sqrt(np.maximum(host, 0.02) / n_rep) + 0.01
    assert np.all(np.abs(dev - host) < tol), (np.abs(dev - host).max(), (np.abs(dev - host) / tol).max())
    assert abs(dev.sum() - host.sum()) < 0.05 * max(1.0, host.sum())
    # same seed, same result: the generator is counter-based
    np.random.seed(3)
    assert np.array_equal(dev, ml.marginal_counts(roots, 'c', model, n_repetitions=n_rep))


@pytest.mark.parametrize('model_name,k', [('F81', 6), ('JTT', 20)])
def test_marginal_counts_with_altered_nodes_on_the_device(model_name, k):
    """
    Forests with zero-length branches: the zero-branch handling alters nodes (pastml/ml.py:352-387), and the reference gives the
    (parent, child) pairs with an altered end fractional counts (ml.py:806-812, 840-853).  Round 6: the scenarios are drawn on the
    device all the same (pml_marginal_counts_altered), the host adds those pairs from the nodes' state counts -- against the host
    sampler that follows the reference line by line (two Monte-Carlo estimates of one expectation), and reproducibly.
    """
    from pastml_amd.models.JTTModel import JTTModel, JTT_STATES
    from pastml_amd.tree import FlatForest
    rng = np.random.default_rng(50 + k)
    flat = FlatForest.random(90, seed=23 + k, max_arity=3, zero_frac=0.3)
    roots = flat.to_tree_nodes()
    states = np.array(JTT_STATES) if model_name == 'JTT' else np.array(['s{}'.format(i) for i in range(k)])
    # (tips of one zero-length cluster with different states: what the alteration is there for)
    for t in flat.tips:
        if rng.random() < 0.9:
            flat.nodes[t].add_feature('c', {states[rng.integers(len(states))]})
    fs = ForestStats(roots)
    if model_name == 'F81':
        model = F81Model(states=states, forest_stats=fs, sf=1.2 / fs.avg_nonzero_brlen, frequencies=rng.dirichlet(np.ones(k) * 3))
    else:
        model = JTTModel(states=states, forest_stats=fs, sf=0.8 / fs.avg_nonzero_brlen)
    model.freeze()
    n_rep = 20000
    calls = []
    real = hip.Engine.marginal_counts_altered

    def spy(self, *a, **kw):
        calls.append(1)
        return real(self, *a, **kw)
    hip.Engine.marginal_counts_altered = spy
    try:
        np.random.seed(3)
        dev = ml.marginal_counts(roots, 'c', model, n_repetitions=n_rep)
        assert calls, 'no node was altered: the test does not test what it says'
        host = ml.marginal_counts(roots, 'c', model, n_repetitions=n_rep, device_sampling=False)
        assert dev.shape == host.shape == (len(states), len(states))
        tol = 8 * np.sqrt(np.maximum(host, 0.02) / n_rep) + 0.01
        assert np.all(np.abs(dev - host) < tol), (np.abs(dev - host).max(), (np.abs(dev - host) / tol).max())
        assert abs(dev.sum() - host.sum()) < 0.05 * max(1.0, host.sum())
        np.random.seed(3)
        np.testing.assert_allclose(dev, ml.marginal_counts(roots, 'c', model, n_repetitions=n_rep), rtol=0, atol=1e-12)
    finally:
        hip.Engine.marginal_counts_altered = real


def test_marginal_counts_beyond_256_states_take_the_host_sampler():
    """ml.marginal_counts with 300 states (the device sampler's tables hold 256): the host sampler on the device's bottom-up and
    top-down vectors, as for forests with altered nodes -- every scenario has one state change per differing parent / child pair,
    so the counts of an all-observed cherry-free star of tips are known; here: shape, non-negativity, total within the number of
    branches, and the same matrix from the same numpy seed."""
    from pastml_amd.tree import FlatForest
    k = 300
    rng = np.random.default_rng(300)
    flat = FlatForest.random(60, seed=300, max_arity=3)
    roots = flat.to_tree_nodes()
    states = np.array(['s{:03d}'.format(i) for i in range(k)])
    for t in flat.tips:
        flat.nodes[t].add_feature('c', {states[rng.integers(k)]})
    fs = ForestStats(roots)
    model = F81Model(states=states, forest_stats=fs, sf=1.0 / fs.avg_nonzero_brlen, frequencies=rng.dirichlet(np.ones(k) * 3))
    model.freeze()
    np.random.seed(4)
    a = ml.marginal_counts(roots, 'c', model, n_repetitions=200)
    assert a.shape == (k, k) and np.all(a >= -1e-12) and np.isfinite(a).all()
    assert 0 < a.sum() <= flat.n_nodes
    np.random.seed(4)
    assert np.array_equal(a, ml.marginal_counts(roots, 'c', model, n_repetitions=200))


def test_engine_pool_reuse_and_model_kind_switch():
    """
    A released engine is handed out again for the same forest / width / k, with no state of the previous analysis
    leaking into the next one -- including a change of the model kind (F81 -> HKY on four states), which a ctx accepts
    when all of its columns are set at once.
    """
    from pastml_amd import hip
    from test_gpu_parity import random_masks, random_spec
    rng = np.random.default_rng(21)
    flat = FlatForest.random(150, seed=9, max_arity=3)
    k = 4
    masks_a, masks_b = random_masks(flat, k, rng)[None], random_masks(flat, k, rng)[None]
    spec_f81, spec_hky = random_spec('F81', k, rng), random_spec('HKY', k, rng)

    def run(eng, spec, masks):
        eng.set_models([(spec, (1.3, 0.0, 1.0))])
        eng.set_masks(masks)
        lnl = eng.bottom_up(True)
        post, _, _ = eng.top_down_marginals()
        lnl_j = eng.bottom_up(False)
        return lnl.copy(), post.copy(), lnl_j.copy(), eng.joint_backtrace().copy()

    hip.drain_engine_pool()
    first = hip.acquire_engine(flat, 1, k)
    a = run(first, spec_f81, masks_a)
    hip.release_engine(first)
    again = hip.acquire_engine(flat, 1, k)
    assert again is first
    b = run(again, spec_hky, masks_b)      # other kind, other masks, on the pooled ctx
    c = run(again, spec_f81, masks_a)      # and back
    hip.release_engine(again)
    with hip.Engine(flat, 1, k) as fresh:
        ref_b = run(fresh, spec_hky, masks_b)
    for x, y in zip(b, ref_b):
        assert np.array_equal(x, y)
    for x, y in zip(a, c):
        assert np.array_equal(x, y)
    # another forest never gets this engine
    other = hip.acquire_engine(FlatForest.random(150, seed=10, max_arity=3), 1, k)
    assert other is not first
    hip.release_engine(other)
    hip.drain_engine_pool()


# ---------------------------------------------------------------------------------------------------------------------
# characters as columns (pastml_amd.batch): one acr() call batches all characters of a group on the device
# ---------------------------------------------------------------------------------------------------------------------
def _multi_character_table(tree, seed=5):
    """Six characters on the Albanian tree: Country (k=5) + five synthetic ones with 2, 2, 3, 5 and 5 states, some tips
    unannotated, so that one call holds three groups (k = 2, 3, 5) of different sizes."""
    rng = np.random.default_rng(seed)
    df = albania_df()
    tips = [t.name for t in tree]
    for name, k, missing in (('bin_a', 2, 0.0), ('bin_b', 2, 0.2), ('tri', 3, 0.1), ('five_a', 5, 0.0), ('five_b', 5, 0.3)):
        values = pd.Series(['v{}'.format(i) for i in rng.integers(0, k, size=len(tips))], index=tips)
        values[rng.random(len(tips)) < missing] = np.nan
        df[name] = values.reindex(df.index)
    return df


def test_batched_characters_equal_one_by_one():
    """
    Every step of ml_acr runs once for all characters of a group (lock-step L-BFGS-B instances served by one sweep of
    sum(n_params + 1) columns, one joint sweep, one marginal pass, one selection): each character must get exactly the
    numbers of a run on its own -- optimum, posteriors, selected states, restricted likelihoods -- bit for bit.
    """
    from pastml_amd.batch import run_tasks
    tree = read_tree(TREE_NWK)
    df = _multi_character_table(tree)
    models = [F81, JC, F81, EFT, F81, JC]
    together = acr(tree, df.copy(), prediction_method=MPPA, model=models)
    stats = dict(run_tasks.last_stats)
    assert stats['groups'] == 3 and [r['character'] for r in together] == list(df.columns)
    flat = FlatForest.from_trees([tree])
    selected = {r['character']: [getattr(n, r['character']) for n in flat.nodes] for r in together}
    rounds_alone = 0
    for column, model, batched in zip(df.columns, models, together):
        tree1 = read_tree(TREE_NWK)
        alone = acr(tree1, df[[column]].copy(), prediction_method=MPPA, model=model)[0]
        rounds_alone += run_tasks.last_stats['rounds']
        assert alone[LOG_LIKELIHOOD] == batched[LOG_LIKELIHOOD], column
        assert alone[MODEL].sf == batched[MODEL].sf
        assert np.array_equal(alone[MODEL].frequencies, batched[MODEL].frequencies)
        assert np.array_equal(alone[MARGINAL_PROBABILITIES].values, batched[MARGINAL_PROBABILITIES].values)
        for m in (JOINT, MAP, MPPA):
            key = RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(m)
            assert alone[key] == batched[key], (column, m)
        for key in ('num_scenarios', 'num_unresolved_nodes', 'num_states_per_node_avg'):
            assert alone[key] == batched[key]
        flat1 = FlatForest.from_trees([tree1])
        assert [getattr(n, column) for n in flat1.nodes] == selected[column]
        assert np.array_equal([getattr(n, column + '_JOINT_STATE') for n in flat1.nodes],
                              [getattr(n, column + '_JOINT_STATE') for n in flat.nodes])
    # the point of the exercise: the batch needs as many sweep rounds as its slowest character, not the sum
    assert stats['rounds'] < rounds_alone


def test_meta_method_ml_and_mixed_methods_in_one_call():
    """Characters with different prediction methods go to different groups; ML reports JOINT, MAP and MPPA."""
    tree = read_tree(TREE_NWK)
    df = _multi_character_table(tree)[['Country', 'bin_a', 'tri']]
    res = acr(tree, df.copy(), prediction_method=[ML, MAP, JOINT], model=F81)
    assert [(r['character'], r['method']) for r in res] == [('Country_JOINT', JOINT), ('Country_MAP', MAP),
                                                            ('Country_MPPA', MPPA), ('bin_a', MAP), ('tri', JOINT)]
    ref = albania_result(F81)[1][0]
    assert res[2][LOG_LIKELIHOOD] == ref[LOG_LIKELIHOOD]
    assert res[2][RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(MPPA)] == ref[RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(MPPA)]
    assert MARGINAL_PROBABILITIES in res[3] and MARGINAL_PROBABILITIES not in res[4]
    assert hasattr(tree, 'Country_MPPA') and hasattr(tree, 'Country_JOINT') and not hasattr(tree, 'bin_a_JOINT_STATE')


def test_columnar_node_features_behave_like_attributes():
    """Results live in columns of the flat forest; per node they read, shadow and delete like ete3 features."""
    tree, results = albania_result(F81)
    node = tree.children[0]
    assert isinstance(getattr(node, feature), set) and feature in node.features
    allowed = getattr(node, feature + '_ALLOWED_STATES')
    assert allowed.dtype == int and set(results[0][STATES][allowed.astype(bool)]) == getattr(node, feature)
    lh = getattr(node, get_personalized_feature_name(feature, LH))
    assert lh.shape == (5,) and isinstance(getattr(node, get_personalized_feature_name(feature, LH_SF)), float)
    node.add_feature(feature, {'Mars'})                 # a value set on the node hides the column's row ...
    assert getattr(node, feature) == {'Mars'} and getattr(tree, feature) != {'Mars'}
    node.del_feature(feature)                           # ... and deleting removes the feature of that node only
    assert not hasattr(node, feature) and hasattr(tree, feature)
    copy = tree.copy()                                  # a copy materialises what its nodes see
    assert getattr(copy, feature) == getattr(tree, feature)
    _cache.clear()


def test_groups_are_cut_to_fit_the_device(monkeypatch):
    """A group whose columns would not fit the free device memory is reconstructed in chunks (characters
    were one unbounded engine each): same results, more batches."""
    from pastml_amd.batch import run_tasks
    tree = read_tree(TREE_NWK)
    df = _multi_character_table(tree)[['bin_a', 'bin_b', 'five_a', 'five_b', 'Country']]
    whole = acr(tree, df.copy(), prediction_method=MPPA, model=F81)
    assert run_tasks.last_stats['groups'] == 2
    monkeypatch.setenv('PASTML_AMD_DEVICE_BYTES', '400000')     # room for one character of this tree at a time
    tree2 = read_tree(TREE_NWK)
    chunked = acr(tree2, df.copy(), prediction_method=MPPA, model=F81)
    assert run_tasks.last_stats['groups'] > 2
    for a, b in zip(whole, chunked):
        assert a['character'] == b['character'] and a[LOG_LIKELIHOOD] == b[LOG_LIKELIHOOD]
        assert np.array_equal(a[MARGINAL_PROBABILITIES].values, b[MARGINAL_PROBABILITIES].values)
        assert a[RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(MPPA)] == b[RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(MPPA)]


def test_zero_likelihood_of_one_character_fails_the_batch_cleanly():
    """One character of a batch has no likelihood (masks in conflict across a zero-length branch, the reference's
    edge case of ml.py:139-145) while the optimisers of the others are running: the batch raises the reference's error
    for that character, every optimiser thread ends, and the engine pool serves the next call."""
    import threading
    from pastml_amd import batch as B
    z = load_golden('edge_zero_likelihood')
    flat = FlatForest(z['parent'], z['n_children'], z['first_child'], z['dist'], np.arange(int(z['n_roots'])))
    flat.to_tree_nodes(names=list(z['node_names']))
    states = z['states']
    fs = ForestStats(flat)
    tasks = [B.Task('c{}'.format(i), MPPA, F81Model(states=states, forest_stats=fs, character='c{}'.format(i)),
                    np.ones(len(states)) / len(states)) for i in range(3)]
    with B.CharacterBatch(flat, len(states), 3) as cb:
        cb.initialize_allowed_states()
        cb.masks[1] = B.words_from_masks(z['masks'], len(states))      # the conflicting masks, for the middle character
        tip_masks = z['masks'].copy()
        tip_masks[flat.n_children > 0] = 1                              # the same tips without the conflict
        cb.masks[0] = cb.masks[2] = B.words_from_masks(tip_masks, len(states))
        with pytest.raises(PastMLLikelihoodError) as e:
            B.optimise_group(cb, tasks)
        assert str(e.value) == str(z['error_message'])
    assert not [th for th in threading.enumerate() if th.name.startswith('pastml-opt')]
    tree, results = albania_result(JC)
    assert results[0][LOG_LIKELIHOOD] < 0


def test_groups_spread_over_devices_with_the_same_bits(monkeypatch):
    """
    A plain acr() call takes every visible GPU: groups of characters are cut into one chunk per device and placed
    largest first (batch.run_tasks, visible_devices); all chunks advance in the one optimiser loop, each on its own
    device context.  Here the "devices" are GPU 0 twice: same placement logic, and every character must come out with
    exactly the numbers of the single-device run.
    """
    from pastml_amd.batch import run_tasks
    tree = read_tree(TREE_NWK)
    df = _multi_character_table(tree)
    models = [F81, JC, F81, EFT, F81, JC]
    one = acr(tree, df.copy(), prediction_method=MPPA, model=models)
    assert run_tasks.last_stats['devices'] == [0] and run_tasks.last_stats['groups'] == 3
    monkeypatch.setenv('PASTML_AMD_DEVICES', '0,0')
    monkeypatch.setenv('PASTML_AMD_SPLIT_MIN_WORK', '0')
    tree2 = read_tree(TREE_NWK)
    two = acr(tree2, df.copy(), prediction_method=MPPA, model=models)
    assert run_tasks.last_stats['groups'] == 5     # k = 2: 2 + 1 characters -> 2 chunks, k = 5: 3 -> 2 chunks, k = 3: 1
    for a, b in zip(one, two):
        assert a['character'] == b['character'] and a[LOG_LIKELIHOOD] == b[LOG_LIKELIHOOD]
        assert np.array_equal(a[MARGINAL_PROBABILITIES].values, b[MARGINAL_PROBABILITIES].values)
        for m in (JOINT, MAP, MPPA):
            key = RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(m)
            assert a[key] == b[key]
        assert a[MODEL].sf == b[MODEL].sf and np.array_equal(a[MODEL].frequencies, b[MODEL].frequencies)
    f1, f2 = FlatForest.from_trees([tree]), FlatForest.from_trees([tree2])
    for column in df.columns:
        assert [getattr(n, column) for n in f1.nodes] == [getattr(n, column) for n in f2.nodes]


def test_byte_model_counts_the_units_the_library_schedules(monkeypatch):
    """bench.schedule_bytes restates from the tree which nodes run as two-level and as stacked units; pml_schedule_info
    says what pml_tree_upload made of it: the same numbers -- balanced trees of several sizes, ragged forests with balanced
    clumps and without, the thresholds at their defaults and forced (tests' switches)."""
    import bench
    from pastml_amd import hip, synthetic
    from test_gpu_parity import _forest_with_balanced_clumps
    forests = [synthetic.balanced_forest(12), synthetic.balanced_forest(16), synthetic.balanced_forest(18),
               _forest_with_balanced_clumps(3000, seed=2, clump_frac=0.7), FlatForest.random(30000, seed=1, max_arity=3)]
    for forced in (False, True):
        for var in ('PASTML_HIP_SUPER_MIN', 'PASTML_HIP_STACK_MIN'):
            if forced:
                monkeypatch.setenv(var, '1')
            else:
                monkeypatch.delenv(var, raising=False)
        for flat in forests:
            for k, n_cols in ((64, 64), (64, 1), (33, 16), (20, 64)):
                with hip.Engine(flat, n_cols, k) as eng:
                    on, n2, ns = eng.schedule_info()
                    sb = bench.schedule_bytes(bench.library_forest(eng, flat), k, n_cols)   # (the library's own numbering)
                if forced and not on:
                    continue    # (forced thresholds only matter where the level schedule runs at all)
                assert (n2, ns) == (sb['n_two_level'], sb['n_stacked']), (flat.n_tips, k, n_cols, forced, on, n2, ns)


def test_two_level_launches_where_the_byte_model_expects_them(monkeypatch):
    """bench.schedule_bytes decides from the tree which sweeps run two-level units; the library decides at upload.  The
    profile slots say what ran: slots 3 / 4 (the two-level launches) are used exactly where the model counts such nodes."""
    import bench
    from pastml_amd import hip, synthetic
    flat = synthetic.balanced_forest(12)
    for k, n_cols in ((64, 64), (64, 2), (32, 64), (20, 64)):
        C = n_cols
        expect = bench.schedule_bytes(flat, k, C)['n_two_level'] > 0
        with hip.Engine(flat, C, k) as eng:
            eng.set_models([(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(C)])
            eng.set_tip_states(np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)]))
            eng.profile_enable(True)
            for w in range(5):
                eng.profile_read(w, reset=True)
            eng.marginal_pass(posterior=False, lh=False)
            ran = [eng.profile_read(w)[1] for w in range(5)]
            eng.profile_enable(False)
        assert (ran[3] > 0) == expect and (ran[4] > 0) == expect, (k, C, ran)
