"""
pastml_amd.parsimony (level-wise array passes on the flat forest) against the reference's pastml/parsimony.py
(tests/golden/parsimony.npz: Albanian tree, a tree with polytomies / zero branches / missing and multi-state tips, a
forest of two trees), and the ALL meta-method of ml_acr, which evaluates the likelihood restricted to each parsimonious
reconstruction (GPU).
"""
import numpy as np
import pytest

from conftest import load_golden
from pastml_amd.parsimony import parsimonious_acr, STEPS, MP, DOWNPASS, ACCTRAN, DELTRAN
from pastml_amd.tree import FlatForest


def forest_of(z, prefix):
    flat = FlatForest(z[prefix + 'parent'], z[prefix + 'n_children'], z[prefix + 'first_child'], z[prefix + 'dist'],
                      np.arange(int(z[prefix + 'n_roots'])))
    roots = flat.to_tree_nodes(names=list(z[prefix + 'node_names']))
    return flat, roots


@pytest.mark.parametrize('prefix,character', [('alb_', 'Country'), ('poly_', 'ch'), ('forest_', 'ch')])
def test_parsimony_matches_reference(prefix, character):
    z = load_golden('parsimony')
    states = z[prefix + 'states']
    ann = z[prefix + 'annotation']
    for method in (MP, DOWNPASS, ACCTRAN, DELTRAN):
        flat, roots = forest_of(z, prefix)
        for i, n in enumerate(flat.nodes):
            if ann[i].any():
                n.add_feature(character, set(states[ann[i].astype(bool)]))
        results = parsimonious_acr(roots, character, method, states, flat.n_nodes, flat.n_tips)
        expected = [m for m in (ACCTRAN, DOWNPASS, DELTRAN) if method in (MP, m)]
        assert [r['method'] for r in results] == expected
        for res in results:
            tag = '{}{}_{}_'.format(prefix, method, res['method'])
            assert res['character'] == str(z[tag + 'character'])
            sel = np.zeros_like(ann)
            s2i = {s: i for i, s in enumerate(states)}
            for i, n in enumerate(flat.nodes):
                for s in getattr(n, res['character']):
                    sel[i, s2i[s]] = 1
            assert np.array_equal(sel, z[tag + 'selected']), tag
            assert res[STEPS] == int(z[tag + 'steps']), tag
            assert float(res['num_scenarios']) == float(z[tag + 'num_scenarios'])
            assert res['num_unresolved_nodes'] == int(z[tag + 'num_unresolved_nodes'])
            assert res['num_states_per_node_avg'] == float(z[tag + 'num_states_per_node_avg'])
            assert res['num_nodes'] == flat.n_nodes and res['num_tips'] == flat.n_tips


def test_acr_accepts_parsimony_and_copy_without_a_gpu():
    """MP methods and COPY never touch the device: acr() serves them on any box."""
    import os
    import pandas as pd
    from conftest import GOLDEN
    from pastml_amd.acr import acr, COPY
    from pastml_amd.tree import read_tree
    z = load_golden('parsimony')
    tree = read_tree(os.path.join(GOLDEN, 'data', 'Albanian.tree.152tax.tre'))
    df = pd.read_csv(os.path.join(GOLDEN, 'data', 'data.txt'), index_col=0, header=0)[['Country']]
    df['Again'] = df['Country']
    res = acr(tree, df, prediction_method=[DOWNPASS, COPY])
    assert [(r['character'], r['method']) for r in res] == [('Country', DOWNPASS), ('Again', COPY)]
    assert res[0][STEPS] == int(z['alb_DOWNPASS_DOWNPASS_steps'])
    assert list(res[1]['states']) == list(z['alb_states'])
    with pytest.raises(ValueError, match='is unknown'):
        acr(tree, df, prediction_method='FITCH')


@pytest.mark.gpu
def test_all_meta_method_matches_reference():
    """ml_acr's ALL (ml.py:718-733): JOINT, MAP, the three parsimonious reconstructions, MPPA -- in this order -- with
    the likelihood restricted to each; the parsimonious selections the reference found inconsistent with the likelihood
    (a zero-length branch without a common state) have no restricted likelihood here either."""
    import os
    import pandas as pd
    from conftest import GOLDEN
    from pastml_amd.acr import acr
    from pastml_amd.tree import read_tree
    z = load_golden('parsimony')
    tree = read_tree(os.path.join(GOLDEN, 'data', 'Albanian.tree.152tax.tre'))
    df = pd.read_csv(os.path.join(GOLDEN, 'data', 'data.txt'), index_col=0, header=0)[['Country']]
    res = acr(tree, df, prediction_method='ALL', model='F81')
    assert [r['method'] for r in res] == list(z['all_methods'])
    assert [r['character'] for r in res] == list(z['all_characters'])
    last = res[-1]
    assert sorted(last.keys()) == list(z['all_mppa_keys'])
    for key in z.files:
        if key.startswith('all_log_likelihood'):
            np.testing.assert_allclose(last[key[4:]], float(z[key]), rtol=1e-6, err_msg=key)
    # the parsimonious results carry steps and statistics, no likelihood
    for r in res[2:5]:
        assert STEPS in r and 'log_likelihood' not in r
        tag = 'alb_MP_{}_'.format(r['method'])
        assert r[STEPS] == int(z[tag + 'steps'])
        assert hasattr(tree, r['character'])
