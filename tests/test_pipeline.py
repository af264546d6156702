"""
The file-to-file layer around the path (pastml_amd.pipeline; pastml/acr.py:316-674 minus visualisation,
:695-826, pastml/tree.py:176-222): tree and table readers and input validation on the CPU, the whole pipeline on the
GPU against the tables of the reference's stored Albania run (examples/Albania/data/pastml/MPPA/F81, PastML 1.9.15).
"""
import os

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN
from pastml_amd import pipeline
from pastml_amd.tree import read_tree

DATA = os.path.join(GOLDEN, 'data')
TREE = os.path.join(DATA, 'Albanian.tree.152tax.tre')
TABLE = os.path.join(DATA, 'data.txt')
STORED = os.path.join(DATA, 'albania_pastml')


def test_read_forest_newick_nexus_and_embedded_annotations(tmp_path):
    nwk = tmp_path / 'two.nwk'
    nwk.write_text("((a:1,b:-2)x:0.5,c:1)r;\n(d:1,e:1);\n")
    roots = pipeline.read_forest(str(nwk))
    assert [len(r) for r in roots] == [3, 2]
    assert min(n.dist for r in roots for n in r.traverse()) == 0      # the negative branch became zero
    nex = tmp_path / 't.nex'
    nex.write_text("#NEXUS\nbegin taxa;\nend;\nbegin trees;\n translate\n 1 tipA,\n 2 'tip B',\n 3 tipC;\n"
                   " tree t1 = [&R] ((1:1,2:2)n1:0.5,3:1)root;\n tree t2 = (3:1,(1:1,2:1):1);\nend;\n")
    roots = pipeline.read_forest(str(nex))
    assert [sorted(t.name for t in r) for r in roots] == [['tip B', 'tipA', 'tipC']] * 2
    assert roots[0].children[0].name == 'n1' and roots[0].children[0].dist == 0.5
    ann = tmp_path / 'ann.nwk'
    ann.write_text('((a:1[&&NHX:loc=X|Y],b:2[&loc="Z",date=2001])i1:0.5[&&NHX:loc=X:other=1],c:1)r;')
    root = pipeline.read_forest(str(ann), columns=['loc'])[0]
    got = {n.name: getattr(n, 'loc', None) for n in root.traverse()}
    assert got == {'r': None, 'i1': {'X'}, 'c': None, 'a': {'X', 'Y'}, 'b': {'Z'}}
    with pytest.raises(ValueError):
        empty = tmp_path / 'empty.nwk'
        empty.write_text('')
        pipeline.read_forest(str(empty))


def test_validate_input_checks(tmp_path):
    roots, columns, column2states, parameters, rates = pipeline.validate_input(
        TREE, columns=['Country'], data=TABLE, data_sep=',', parameters={'Country': {'scaling_factor': 3.0}})
    assert columns == ['Country'] and list(column2states['Country']) == ['Africa', 'Albania', 'EastEurope', 'Greece',
                                                                         'WestEurope']
    assert parameters == {'Country': {'scaling_factor': 3.0}} and rates == {}
    names = [n.name for n in roots[0].traverse()]
    assert len(set(names)) == len(names) == 305 and all(names)          # every node named, uniquely
    assert sum(1 for t in roots[0] if getattr(t, 'Country', None)) == 154
    with pytest.raises(ValueError, match='not found among the annotation columns'):
        pipeline.validate_input(TREE, columns=['Nope'], data=TABLE, data_sep=',')
    with pytest.raises(ValueError, match="If you don't provide the metadata file"):
        pipeline.validate_input(TREE)
    # ids that do not match the tree
    bad = tmp_path / 'bad.csv'
    pd.DataFrame({'Country': ['x', 'y']}, index=['nobody', 'noone']).to_csv(bad)
    with pytest.raises(ValueError, match='do not correspond to annotation id column values'):
        pipeline.validate_input(TREE, data=str(bad), data_sep=',')
    # 90 % of the tips unknown
    few = tmp_path / 'few.csv'
    tips = [t.name for t in read_tree(TREE)]
    pd.DataFrame({'Country': ['A', 'B', 'A']}, index=tips[:3]).to_csv(few)
    with pytest.raises(ValueError, match='are unknown'):
        pipeline.validate_input(TREE, data=str(few), data_sep=',')
    # a "character" with a state per tip
    many = tmp_path / 'many.csv'
    pd.DataFrame({'id': tips}, index=tips).to_csv(many)
    with pytest.raises(ValueError, match='unique states'):
        pipeline.validate_input(TREE, data=str(many), data_sep=',')
    # quoted tip names in the tree are matched after stripping the quotes
    quoted = tmp_path / 'q.nwk'
    quoted.write_text("(('t 1':1,'t 2':1):1,'t 3':2);")
    tab = tmp_path / 'q.tab'
    tab.write_text("id\tch\nt 1\tA\nt 2\tB\nt 3\tA\n")
    roots, columns, c2s, _, _ = pipeline.validate_input(str(quoted), data=str(tab))
    assert [getattr(t, 'ch') for t in roots[0]] == [{'A'}, {'B'}, {'A'}]
    assert pipeline.read_annotation_table(str(tab)).columns[0] == 'ch'


def test_visualisation_options_are_refused():
    with pytest.raises(NotImplementedError):
        pipeline.pastml_pipeline(TREE, data=TABLE, data_sep=',', columns=['Country'], html_compressed='map.html')


@pytest.mark.gpu
def test_pipeline_reproduces_the_stored_reference_run(tmp_path):
    """
    tree + table in, the reference's four output files out; the tables agree with the ones PastML 1.9.15 stored for
    the same run (it collapsed zero-length branches, so its internal node names differ: tips and the root are compared
    by name, everything by value) and can be fed back as parameters.
    """
    work = tmp_path / 'out'
    results = pipeline.pastml_pipeline(TREE, data=TABLE, data_sep=',', columns=['Country'], work_dir=str(work))
    files = sorted(os.listdir(work))
    assert files == ['combined_ancestral_states.tab', 'marginal_probabilities.character_Country.model_F81.tab',
                     'named.tree_Albanian.tree.152tax.nwk', 'params.character_Country.method_MPPA.model_F81.tab']
    ours = pd.read_csv(work / files[3], sep='\t', index_col=0)['value']
    ref = pd.read_csv(os.path.join(STORED, files[3]), sep='\t', index_col=0)['value']
    assert set(ours.index) - {'smoothing_factor'} == set(ref.index) - {'smoothing_factor'}
    for key, tol in (('log_likelihood', 5e-4), ('log_likelihood_restricted_MPPA', 5e-4), ('scaling_factor', 5e-3),
                     ('Africa', 5e-4), ('Albania', 5e-4), ('EastEurope', 5e-4), ('Greece', 5e-4), ('WestEurope', 5e-4)):
        assert abs(float(ours[key]) - float(ref[key])) < tol, key
    assert ours['method'] == ref['method'] == 'MPPA' and ours['model'] == ref['model'] == 'F81'
    assert int(ours['num_tips']) == int(ref['num_tips']) == 154
    mp = pd.read_csv(work / files[1], sep='\t', index_col=0)
    mp_ref = pd.read_csv(os.path.join(STORED, files[1]), sep='\t', index_col=0)
    assert mp.index.name == mp_ref.index.name == 'node' and list(mp.columns) == list(mp_ref.columns)
    # tips on zero-length branches are left out: 1.9.15 merged them with their parents, the current code (and we)
    # keep them and alter their allowed states instead (ml.py:352-387)
    tips = [t.name for t in read_tree(TREE) if t.dist > 0]
    assert len(tips) > 140
    shared = tips + ['ROOT']
    np.testing.assert_allclose(mp.loc[shared].values, mp_ref.loc[shared].values, atol=2e-3)
    states = pd.read_csv(work / files[0], sep='\t', index_col=0)
    states_ref = pd.read_csv(os.path.join(STORED, files[0]), sep='\t', index_col=0)
    assert list(states.columns) == list(states_ref.columns) == ['Country']
    assert len(set(states.index)) == 305
    for name in shared:
        assert sorted(np.atleast_1d(states.loc[name, 'Country'])) == sorted(np.atleast_1d(states_ref.loc[name, 'Country']))
    named = read_tree(str(work / files[2]))
    assert sorted(n.name for n in named.traverse()) == sorted(states.index.unique())
    # the parameter table goes back in: nothing left to optimise, same likelihood
    again = pipeline.pastml_pipeline(TREE, data=TABLE, data_sep=',', columns=['Country'], work_dir=str(tmp_path / 'again'),
                                     parameters={'Country': str(work / files[3])})
    assert again[0]['model'].get_num_params() == 0
    np.testing.assert_allclose(again[0]['log_likelihood'], results[0]['log_likelihood'], rtol=1e-12)


def test_combined_states_table_array_writer_equals_the_per_node_loop(tmp_path):
    """
    serialize_predicted_states expands packed state-set columns array-wise; the rows must be those of the reference's
    per-node loop (pastml/acr.py:831-858): a node with several states takes several lines, every column's states in
    ascending order, exhausted columns empty -- also with multi-state nodes in several columns and unsorted state lists.
    """
    from pastml_amd.tree import FlatForest, StateSetColumn, get_flat_forest
    from pastml_amd.hip import pack_masks
    flat0 = FlatForest.random(40, seed=3, max_arity=3, n_trees=2)
    roots = flat0.to_tree_nodes(names=['n{}'.format(i) for i in range(flat0.n_nodes)])
    flat = get_flat_forest(roots)
    rng = np.random.default_rng(1)
    spec = {'x': np.array(['b', 'a', 'c']), 'y': np.array(['s10', 's2', 's1', 's3', 's0'])}
    expect = {}
    for c, states in spec.items():
        masks = (rng.random((flat.n_nodes, len(states))) < 0.4).astype(np.int8)
        masks[masks.sum(axis=1) == 0, 0] = 1
        flat.set_column(c, StateSetColumn(pack_masks(masks, len(states)), states))
        expect[c] = [sorted(states[m.astype(bool)]) for m in masks]
    out = tmp_path / 'combined.tab'
    pipeline.serialize_predicted_states(['x', 'y'], str(out), roots)
    order = np.lexsort((np.arange(flat.n_nodes), flat.tree_id))
    lines = ['node\tx\ty']
    for i in order:
        values = [expect[c][i] for c in ('x', 'y')]
        for line in range(max(len(v) for v in values)):
            lines.append('{}\t{}'.format(flat.nodes[i].name, '\t'.join(v[line] if line < len(v) else '' for v in values)))
    assert out.read_text() == '\n'.join(lines) + '\n'


@pytest.mark.gpu
def test_pipeline_at_size(tmp_path):
    """
    SURVEY 8f-4 at the size it names: pastml_pipeline -- newick and table readers, validation, naming, one batched acr()
    (F81 + MPPA with parameter optimisation), the result writers -- on a 262 144-tip tree with two characters, inside a
    time budget; outputs complete and consistent with each other.  (At this size the number of scenarios has tens of
    thousands of digits: the parameter table writes it as mantissa and exponent.)
    """
    import time
    L, C, k = 18, 2, 4
    rng = np.random.default_rng(3)
    level = ['t%d:%.4f' % (i, rng.uniform(0.01, 0.2)) for i in range(2 ** L)]
    while len(level) > 1:
        level = ['(%s,%s):%.4f' % (level[i], level[i + 1], rng.uniform(0.01, 0.2)) for i in range(0, len(level), 2)]
    nwk, tab = tmp_path / 'tree.nwk', tmp_path / 'data.tab'
    nwk.write_text(level[0] + ';')
    states = np.array(['s%d' % s for s in range(k)])
    table = pd.DataFrame({'char%d' % c: states[rng.integers(0, k, size=2 ** L)] for c in range(C)},
                         index=['t%d' % i for i in range(2 ** L)])
    table.to_csv(tab, sep='\t', index_label='id')
    t0 = time.perf_counter()
    results = pipeline.pastml_pipeline(str(nwk), data=str(tab), work_dir=str(tmp_path / 'out'))
    seconds = time.perf_counter() - t0
    print('pastml_pipeline on {} tips x {} characters: {:.1f} s'.format(2 ** L, C, seconds))
    assert seconds < 60
    n_nodes = 2 ** (L + 1) - 1
    assert [r['character'] for r in results] == ['char0', 'char1']
    combined = pd.read_csv(tmp_path / 'out' / 'combined_ancestral_states.tab', sep='\t', index_col=0, dtype=str,
                           keep_default_na=False)
    assert combined.index.nunique() == n_nodes and len(combined) >= n_nodes
    tips = combined.loc[['t0', 't77', 't{}'.format(2 ** L - 1)]]
    assert tips['char0'].tolist() == table.loc[tips.index, 'char0'].tolist()       # observed tips keep their states
    for r in results:
        mp = pd.read_csv(tmp_path / 'out' / 'marginal_probabilities.character_{}.model_F81.tab'.format(r['character']),
                         sep='\t', index_col=0, nrows=2000)
        np.testing.assert_allclose(mp.values.sum(axis=1), 1, rtol=1e-9)
        params = pd.read_csv(tmp_path / 'out' / 'params.character_{}.method_MPPA.model_F81.tab'.format(r['character']),
                             sep='\t', index_col=0)['value']
        assert abs(float(params['log_likelihood']) - r['log_likelihood']) < 1e-6 * abs(r['log_likelihood'])
        assert 'e+' in params['num_scenarios'] and int(params['num_tips']) == 2 ** L
        # the nodes MPPA left unresolved are the ones that take more than one line
        multi = combined[r['character']].replace('', np.nan).groupby(level=0).count()
        assert int((multi > 1).sum()) == int(params['num_unresolved_nodes'])
    named = read_tree(str(tmp_path / 'out' / 'named.tree_tree.nwk'))
    assert len(named) == 2 ** L
