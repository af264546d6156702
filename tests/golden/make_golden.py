#!/usr/bin/env python3
"""
Generates the golden vectors under tests/golden/ by running the REAL reference (evolbioinfo/pastml, imported
unmodified from /root/reference) in this container.  Run as:

    python3 -B tests/golden/make_golden.py [case ...]

Nothing of the reference is copied: ete3 / Bio / itolapi (absent here) are replaced in sys.modules by thin stand-ins
(our own TreeNode plays ete3.Tree), the reference package is imported from /root/reference, its functions are called
and inputs + outputs are stored as .npz.  The small data files the reference's own tests use
(tests/data/Albanian.tree.152tax.tre, data.txt, the two simulated nucleotide trees) are copied as fixtures.

All arrays indexed by node follow the forest-wide level order of pastml_amd.tree.FlatForest (== tree.traverse()).
"""
import os
import shutil
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)

import numpy as np
import pandas as pd

from pastml_amd import tree as our_tree
from pastml_amd import synthetic


def install_stubs():
    ete3 = types.ModuleType('ete3')
    ete3.Tree = our_tree.TreeNode
    ete3.TreeNode = our_tree.TreeNode
    parser = types.ModuleType('ete3.parser')
    newick = types.ModuleType('ete3.parser.newick')
    newick.NewickError = our_tree.NewickError
    parser.newick = newick
    ete3.parser = parser
    sys.modules.update({'ete3': ete3, 'ete3.parser': parser, 'ete3.parser.newick': newick})
    bio = types.ModuleType('Bio')
    phylo = types.ModuleType('Bio.Phylo')
    nio = types.ModuleType('Bio.Phylo.NewickIO')
    import io
    nio.StringIO = io.StringIO
    phylo.NewickIO = nio
    phylo.write = lambda *a, **k: None
    phylo.parse = lambda *a, **k: iter(())
    bio.Phylo = phylo
    sys.modules.update({'Bio': bio, 'Bio.Phylo': phylo, 'Bio.Phylo.NewickIO': nio})
    itol = types.ModuleType('itolapi')
    itol.Itol = object
    sys.modules['itolapi'] = itol


install_stubs()
sys.path.insert(0, REF)
import pastml  # noqa: E402
from pastml import ml as rml  # noqa: E402
from pastml.acr import acr as racr  # noqa: E402
from pastml.annotation import ForestStats as RForestStats  # noqa: E402
from pastml.models.F81Model import F81Model as RF81  # noqa: E402
from pastml.models.JCModel import JCModel as RJC  # noqa: E402
from pastml.models.EFTModel import EFTModel as REFT  # noqa: E402
from pastml.models.HKYModel import HKYModel as RHKY  # noqa: E402
from pastml.models.JTTModel import JTTModel as RJTT, JTT_STATES  # noqa: E402
from pastml.models.CustomRatesModel import CustomRatesModel as RCR  # noqa: E402

assert pastml.__file__.startswith(REF)

DATA = os.path.join(HERE, 'data')
os.makedirs(DATA, exist_ok=True)


def feat(character, name):
    return '{}_{}'.format(character, name)


def copy_data():
    for f in ('Albanian.tree.152tax.tre', 'data.txt',
              'tree.152taxa.sf_0.5.A_0.6.C_0.15.G_0.2.T_0.05.nwk',
              'tree.152taxa.sf_0.5.A_0.6.C_0.15.G_0.2.T_0.05.pastml.tab',
              'tree.152taxa.sf_0.5.A_0.25.C_0.25.G_0.25.T_0.25.nwk',
              'tree.152taxa.sf_0.5.A_0.25.C_0.25.G_0.25.T_0.25.pastml.tab'):
        shutil.copy(os.path.join(REF, 'tests', 'data', f), os.path.join(DATA, f))


# ---------------------------------------------------------------------------------------------------------------------
def model_arrays(model):
    out = dict(model_name=model.name, sf=float(model.sf), tau=float(model.tau), tau_factor=float(model._tau_factor),
               states=np.array(model.states, dtype=str))
    if hasattr(model, 'frequencies'):
        out['frequencies'] = np.array(model.frequencies, dtype=np.float64)
    if hasattr(model, 'kappa'):
        out['kappa'] = float(model.kappa)
    if hasattr(model, 'D_DIAGONAL'):
        out['eig_d'] = np.array(model.D_DIAGONAL)
        out['eig_A'] = np.array(model.A)
        out['eig_Ainv'] = np.array(model.A_INV)
        out['rate_matrix'] = np.array(model.rate_matrix)
    return out


def collect(nodes, name, k=None, dtype=np.float64):
    vals = [getattr(n, name) for n in nodes]
    return np.array(vals, dtype=dtype)


def sweep_capture(forest, character, model, force_joint=True, prefix=''):
    """
    Re-runs the steps of pastml.ml.ml_acr (ml.py:640-750) for fixed parameters with the reference's own functions,
    capturing every intermediate the kernels must reproduce.
    """
    flat = our_tree.FlatForest.from_trees(forest)
    nodes = flat.nodes
    states = model.states
    k = len(states)
    out = {}
    A = feat(character, rml.ALLOWED_STATES)

    for tree in forest:
        rml.initialize_allowed_states(tree, character, states)
    out['masks_initial'] = collect(nodes, A, dtype=np.int8)

    # marginal log-likelihood with alteration (what the optimiser evaluates)
    out['loglik'] = sum(rml.get_bottom_up_loglikelihood(tree=t, character=character, model=model,
                                                        is_marginal=True, alter=True) for t in forest)
    # joint sweep
    out['loglik_joint'] = sum(rml.get_bottom_up_loglikelihood(tree=t, character=character, model=model,
                                                              is_marginal=False, alter=True) for t in forest)
    out['masks_after_joint_sweep'] = collect(nodes, A, dtype=np.int8)
    jt = np.full((len(nodes), k), -1, dtype=np.int64)
    JS = feat(character, rml.BU_LH_JOINT_STATES)
    for i, n in enumerate(nodes):
        if hasattr(n, JS):
            jt[i] = getattr(n, JS)
    out['joint_table'] = jt
    out['bu_joint'] = collect(nodes, feat(character, rml.BU_LH))
    out['bu_joint_sf'] = collect(nodes, feat(character, rml.BU_LH_SF))
    for t in forest:
        rml.choose_ancestral_states_joint(t, character, states, model.frequencies)
    out['joint_state'] = collect(nodes, feat(character, rml.JOINT_STATE), dtype=np.int64)

    # marginal: explicit alteration, BU (alter=False), TD, marginals
    altered_all = []
    for t in forest:
        rml.initialize_allowed_states(t, character, states)
        altered = []
        if 0 == model.tau:
            altered = rml.alter_zero_node_allowed_states(t, character)
        altered_all.extend(altered)
    out['masks_altered'] = collect(nodes, A, dtype=np.int8)
    ids = {id(n): i for i, n in enumerate(nodes)}
    out['altered_nodes'] = np.array(sorted(ids[id(n)] for n in altered_all), dtype=np.int32)
    for t in forest:
        rml.get_bottom_up_loglikelihood(tree=t, character=character, is_marginal=True, model=model, alter=False)
        rml.calculate_top_down_likelihood(t, character, model=model)
        rml.calculate_marginal_likelihoods(t, character, model.frequencies, clean_up=False)
    out['bu'] = collect(nodes, feat(character, rml.BU_LH))
    out['bu_sf'] = collect(nodes, feat(character, rml.BU_LH_SF))
    out['td'] = collect(nodes, feat(character, rml.TD_LH))
    out['td_sf'] = collect(nodes, feat(character, rml.TD_LH_SF))
    out['lh'] = collect(nodes, feat(character, rml.LH))
    out['lh_sf'] = collect(nodes, feat(character, rml.LH_SF))
    mps = [rml.convert_likelihoods_to_probabilities(t, character, states) for t in forest]
    mp = pd.concat(mps) if len(mps) > 1 else mps[0]
    # forest-wide level order
    out['posterior'] = np.array([mp.loc[n.name].values if mp.index.is_unique else None for n in nodes]) \
        if mp.index.is_unique else mp.values
    if altered_all:
        rml.unalter_zero_node_allowed_states(altered_all, character)
    for t in forest:
        rml.choose_ancestral_states_map(t, character, states)
    out['masks_map'] = collect(nodes, A, dtype=np.int8)
    out['loglik_restricted_MAP'] = sum(rml.get_bottom_up_loglikelihood(tree=t, character=character, model=model,
                                                                       is_marginal=True, alter=True) for t in forest)
    ns, nun, nst = 1, 0, 0
    for t in forest:
        a, b, c = rml.choose_ancestral_states_mppa(t, character, states, force_joint=force_joint)
        ns, nun, nst = ns * a, nun + b, nst + c
    out['masks_mppa'] = collect(nodes, A, dtype=np.int8)
    out['mppa_log_num_scenarios'] = float(np.sum(np.log(out['masks_mppa'].sum(axis=1).astype(np.float64))))
    out['mppa_num_unresolved'] = nun
    out['mppa_num_states'] = nst
    out['loglik_restricted_MPPA'] = sum(rml.get_bottom_up_loglikelihood(tree=t, character=character, model=model,
                                                                        is_marginal=True, alter=True) for t in forest)
    out['force_joint'] = force_joint
    return flat, {prefix + k_: v for k_, v in out.items()}


def tree_arrays(flat):
    return dict(parent=flat.parent, first_child=flat.first_child, n_children=flat.n_children, dist=flat.dist,
                n_roots=len(flat.roots), node_names=np.array([n.name for n in flat.nodes], dtype=str))


def forest_stats_arrays(fs):
    return dict(fs_avg_nonzero_brlen=fs.avg_nonzero_brlen, fs_num_nodes=fs.num_nodes, fs_num_tips=fs.num_tips,
                fs_forest_length=fs.forest_length)


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('wrote', path, '{:.1f} KB'.format(os.path.getsize(path) / 1024))


# ---------------------------------------------------------------------------------------------------------------------
# cases
# ---------------------------------------------------------------------------------------------------------------------
class FS(object):
    """forest_stats stand-in for P(t)-only cases."""

    def __init__(self, length=10., nodes=20, tips=11, avg=0.5):
        self.forest_length, self.num_nodes, self.num_tips, self.avg_nonzero_brlen = length, nodes, tips, avg


def case_pij():
    rng = np.random.default_rng(7)
    ts = np.array([0, 1e-6, 1e-3, 0.1, 1, 10], dtype=np.float64)
    out = {'ts': ts}
    idx = 0
    names = []

    def add(label, model):
        nonlocal idx
        P = np.array([model.get_Pij_t(t) for t in ts])
        pre = 'c{}_'.format(idx)
        names.append(label)
        for k_, v in model_arrays(model).items():
            out[pre + k_] = v
        out[pre + 'P'] = P
        idx += 1

    for tau in (0, 0.05):
        fs = FS()
        for k in (4, 5, 64):
            pi = rng.dirichlet(np.ones(k))
            add('F81_k{}_tau{}'.format(k, tau),
                RF81(states=synthetic.state_names(k), forest_stats=fs, sf=1.3, frequencies=pi, tau=tau))
        add('JC_k4_tau{}'.format(tau), RJC(states=synthetic.state_names(4), forest_stats=fs, sf=0.7, tau=tau))
        add('JC_k1_tau{}'.format(tau), RJC(states=synthetic.state_names(1), forest_stats=fs, sf=0.7, tau=tau))
        for kappa in (1., 4.):
            add('HKY_kappa{}_tau{}'.format(kappa, tau),
                RHKY(forest_stats=fs, sf=2., frequencies=rng.dirichlet(np.ones(4)), kappa=kappa, tau=tau))
        add('JTT_tau{}'.format(tau), RJTT(forest_stats=fs, sf=1.1, tau=tau))
        k = 10
        R = rng.uniform(0.1, 3, size=(k, k))
        R = np.triu(R, 1)
        R = R + R.T
        add('CUSTOM_k10_tau{}'.format(tau),
            RCR(forest_stats=fs, sf=0.9, frequencies=rng.dirichlet(np.ones(k)), rate_matrix=R,
                states=synthetic.state_names(k), tau=tau))
    out['labels'] = np.array(names, dtype=str)
    save('pij', **out)


def albania_inputs():
    tree = our_tree.read_tree(os.path.join(DATA, 'Albanian.tree.152tax.tre'))
    df = pd.read_csv(os.path.join(DATA, 'data.txt'), index_col=0, header=0)[['Country']]
    return tree, df


def case_albania():
    """Albania 152-tip tree, Country: optimised (acr() end to end) + fixed-parameter intermediates, F81/JC/EFT."""
    for model_name in ('F81', 'JC', 'EFT'):
        tree, df = albania_inputs()
        res = racr(tree, df, prediction_method='MPPA', model=model_name, threads=1)[0]
        model = res['model']
        flat = our_tree.FlatForest.from_trees([tree])
        nodes = flat.nodes
        out = dict(tree_arrays(flat))
        out.update(forest_stats_arrays(model.forest_stats))
        out.update({'opt_' + k: v for k, v in model_arrays(model).items()})
        out['opt_loglik'] = res['log_likelihood']
        out['opt_loglik_restricted_JOINT'] = res['log_likelihood_restricted_JOINT']
        out['opt_loglik_restricted_MAP'] = res['log_likelihood_restricted_MAP']
        out['opt_loglik_restricted_MPPA'] = res['log_likelihood_restricted_MPPA']
        out['opt_num_scenarios'] = float(res['num_scenarios'])
        out['opt_num_unresolved_nodes'] = res['num_unresolved_nodes']
        out['opt_num_states_per_node_avg'] = res['num_states_per_node_avg']
        mp = res['marginal_probabilities']
        out['opt_posterior'] = mp.loc[[n.name for n in nodes]].values
        out['opt_posterior_index'] = np.array(mp.index, dtype=str)
        states = model.states
        s2i = {s: i for i, s in enumerate(states)}
        sel = np.zeros((len(nodes), len(states)), dtype=np.int8)
        for i, n in enumerate(nodes):
            for s in getattr(n, 'Country'):
                sel[i, s2i[s]] = 1
        out['opt_selected_mppa'] = sel
        out['opt_lh'] = collect(nodes, 'Country_LIKELIHOOD')
        out['opt_lh_sf'] = collect(nodes, 'Country_LIKELIHOOD_SF')
        out['opt_joint_state'] = collect(nodes, 'Country_JOINT_STATE', dtype=np.int64)

        # tip annotation as given to acr (sets of state names -> mask), observed frequencies
        tree2, df2 = albania_inputs()
        from pastml.annotation import preannotate_forest
        preannotate_forest([tree2], df=df2)
        flat2 = our_tree.FlatForest.from_trees([tree2])
        ann = np.zeros((flat2.n_nodes, len(states)), dtype=np.int8)
        for i, n in enumerate(flat2.nodes):
            for s in getattr(n, 'Country', set()):
                ann[i, s2i[s]] = 1
        out['annotation'] = ann
        from pastml.acr import calculate_observed_freqs
        _, obs, _ = calculate_observed_freqs('Country', [tree2], states)
        out['observed_frequencies'] = obs

        # fixed-parameter intermediates at the optimum
        model.freeze()
        _, cap = sweep_capture([tree2], 'Country', model, force_joint=True, prefix='fix_')
        out.update(cap)
        # and with tau > 0 (no alteration)
        tree3, df3 = albania_inputs()
        preannotate_forest([tree3], df=df3)
        fs3 = RForestStats([tree3])
        kwargs = dict(states=states, forest_stats=fs3, sf=float(model.sf), tau=0.01)
        if model_name == 'F81':
            m3 = RF81(frequencies=np.array(model.frequencies), **kwargs)
        elif model_name == 'JC':
            m3 = RJC(**kwargs)
        else:
            m3 = REFT(observed_frequencies=obs, **kwargs)
        m3.freeze()
        _, cap = sweep_capture([tree3], 'Country', m3, force_joint=False, prefix='tau_')
        out.update(cap)
        out.update({'tau_' + k: v for k, v in model_arrays(m3).items()})
        save('albania_{}'.format(model_name), **out)


def annotate_synthetic(flat, roots, character, k, char_index, missing_frac=0.0, seed=0):
    states = synthetic.state_names(k)
    tips_states = synthetic.tip_states(flat.n_tips, k, char_index)
    rng = np.random.default_rng(seed)
    for j, t in enumerate(flat.tips):
        node = flat.nodes[t]
        if missing_frac and rng.random() < missing_frac:
            continue
        node.add_feature(character, {states[tips_states[j]]})
    return states, tips_states


def case_synthetic_small():
    """Balanced trees at reduced size, every intermediate stored (cfg2/cfg3/cfg4 shapes)."""
    # cfg2 shape: JC k=4
    for tag, n_levels, k, model_kind in (('jc_k4_L10', 10, 4, 'JC'), ('f81_k64_L8', 8, 64, 'F81'),
                                        ('jtt_k20_L8', 8, 20, 'JTT'), ('hky_L8', 8, 4, 'HKY'),
                                        ('f81_k5_L9', 9, 5, 'F81'), ('f81_k67_L5', 5, 67, 'F81'),
                                        ('f81_k130_L4', 4, 130, 'F81')):
        flat = synthetic.balanced_forest(n_levels)
        roots = flat.to_tree_nodes()
        fs = RForestStats(roots)
        if model_kind == 'JTT':
            states = JTT_STATES
            tips_states = synthetic.tip_states(flat.n_tips, 20, 0)
            for j, t in enumerate(flat.tips):
                flat.nodes[t].add_feature('c0', {states[tips_states[j]]})
            model = RJTT(forest_stats=fs, sf=1.)
        elif model_kind == 'HKY':
            from pastml.models.HKYModel import HKY_STATES
            states = HKY_STATES
            tips_states = synthetic.tip_states(flat.n_tips, 4, 0)
            for j, t in enumerate(flat.tips):
                flat.nodes[t].add_feature('c0', {states[tips_states[j]]})
            model = RHKY(forest_stats=fs, sf=1., frequencies=synthetic.f81_frequencies(4, 0), kappa=3.)
        else:
            states, tips_states = annotate_synthetic(flat, roots, 'c0', k, 0)
            if model_kind == 'JC':
                model = RJC(states=states, forest_stats=fs, sf=1.)
            else:
                model = RF81(states=states, forest_stats=fs, sf=1., frequencies=synthetic.f81_frequencies(k, 0))
        model.freeze()
        flat2, cap = sweep_capture(roots, 'c0', model)
        out = dict(tree_arrays(flat2))
        out.update(forest_stats_arrays(fs))
        out.update(model_arrays(model))
        out.update(cap)
        out['tip_states'] = tips_states
        out['n_levels'] = n_levels
        save('synthetic_' + tag, **out)


def case_synthetic_large():
    """cfg2 at full size (65 536 tips, JC k=4): lnL + strided samples; F81 k=64 at 16 384 tips for 2 characters."""
    stride = 4099
    flat = synthetic.balanced_forest(16)
    roots = flat.to_tree_nodes()
    fs = RForestStats(roots)
    states, tips_states = annotate_synthetic(flat, roots, 'c0', 4, 0)
    model = RJC(states=states, forest_stats=fs, sf=1.)
    model.freeze()
    flat2, cap = sweep_capture(roots, 'c0', model)
    out = dict(forest_stats_arrays(fs))
    out.update(model_arrays(model))
    sample = np.arange(0, flat.n_nodes, stride)
    out['sample'] = sample
    for key, v in cap.items():
        if isinstance(v, np.ndarray) and v.ndim >= 1 and len(v) == flat.n_nodes:
            out[key] = v[sample]
        else:
            out[key] = v
    out['n_levels'] = 16
    save('synthetic_cfg2_full', **out)

    flat = synthetic.balanced_forest(14)
    for c in (0, 1):
        roots = flat.to_tree_nodes()
        fs = RForestStats(roots)
        states, tips_states = annotate_synthetic(flat, roots, 'c', 64, c)
        model = RF81(states=states, forest_stats=fs, sf=1., frequencies=synthetic.f81_frequencies(64, c))
        model.freeze()
        flat2, cap = sweep_capture(roots, 'c', model)
        out = dict(forest_stats_arrays(fs))
        out.update(model_arrays(model))
        sample = np.arange(0, flat.n_nodes, 257)
        out['sample'] = sample
        for key, v in cap.items():
            if isinstance(v, np.ndarray) and v.ndim >= 1 and len(v) == flat.n_nodes:
                out[key] = v[sample]
            else:
                out[key] = v
        out['n_levels'] = 14
        out['character'] = c
        save('synthetic_cfg4_L14_c{}'.format(c), **out)


def case_edge():
    """Polytomies, zero branches with conflicting annotations, missing / multi-state tips, forests, tau > 0."""
    rng = np.random.default_rng(11)
    for tag, kwargs, k, tau in (('poly', dict(n_tips=60, seed=3, max_arity=5, zero_frac=0.0), 6, 0),
                                ('zero', dict(n_tips=80, seed=4, max_arity=3, zero_frac=0.3), 4, 0),
                                ('zero_tau', dict(n_tips=80, seed=4, max_arity=3, zero_frac=0.3), 4, 0.02),
                                ('forest', dict(n_tips=90, seed=5, max_arity=4, zero_frac=0.15, n_trees=3), 5, 0)):
        flat = our_tree.FlatForest.random(**kwargs)
        roots = [flat.nodes[r] for r in flat.roots]
        states = synthetic.state_names(k)
        # tips: 10% missing, 10% two states; a few internal nodes annotated too
        for i, n in enumerate(flat.nodes):
            u = rng.random()
            if n.is_leaf():
                if u < 0.1:
                    continue
                if u < 0.2:
                    n.add_feature('ch', set(rng.choice(states, size=2, replace=False)))
                else:
                    n.add_feature('ch', {states[int(rng.integers(k))]})
            elif u < 0.08:
                n.add_feature('ch', {states[int(rng.integers(k))]})
        # the annotation as given (sets of states per node), before any state selection overwrites the feature
        ann = np.zeros((flat.n_nodes, k), dtype=np.int8)
        s2i = {s: i for i, s in enumerate(states)}
        for i, n in enumerate(flat.nodes):
            for s in getattr(n, 'ch', set()):
                ann[i, s2i[s]] = 1
        fs = RForestStats(roots)
        model = RF81(states=states, forest_stats=fs, frequencies=rng.dirichlet(np.ones(k)), tau=tau)
        model.freeze()
        out = {}
        try:
            flat2, cap = sweep_capture(roots, 'ch', model)
            out.update(cap)
            out['raised'] = False
        except rml.PastMLLikelihoodError as e:
            flat2 = our_tree.FlatForest.from_trees(roots)
            out['raised'] = True
            out['error_message'] = str(e)
        out.update(tree_arrays(flat2))
        out.update(forest_stats_arrays(fs))
        out.update(model_arrays(model))
        out['annotation'] = ann
        save('edge_' + tag, **out)

    # non-intersecting states across a zero-length internal branch whose nodes are both annotated ... with tau == 0
    # the alteration rescues it; a genuine zero likelihood needs a zero branch to an *unannotated-cluster* conflict:
    t = our_tree.read_tree('((a:0.1,b:0.2)i1:0,(c:0.1,d:0.3)i2:0.2)r:0;')
    for n in t.traverse():
        n.del_feature('ch')
    states = synthetic.state_names(3)
    byname = {n.name: n for n in t.traverse()}
    byname['a'].add_feature('ch', {states[0]})
    byname['b'].add_feature('ch', {states[0]})
    byname['c'].add_feature('ch', {states[1]})
    byname['d'].add_feature('ch', {states[2]})
    fs = RForestStats([t])
    model = RF81(states=states, forest_stats=fs, frequencies=np.array([0.2, 0.3, 0.5]))
    model.freeze()
    # restrict the root to state 1 and i1 (zero branch below the root) to state 0 *after* initialisation, without the
    # annotation feature, so that alteration does not see them
    rml.initialize_allowed_states(t, 'ch', states)
    byname['r'].add_feature('ch_ALLOWED_STATES', np.array([0, 1, 0]))
    byname['i1'].add_feature('ch_ALLOWED_STATES', np.array([1, 0, 0]))
    flat = our_tree.FlatForest.from_trees([t])
    masks = collect(flat.nodes, 'ch_ALLOWED_STATES', dtype=np.int8)
    try:
        rml.get_bottom_up_loglikelihood(t, 'ch', model, is_marginal=True, alter=True)
        raised, msg = False, ''
    except rml.PastMLLikelihoodError as e:
        raised, msg = True, str(e)
    out = dict(tree_arrays(flat))
    out.update(model_arrays(model))
    out.update(masks=masks, raised=raised, error_message=msg)
    save('edge_zero_likelihood', **out)


def case_hky_nucleotide():
    """The reference's simulated nucleotide tree (tests/HKYF81Test.py): optimised F81 and HKY results."""
    tab = 'tree.152taxa.sf_0.5.A_0.6.C_0.15.G_0.2.T_0.05'
    df = pd.read_csv(os.path.join(DATA, tab + '.pastml.tab'), index_col=0, header=0, sep='\t')[['ACR']]
    out = {}
    for label, kw in (('f81', dict(model='F81')), ('hky_k1', dict(model='HKY', column2parameters={'ACR': {'kappa': 1}})),
                      ('hky', dict(model='HKY'))):
        tree = our_tree.read_tree(os.path.join(DATA, tab + '.nwk'))
        res = racr(tree, df.copy(), prediction_method='MPPA', threads=1, **kw)[0]
        model = res['model']
        out.update({label + '_' + k: v for k, v in model_arrays(model).items()})
        out[label + '_loglik'] = res['log_likelihood']
        out[label + '_loglik_restricted_MPPA'] = res['log_likelihood_restricted_MPPA']
        flat = our_tree.FlatForest.from_trees([tree])
        out[label + '_posterior'] = res['marginal_probabilities'].loc[[n.name for n in flat.nodes]].values
        out['node_names'] = np.array([n.name for n in flat.nodes], dtype=str)
    save('nucleotide_hky_f81', **out)


def case_hiv1c():
    """
    BASELINE config 5: examples/HIV1C (3 619 tips).  The tree and a 4-column subset of metadata.tab are copied as
    fixtures; goldens: Loc (k=12) at the parameters stored in examples/HIV1C/data/pastml_params (pinned -3692.227),
    Loc optimised by the reference (about 3 minutes), two binary drug-resistance columns optimised.
    """
    src = os.path.join(REF, 'examples', 'HIV1C', 'data')
    dst = os.path.join(DATA, 'hiv1c')
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, 'best', 'pastml_phyml_tree.nwk'), os.path.join(dst, 'pastml_phyml_tree.nwk'))
    columns = ['Loc', 'RT:K103N', 'RT:M184V', 'PR:L90M']
    meta = pd.read_csv(os.path.join(src, 'metadata.tab'), sep='\t', index_col=0, header=0)[columns]
    meta.to_csv(os.path.join(dst, 'metadata_subset.tab'), sep='\t')
    stored = pd.read_csv(os.path.join(src, 'pastml_params', 'params.tree_pastml_phyml_tree.loc_Loc.tab'), sep='\t',
                         index_col=0, header=0)['value']
    stride = 37

    def load():
        tree = our_tree.read_tree(os.path.join(dst, 'pastml_phyml_tree.nwk'))
        df = pd.read_csv(os.path.join(dst, 'metadata_subset.tab'), sep='\t', index_col=0, header=0)
        df.index = df.index.map(str)
        return tree, df

    out = {}
    # ---- Loc at the stored parameters
    tree, df = load()
    states = np.array(sorted([_ for _ in df['Loc'].unique() if not pd.isna(_) and '' != _]))
    params = {'scaling_factor': float(stored['scaling_factor'])}
    params.update({s: float(stored[s]) for s in states})
    t0 = __import__('time').time()
    res = racr(tree, df[['Loc']].copy(), prediction_method='MPPA', model='F81', column2parameters={'Loc': params},
               threads=1)[0]
    print('Loc fixed: {:.1f} s'.format(__import__('time').time() - t0))
    flat = our_tree.FlatForest.from_trees([tree])
    names = [n.name for n in flat.nodes]
    sample = np.arange(0, flat.n_nodes, stride)
    out['sample'] = sample
    out['n_nodes'] = flat.n_nodes
    out['stored_loglik'] = float(stored['log_likelihood'])
    out['loc_states'] = np.array(states, dtype=str)
    out['locfix_sf'] = float(res['model'].sf)
    out['locfix_frequencies'] = np.array(res['model'].frequencies)
    out['locfix_loglik'] = res['log_likelihood']
    for m in ('JOINT', 'MAP', 'MPPA'):
        out['locfix_loglik_restricted_' + m] = res['log_likelihood_restricted_' + m]
    out['locfix_num_unresolved_nodes'] = res['num_unresolved_nodes']
    out['locfix_num_states_per_node_avg'] = res['num_states_per_node_avg']
    mp = res['marginal_probabilities']
    out['locfix_posterior_sample'] = mp.loc[[names[i] for i in sample]].values
    s2i = {s: i for i, s in enumerate(res['states'])}
    sel = np.zeros((flat.n_nodes, len(s2i)), dtype=np.int8)
    for i, n in enumerate(flat.nodes):
        for s in getattr(n, 'Loc'):
            sel[i, s2i[s]] = 1
    out['locfix_selected_mppa'] = sel
    out['locfix_joint_state'] = collect(flat.nodes, 'Loc_JOINT_STATE', dtype=np.int64)
    out['forest_stats'] = np.array([res['model'].forest_stats.avg_nonzero_brlen, res['model'].forest_stats.num_nodes,
                                    res['model'].forest_stats.num_tips, res['model'].forest_stats.forest_length])

    # ---- optimised: two binary columns (seconds) and Loc (minutes)
    for col, label in (('RT:K103N', 'k103n'), ('PR:L90M', 'l90m'), ('Loc', 'locopt')):
        tree, df = load()
        t0 = __import__('time').time()
        res = racr(tree, df[[col]].copy(), prediction_method='MPPA', model='F81', threads=1)[0]
        print('{} optimised: {:.1f} s'.format(col, __import__('time').time() - t0))
        out[label + '_states'] = np.array(res['states'], dtype=str)
        out[label + '_sf'] = float(res['model'].sf)
        out[label + '_frequencies'] = np.array(res['model'].frequencies)
        out[label + '_loglik'] = res['log_likelihood']
        out[label + '_loglik_restricted_MPPA'] = res['log_likelihood_restricted_MPPA']
        out[label + '_num_unresolved_nodes'] = res['num_unresolved_nodes']
        out[label + '_posterior_sample'] = res['marginal_probabilities'].loc[[names[i] for i in sample]].values
        out[label + '_reference_seconds'] = __import__('time').time() - t0
    save('hiv1c', **out)


CASES = dict(hiv1c=case_hiv1c, data=copy_data, pij=case_pij, albania=case_albania, synthetic_small=case_synthetic_small,
             synthetic_large=case_synthetic_large, edge=case_edge, nucleotide=case_hky_nucleotide)



def case_marginal_counts():
    """pastml.ml.marginal_counts (ml.py:753-862) with many repetitions: expected transition counts, for statistical parity."""
    out = {}
    np.random.seed(12345)
    # balanced 64-tip tree, JC k=4
    flat = synthetic.balanced_forest(6)
    roots = flat.to_tree_nodes()
    fs = RForestStats(roots)
    states, tips_states = annotate_synthetic(flat, roots, 'c', 4, 3)
    model = RJC(states=states, forest_stats=fs, sf=2.0)
    model.freeze()
    out['jc_counts'] = rml.marginal_counts(roots, 'c', model, n_repetitions=40000)
    out['jc_sf'] = 2.0
    out['jc_tip_states'] = tips_states
    # Albania, F81 at the optimum (zero branches -> altered nodes)
    z = np.load(os.path.join(HERE, 'albania_F81.npz'))
    tree, df = albania_inputs()
    from pastml.annotation import preannotate_forest
    preannotate_forest([tree], df=df)
    model = RF81(states=z['opt_states'], forest_stats=RForestStats([tree]), sf=float(z['opt_sf']),
                 frequencies=z['opt_frequencies'])
    model.freeze()
    out['albania_counts'] = rml.marginal_counts([tree], 'Country', model, n_repetitions=40000)
    out['n_repetitions'] = 40000
    save('marginal_counts', **out)


CASES['marginal_counts'] = case_marginal_counts


def case_eigen_optimised():
    """acr() with the eigen-decomposed models on the Albanian tree with seeded random tip states:
    JTT (sf free) and CUSTOM_RATES with a random symmetric 5-state rate matrix (sf + frequencies free)."""
    import time
    out = {}
    rng = np.random.default_rng(2024)
    # JTT
    tree, _ = albania_inputs()
    tips = list(tree.iter_leaves())
    jtt_states = rng.integers(0, 20, size=len(tips))
    for tip, s in zip(tips, jtt_states):
        tip.add_feature('aa', {JTT_STATES[s]})
    t0 = time.time()
    res = racr(tree, columns=['aa'], column2states={'aa': JTT_STATES}, prediction_method='MPPA', model='JTT',
               threads=1)[0]
    out['jtt_seconds'] = time.time() - t0
    flat = our_tree.FlatForest.from_trees([tree])
    names = [n.name for n in flat.nodes]
    out['jtt_tip_states'] = jtt_states
    out['jtt_tip_names'] = np.array([t.name for t in tips], dtype=str)
    out['jtt_loglik'] = res['log_likelihood']
    out['jtt_sf'] = float(res['model'].sf)
    out['jtt_loglik_restricted_MPPA'] = res['log_likelihood_restricted_MPPA']
    out['jtt_posterior'] = res['marginal_probabilities'].loc[names].values
    out['jtt_num_unresolved_nodes'] = res['num_unresolved_nodes']
    # CUSTOM_RATES, k = 5
    k = 5
    states = synthetic.state_names(k)
    R = np.triu(rng.uniform(0.2, 3, size=(k, k)), 1)
    R = R + R.T
    rate_file = os.path.join(DATA, 'custom_rates_k5.txt')
    from pastml.models.generator import save_matrix
    save_matrix(states, R, rate_file)
    tree, _ = albania_inputs()
    tips = list(tree.iter_leaves())
    cr_states = rng.choice(k, size=len(tips), p=[0.4, 0.25, 0.2, 0.1, 0.05])
    for tip, s in zip(tips, cr_states):
        tip.add_feature('cr', {states[s]})
    t0 = time.time()
    res = racr(tree, columns=['cr'], column2states={'cr': states}, prediction_method='MPPA', model='CUSTOM_RATES',
               column2rates={'cr': rate_file}, threads=1)[0]
    out['cr_seconds'] = time.time() - t0
    out['cr_tip_states'] = cr_states
    out['cr_rate_matrix'] = R
    out['cr_states'] = np.array(states, dtype=str)
    out['cr_loglik'] = res['log_likelihood']
    out['cr_sf'] = float(res['model'].sf)
    out['cr_frequencies'] = np.array(res['model'].frequencies)
    out['cr_loglik_restricted_MPPA'] = res['log_likelihood_restricted_MPPA']
    out['cr_posterior'] = res['marginal_probabilities'].loc[names].values
    save('eigen_optimised', **out)


CASES['eigen_optimised'] = case_eigen_optimised


def case_cfg3_full():
    """
    BASELINE config 3 at full size: balanced 262 144-tip tree, JTT (k = 20), one character, joint
    (Pupko) sweep of the reference (ml.py:82-148 with is_marginal=False) + choose_ancestral_states_joint (:598-622).
    Stored: lnL_joint, the marginal lnL, and at a strided node sample the joint states, arg-max rows, log10 bottom-up
    values (bu, sf).  The full joint-state vector is stored as well (one byte per node, compresses well).
    """
    import time
    n_levels = 18
    flat = synthetic.balanced_forest(n_levels)
    roots = flat.to_tree_nodes()
    fs = RForestStats(roots)
    states = JTT_STATES
    tips_states = synthetic.tip_states(flat.n_tips, 20, 0)
    for j, t in enumerate(flat.tips):
        flat.nodes[t].add_feature('c0', {states[tips_states[j]]})
    model = RJTT(forest_stats=fs, sf=1.)
    model.freeze()
    sys.setrecursionlimit(10000)
    t0 = time.time()
    for tree in roots:
        rml.initialize_allowed_states(tree, 'c0', states)
    out = dict(forest_stats_arrays(fs))
    out.update(model_arrays(model))
    out['loglik'] = sum(rml.get_bottom_up_loglikelihood(tree=t, character='c0', model=model, is_marginal=True,
                                                        alter=True) for t in roots)
    t1 = time.time()
    out['loglik_joint'] = sum(rml.get_bottom_up_loglikelihood(tree=t, character='c0', model=model,
                                                              is_marginal=False, alter=True) for t in roots)
    out['reference_seconds_joint_sweep'] = time.time() - t1
    nodes = flat.nodes
    for t in roots:
        rml.choose_ancestral_states_joint(t, 'c0', states, model.frequencies)
    out['reference_seconds'] = time.time() - t0
    out['joint_state'] = collect(nodes, feat('c0', rml.JOINT_STATE), dtype=np.int8)
    sample = np.arange(0, flat.n_nodes, 4099)
    out['sample'] = sample
    JS = feat('c0', rml.BU_LH_JOINT_STATES)
    jt = np.full((len(sample), 20), -1, dtype=np.int64)
    for r, i in enumerate(sample):
        if hasattr(nodes[i], JS):
            jt[r] = getattr(nodes[i], JS)
    out['joint_table'] = jt
    out['bu_joint'] = np.array([getattr(nodes[i], feat('c0', rml.BU_LH)) for i in sample])
    out['bu_joint_sf'] = np.array([getattr(nodes[i], feat('c0', rml.BU_LH_SF)) for i in sample])
    out['n_levels'] = n_levels
    save('synthetic_cfg3_full', **out)


CASES['cfg3_full'] = case_cfg3_full


def case_eigen_k61():
    """
    A codon-sized eigen model: the reference's CustomRatesModel (pastml/models/CustomRatesModel.py:35-79,
    generator.py:54-65) with k = 61 states, seeded symmetric rates and frequencies, on a balanced 4 096-tip tree: the
    marginal pass and the joint (Pupko) sweep with its back-trace, fixed parameters.  Stored: every scalar, the joint states of
    all nodes, and the vectors (posteriors, BU / TD / LH with their scales, arg-max rows) at every 7th node.
    """
    k, n_levels = 61, 12
    rng = np.random.default_rng(61)
    flat = synthetic.balanced_forest(n_levels)
    roots = flat.to_tree_nodes()
    fs = RForestStats(roots)
    states, tips_states = annotate_synthetic(flat, roots, 'c0', k, 0)
    rates = np.triu(rng.uniform(0.05, 3.0, size=(k, k)), 1)
    rates = rates + rates.T
    freqs = rng.dirichlet(np.ones(k) * 4)
    model = RCR(forest_stats=fs, sf=1., states=np.array(states), rate_matrix=rates, frequencies=freqs)
    model.freeze()
    sys.setrecursionlimit(10000)
    flat2, cap = sweep_capture(roots, 'c0', model)
    out = dict(forest_stats_arrays(fs))
    out.update(model_arrays(model))
    sample = np.arange(0, flat.n_nodes, 7)
    out['sample'] = sample
    for key, v in cap.items():
        if isinstance(v, np.ndarray) and v.ndim == 2 and len(v) == flat.n_nodes:
            out[key] = v[sample]      # vectors / mask rows: the sample
        else:
            out[key] = v              # scalars and per-node scalars (joint states, scales) in full
    out['tip_states'] = tips_states
    out['n_levels'] = n_levels
    save('synthetic_cr_k61_L12', **out)


CASES['eigen_k61'] = case_eigen_k61


def case_eigen_k100():
    """
    An eigen model beyond 64 states (round 6: sum sweeps with one matrix in LDS, P(t) batch on the matrix cores): the reference's
    CustomRatesModel (pastml/models/CustomRatesModel.py:35-79, generator.py:54-65) with k = 100 states, seeded symmetric rates
    and frequencies, on a balanced 2 048-tip tree, a twentieth of the tips unannotated: marginal pass and joint sweep with its
    back-trace, fixed parameters.  Stored: every scalar, the joint states of all nodes, the vectors at every 13th node.
    """
    k, n_levels = 100, 11
    rng = np.random.default_rng(100)
    flat = synthetic.balanced_forest(n_levels)
    roots = flat.to_tree_nodes()
    fs = RForestStats(roots)
    states, tips_states = annotate_synthetic(flat, roots, 'c0', k, 0, missing_frac=0.05, seed=100)
    observed = np.array([bool(getattr(flat.nodes[t], 'c0', None)) for t in flat.tips])
    rates = np.triu(rng.uniform(0.05, 3.0, size=(k, k)), 1)
    rates = rates + rates.T
    freqs = rng.dirichlet(np.ones(k) * 4)
    model = RCR(forest_stats=fs, sf=0.8, states=np.array(states), rate_matrix=rates, frequencies=freqs)
    model.freeze()
    sys.setrecursionlimit(10000)
    flat2, cap = sweep_capture(roots, 'c0', model)
    out = dict(forest_stats_arrays(fs))
    out.update(model_arrays(model))
    sample = np.arange(0, flat.n_nodes, 13)
    out['sample'] = sample
    for key, v in cap.items():
        if isinstance(v, np.ndarray) and v.ndim == 2 and len(v) == flat.n_nodes:
            out[key] = v[sample]
        else:
            out[key] = v
    out['tip_states'] = tips_states
    out['tip_observed'] = observed
    out['n_levels'] = n_levels
    save('synthetic_cr_k100_L11', **out)


CASES['eigen_k100'] = case_eigen_k100


def case_f81_k300():
    """
    More than 256 states (the reference has no bound on k, pastml/ml.py:134): the reference's F81Model with 300 states on a
    balanced 1 024-tip tree, a tenth of the tips unannotated -- marginal pass and joint sweep with its back-trace, fixed
    parameters.  Stored: every scalar, the joint states of all nodes, and the vectors at every 7th node.
    """
    k, n_levels = 300, 10
    flat = synthetic.balanced_forest(n_levels)
    roots = flat.to_tree_nodes()
    fs = RForestStats(roots)
    states, tips_states = annotate_synthetic(flat, roots, 'c0', k, 0, missing_frac=0.1, seed=300)
    observed = np.array([bool(getattr(flat.nodes[t], 'c0', None)) for t in flat.tips])
    model = RF81(states=states, forest_stats=fs, sf=1.3, frequencies=synthetic.f81_frequencies(k, 0))
    model.freeze()
    flat2, cap = sweep_capture(roots, 'c0', model)
    out = dict(forest_stats_arrays(fs))
    out['tip_observed'] = observed
    out.update(model_arrays(model))
    sample = np.arange(0, flat.n_nodes, 7)
    out['sample'] = sample
    for key, v in cap.items():
        if isinstance(v, np.ndarray) and v.ndim == 2 and len(v) == flat.n_nodes:
            out[key] = v[sample]
        else:
            out[key] = v
    out['tip_states'] = tips_states
    out['n_levels'] = n_levels

    # acr() end to end on the same tree: JC (one free parameter) and EFT, MPPA.  The tip states are simulated down the tree
    # (400 candidate states, three expected changes per unit of branch length), so that the likelihood has an interior
    # optimum -- states dealt out without phylogenetic signal drive the scaling factor to its bound, where the posteriors of
    # the deep nodes are equal up to rounding and every choice among them is noise, in the reference as anywhere.
    import time
    sim = np.random.default_rng(3000)
    n_cand, sf_true = 400, 3.0
    sim_state = np.zeros(flat.n_nodes, dtype=np.int64)
    for n in range(flat.n_nodes):   # ids are in level order: parents first
        p = flat.parent[n]
        if p < 0:
            sim_state[n] = sim.integers(n_cand)
        elif sim.random() < np.exp(-sf_true * flat.dist[n]):
            sim_state[n] = sim_state[p]
        else:
            sim_state[n] = sim.integers(n_cand)
    acr_tip_states = sim_state[np.asarray(flat.tips)]
    acr_names = synthetic.state_names(n_cand)
    out['acr_tip_states'] = acr_tip_states
    out['acr_n_candidates'] = n_cand
    for model_name in ('JC', 'EFT'):
        flat3 = synthetic.balanced_forest(n_levels)
        tree = flat3.to_tree_nodes()[0]
        tips = [flat3.nodes[t] for t in flat3.tips]
        df = pd.DataFrame({'c0': [acr_names[acr_tip_states[j]] if observed[j] else None for j in range(len(tips))]},
                          index=[t.name for t in tips])
        t0 = time.time()
        res = racr(tree, df, prediction_method='MPPA', model=model_name, threads=1)[0]
        print(model_name, 'acr: {:.1f} s, lnL {:.6f}, sf {:.6f}'.format(time.time() - t0, res['log_likelihood'], res['model'].sf))
        nodes3 = our_tree.FlatForest.from_trees([tree]).nodes
        pre = 'acr_{}_'.format(model_name)
        out[pre + 'states'] = np.array(res['states'], dtype=str)
        out[pre + 'sf'] = float(res['model'].sf)
        out[pre + 'loglik'] = res['log_likelihood']
        for m in ('JOINT', 'MAP', 'MPPA'):
            out[pre + 'loglik_restricted_' + m] = res['log_likelihood_restricted_' + m]
        out[pre + 'num_unresolved_nodes'] = res['num_unresolved_nodes']
        out[pre + 'num_states_per_node_avg'] = res['num_states_per_node_avg']
        mp = res['marginal_probabilities']
        out[pre + 'posterior'] = mp.loc[[n.name for n in nodes3]].values[sample]
        s2i = {s: i for i, s in enumerate(res['states'])}
        sel = np.zeros((len(nodes3), len(s2i)), dtype=np.int8)
        for i, n in enumerate(nodes3):
            for st in getattr(n, 'c0'):
                sel[i, s2i[st]] = 1
        out[pre + 'selected_mppa'] = sel[sample]
        out[pre + 'n_selected'] = sel.sum(axis=1)
        out[pre + 'joint_state'] = collect(nodes3, 'c0_JOINT_STATE', dtype=np.int64)
    save('synthetic_f81_k300_L10', **out)


CASES['f81_k300'] = case_f81_k300


HIV1C_SRC = os.path.join(REF, 'examples', 'HIV1C', 'data')
HIV1C_DST = os.path.join(DATA, 'hiv1c')
HIV1C_SCRATCH = os.path.join(REPO, 'scratch', 'hiv1c_all')


def _hiv1c_column_job(col):
    """One column of BASELINE config 5 through the reference's acr() (optimised); cached per column in scratch/."""
    import pickle
    import time
    import hashlib
    tag = hashlib.md5(col.encode()).hexdigest()[:12]
    path = os.path.join(HIV1C_SCRATCH, tag + '.pkl')
    if os.path.exists(path):
        return col, path
    tree = our_tree.read_tree(os.path.join(HIV1C_DST, 'pastml_phyml_tree.nwk'))
    df = pd.read_csv(os.path.join(HIV1C_DST, 'metadata_all.tab.gz'), sep='\t', index_col=0, header=0)
    df.index = df.index.map(str)
    np.random.seed(239)
    t0 = time.time()
    res = racr(tree, df[[col]].copy(), prediction_method='MPPA', model='F81', threads=1)[0]
    dt = time.time() - t0
    flat = our_tree.FlatForest.from_trees([tree])
    names = [n.name for n in flat.nodes]
    states = res['states']
    s2i = {s: i for i, s in enumerate(states)}
    sel = np.zeros((flat.n_nodes, len(states)), dtype=np.uint8)
    for i, n in enumerate(flat.nodes):
        for s in getattr(n, col):
            sel[i, s2i[s]] = 1
    sample = np.arange(0, flat.n_nodes, 37)
    rec = dict(column=col, states=np.array(states, dtype=str), sf=float(res['model'].sf),
               frequencies=np.array(res['model'].frequencies), loglik=res['log_likelihood'],
               loglik_restricted_JOINT=res['log_likelihood_restricted_JOINT'],
               loglik_restricted_MAP=res['log_likelihood_restricted_MAP'],
               loglik_restricted_MPPA=res['log_likelihood_restricted_MPPA'],
               num_unresolved_nodes=res['num_unresolved_nodes'],
               num_states_per_node_avg=res['num_states_per_node_avg'],
               posterior_sample=res['marginal_probabilities'].loc[[names[i] for i in sample]].values,
               selected_bits=np.packbits(sel, axis=1),
               joint_state=collect(flat.nodes, col + '_JOINT_STATE', dtype=np.int16),
               reference_seconds=dt)
    os.makedirs(HIV1C_SCRATCH, exist_ok=True)
    with open(path + '.tmp', 'wb') as f:
        pickle.dump(rec, f)
    os.replace(path + '.tmp', path)
    print('{} (k={}): {:.1f} s, lnL {:.6f}'.format(col, len(states), dt, rec['loglik']), flush=True)
    return col, path


def case_hiv1c_all():
    """
    BASELINE config 5 in full: every usable annotation column of examples/HIV1C/data/metadata.tab (all but `Name`,
    which has one state per tip) through the reference's acr(MPPA, F81) with parameter optimisation.  The table (minus
    `Name`) is stored gzipped as a fixture.  Columns run in a pool of worker processes (PASTML_GOLDEN_WORKERS, default
    5), largest k first, each cached under scratch/hiv1c_all/ so that the run can be resumed; the k = 67 columns take
    the reference hours.  PASTML_GOLDEN_MAX_K skips columns with more states (they are then absent from the fixture).
    """
    import pickle
    import multiprocessing as mp
    os.makedirs(HIV1C_DST, exist_ok=True)
    meta = pd.read_csv(os.path.join(HIV1C_SRC, 'metadata.tab'), sep='\t', index_col=0, header=0)
    meta = meta[[c for c in meta.columns if c != 'Name']]
    meta.to_csv(os.path.join(HIV1C_DST, 'metadata_all.tab.gz'), sep='\t', compression={'method': 'gzip', 'mtime': 0})
    ks = {c: len([_ for _ in meta[c].unique() if not pd.isna(_) and '' != _]) for c in meta.columns}
    max_k = int(os.environ.get('PASTML_GOLDEN_MAX_K', '1000'))
    cols = sorted([c for c in meta.columns if ks[c] <= max_k], key=lambda c: -ks[c])
    workers = int(os.environ.get('PASTML_GOLDEN_WORKERS', '5'))
    if os.environ.get('PASTML_GOLDEN_ASSEMBLE_ONLY'):
        # build the fixture from the columns finished so far (the k = 67 columns take the reference hours each)
        import hashlib
        done = {}
        for c in cols:
            path = os.path.join(HIV1C_SCRATCH, hashlib.md5(c.encode()).hexdigest()[:12] + '.pkl')
            if os.path.exists(path):
                done[c] = path
    else:
        with mp.get_context('fork').Pool(workers) as pool:
            done = dict(pool.imap_unordered(_hiv1c_column_job, cols, chunksize=1))
    out = dict(columns=np.array(list(meta.columns), dtype=str), n_states=np.array([ks[c] for c in meta.columns]),
               sample=np.arange(0, 7237, 37))
    for ci, c in enumerate(meta.columns):
        if c not in done:
            continue
        with open(done[c], 'rb') as f:
            rec = pickle.load(f)
        for key, v in rec.items():
            if key != 'column':
                out['c{}_{}'.format(ci, key)] = v
    out['done'] = np.array([c in done for c in meta.columns])
    save('hiv1c_all', **out)


CASES['hiv1c_all'] = case_hiv1c_all


def case_parsimony():
    """
    pastml/parsimony.py on the Albanian tree (Country) and on random trees with polytomies, missing and multi-state
    tips: the three reconstructions of the meta-method MP (selected states per node, steps, statistics), each method on
    its own (DELTRAN alone reports other steps than inside MP), and the ALL meta-method of ml_acr on Albania / F81
    (order of the results, likelihoods restricted to the parsimonious reconstructions).
    """
    from pastml.annotation import preannotate_forest
    from pastml.parsimony import parsimonious_acr as rpars, STEPS
    out = {}

    def capture(prefix, roots, character, states, num_nodes, num_tips):
        flat = our_tree.FlatForest.from_trees(roots)
        s2i = {s: i for i, s in enumerate(states)}
        ann = np.zeros((flat.n_nodes, len(states)), dtype=np.int8)
        for i, n in enumerate(flat.nodes):
            for s in getattr(n, character, set()):
                ann[i, s2i[s]] = 1
        out[prefix + 'annotation'] = ann
        out[prefix + 'states'] = np.array(states, dtype=str)
        out.update({prefix + k_: v for k_, v in tree_arrays(flat).items()})
        for method in ('MP', 'DOWNPASS', 'ACCTRAN', 'DELTRAN'):
            # every run starts from the annotation (a run overwrites the character feature when the method is not meta)
            for i, n in enumerate(flat.nodes):
                if ann[i].any():
                    n.add_feature(character, set(np.array(states)[ann[i].astype(bool)]))
                else:
                    n.del_feature(character)
            for res in rpars(roots, character, method, np.array(states), num_nodes, num_tips):
                tag = '{}{}_{}_'.format(prefix, method, res['method'])
                sel = np.zeros((flat.n_nodes, len(states)), dtype=np.int8)
                for i, n in enumerate(flat.nodes):
                    for s in getattr(n, res['character']):
                        sel[i, s2i[s]] = 1
                out[tag + 'selected'] = sel
                out[tag + 'character'] = res['character']
                out[tag + 'steps'] = res[STEPS]
                out[tag + 'num_scenarios'] = float(res['num_scenarios'])
                out[tag + 'num_unresolved_nodes'] = res['num_unresolved_nodes']
                out[tag + 'num_states_per_node_avg'] = res['num_states_per_node_avg']

    tree, df = albania_inputs()
    preannotate_forest([tree], df=df)
    fs = RForestStats([tree])
    states = sorted(df['Country'].unique())
    capture('alb_', [tree], 'Country', states, fs.num_nodes, fs.num_tips)
    rng = np.random.default_rng(17)
    for tag, kwargs, k in (('poly_', dict(n_tips=70, seed=13, max_arity=5, zero_frac=0.1), 4),
                           ('forest_', dict(n_tips=60, seed=14, max_arity=3, zero_frac=0.0, n_trees=2), 3)):
        flat = our_tree.FlatForest.random(**kwargs)
        roots = [flat.nodes[r] for r in flat.roots]
        states = list(synthetic.state_names(k))
        for n in flat.nodes:
            u = rng.random()
            if n.is_leaf():
                if u < 0.1:
                    continue
                n.add_feature('ch', set(rng.choice(states, size=2, replace=False)) if u < 0.2
                              else {states[int(rng.integers(k))]})
            elif u < 0.05:
                n.add_feature('ch', {states[int(rng.integers(k))]})
        fs = RForestStats(roots)
        capture(tag, roots, 'ch', states, fs.num_nodes, fs.num_tips)

    # ALL on Albania, F81
    tree, df = albania_inputs()
    res = racr(tree, df, prediction_method='ALL', model='F81', threads=1)
    out['all_methods'] = np.array([r['method'] for r in res], dtype=str)
    out['all_characters'] = np.array([r['character'] for r in res], dtype=str)
    last = res[-1]
    for key in ('log_likelihood', 'log_likelihood_restricted_JOINT', 'log_likelihood_restricted_MAP',
                'log_likelihood_restricted_MPPA', 'log_likelihood_restricted_ACCTRAN',
                'log_likelihood_restricted_DOWNPASS', 'log_likelihood_restricted_DELTRAN'):
        if key in last:
            out['all_' + key] = last[key]
    out['all_mppa_keys'] = np.array(sorted(k_ for k_ in last.keys()), dtype=str)
    out['all_sf'] = float(last['model'].sf)
    save('parsimony', **out)


CASES['parsimony'] = case_parsimony


def _cfg4_marginal_job(job):
    """
    One character of BASELINE config 4 through the reference's marginal path with fixed parameters (sf = 1, tau = 0,
    the character's own frequencies): initialize_allowed_states (ml.py:293-318), get_bottom_up_loglikelihood
    (:82-148), calculate_top_down_likelihood (:240-290), calculate_marginal_likelihoods (:431-465) and the
    normalisation of convert_likelihoods_to_probabilities (:498-500) at a strided node sample.
    """
    import time
    n_levels, c, stride, keep_vectors = job
    k = 64
    flat = synthetic.balanced_forest(n_levels)
    roots = flat.to_tree_nodes()
    fs = RForestStats(roots)
    states, tips_states = annotate_synthetic(flat, roots, 'c', k, c)
    model = RF81(states=states, forest_stats=fs, sf=1., frequencies=synthetic.f81_frequencies(k, c))
    model.freeze()
    t0 = time.time()
    for t in roots:
        rml.initialize_allowed_states(t, 'c', states)
    lnl = sum(rml.get_bottom_up_loglikelihood(tree=t, character='c', model=model, is_marginal=True, alter=False)
              for t in roots)
    for t in roots:
        rml.calculate_top_down_likelihood(t, 'c', model=model)
        rml.calculate_marginal_likelihoods(t, 'c', model.frequencies, clean_up=False)
    seconds = time.time() - t0
    nodes = flat.nodes
    sample = np.arange(0, flat.n_nodes, stride)
    lh = np.array([getattr(nodes[i], feat('c', rml.LH)) for i in sample])
    out = dict(loglik=lnl, sample=sample, reference_seconds=seconds,
               posterior=np.array([row / row.sum() for row in lh]),
               lh_sf=np.array([getattr(nodes[i], feat('c', rml.LH_SF)) for i in sample], dtype=np.float64))
    if keep_vectors:
        out['lh'] = lh
        for key, name, sf_name in (('bu', rml.BU_LH, rml.BU_LH_SF), ('td', rml.TD_LH, rml.TD_LH_SF)):
            out[key] = np.array([getattr(nodes[i], feat('c', name)) for i in sample])
            out[key + '_sf'] = np.array([getattr(nodes[i], feat('c', sf_name)) for i in sample], dtype=np.float64)
    print('cfg4 character {} on 2^{} tips: lnL {:.6f}, {:.0f} s of reference time'.format(c, n_levels, lnl, seconds),
          flush=True)
    return c, out


def case_cfg4_full():
    """
    BASELINE config 4 at full size: (i) the 1 048 576-tip tree, k = 64, F81 with per-character
    frequencies, characters 0 and 1 (= columns 0 and 1 of rank 0's shard in bench.py): lnL and, at every 4 099th node,
    posteriors, marginal likelihoods, bottom-up and top-down vectors with their scaling factors; (ii) all 32
    characters of the bench shard on a 4 096-tip tree: lnL and posteriors at every 37th node.
    ~4 minutes of reference time per sweep-triple of the full tree; the three jobs run as separate processes.
    """
    import multiprocessing as mp
    sys.setrecursionlimit(10000)
    jobs = [(20, 0, 4099, True), (20, 1, 4099, True)]
    small = [(12, c, 37, False) for c in range(32)]
    with mp.get_context('fork').Pool(3) as pool:
        big = pool.map_async(_cfg4_marginal_job, jobs, chunksize=1)
        small_res = dict(pool.map(_cfg4_marginal_job, small, chunksize=32))
        big_res = dict(big.get())
    out = dict(n_levels=20, k=64, characters=np.array([0, 1]))
    for c, res in big_res.items():
        for key, v in res.items():
            out['c{}_{}'.format(c, key)] = v
        out['c{}_frequencies'.format(c)] = synthetic.f81_frequencies(64, c)
    out['small_n_levels'] = 12
    out['small_sample'] = small_res[0]['sample']
    out['small_loglik'] = np.array([small_res[c]['loglik'] for c in range(32)])
    out['small_posterior'] = np.stack([small_res[c]['posterior'] for c in range(32)])
    out['small_lh_sf'] = np.stack([small_res[c]['lh_sf'] for c in range(32)])
    save('synthetic_cfg4_full', **out)


CASES['cfg4_full'] = case_cfg4_full


def case_hiv1c_year_trace(perturb=0.0):
    """
    The reference's optimiser path for the one HIV1C column whose optimum ours misses by more than 1e-6 relative ('Year',
    k = 30): pastml.acr.acr() as in hiv1c_all, with the `minimize` that pastml/ml.py:231 calls wrapped (module attribute,
    the reference's files untouched) so that every L-BFGS-B run leaves its start, its iterates (x_k, f(x_k)), its end and
    scipy's message.  ~25 minutes of reference time.
    """
    import time
    import scipy.optimize
    col = 'Year'
    tree = our_tree.read_tree(os.path.join(HIV1C_DST, 'pastml_phyml_tree.nwk'))
    df = pd.read_csv(os.path.join(HIV1C_DST, 'metadata_all.tab.gz'), sep='\t', index_col=0, header=0)
    df.index = df.index.map(str)
    runs = []
    real = scipy.optimize.minimize

    def traced(fun, x0, **kw):
        its = []
        if perturb:   # (the sensitivity run: every start point moved by a relative `perturb`, the size of a rounding error)
            x0 = np.array(x0, dtype=np.float64) * (1.0 + perturb)
        res = real(fun, x0=x0, callback=lambda xk: its.append(np.array(xk, dtype=np.float64)), **kw)
        runs.append(dict(x0=np.array(x0, dtype=np.float64), x=np.array(res.x), fun=float(res.fun), success=bool(res.success),
                         nit=int(res.nit), nfev=int(res.nfev), message=str(res.message), iterates=np.array(its)))
        return res

    rml.minimize = traced
    try:
        np.random.seed(239)
        t0 = time.time()
        res = racr(tree, df[[col]].copy(), prediction_method='MPPA', model='F81', threads=1)[0]
        dt = time.time() - t0
    finally:
        rml.minimize = real
    out = dict(column=col, loglik=res['log_likelihood'], sf=float(res['model'].sf),
               frequencies=np.array(res['model'].frequencies), reference_seconds=dt, n_runs=len(runs))
    for i, r in enumerate(runs):
        for key, v in r.items():
            out['run{}_{}'.format(i, key)] = v
    out['start_points_moved_by'] = perturb
    save('hiv1c_year_trace' if not perturb else 'hiv1c_year_trace_perturbed', **out)


CASES['hiv1c_year_trace'] = case_hiv1c_year_trace
# the same run with every L-BFGS-B start point moved by 1e-12 relative: how far the REFERENCE's own optimum moves when its
# input moves by a rounding error (the fixture keeps the run summaries and the final values)
CASES['hiv1c_year_trace_perturbed'] = lambda: case_hiv1c_year_trace(1e-12)

if __name__ == '__main__':
    np.random.seed(239)
    todo = sys.argv[1:] or list(CASES)
    for name in todo:
        print('=== ', name)
        CASES[name]()
