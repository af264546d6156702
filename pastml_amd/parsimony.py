"""
Maximum-parsimony reconstruction (DOWNPASS, ACCTRAN, DELTRAN, MP): what pastml/parsimony.py computes, on the flattened
forest.  Integer set work, no likelihoods: it stays on the host, but as array operations per tree level instead of
per-node Python sets -- state sets are 0/1 rows of an [N, k] array, "the most common states among these sets"
(parsimony.py:73-87) is a segment sum over the contiguous children of each parent followed by a row maximum, and the
passes run level by level over the height / depth ranges the device sweeps use too.

It is here because the ``ALL`` meta-method of ``ml_acr`` (pastml/ml.py:718-733) evaluates the likelihood restricted to
each parsimonious reconstruction, and so that ``acr()`` accepts every prediction method of the reference.
"""
import logging

import numpy as np

from pastml_amd import get_personalized_feature_name, METHOD, STATES, CHARACTER, NUM_SCENARIOS, NUM_UNRESOLVED_NODES, \
    NUM_NODES, NUM_TIPS, NUM_STATES_PER_NODE, PERC_UNRESOLVED

STEPS = 'steps'

DOWNPASS = 'DOWNPASS'
ACCTRAN = 'ACCTRAN'
DELTRAN = 'DELTRAN'
MP = 'MP'

MP_METHODS = {DOWNPASS, ACCTRAN, DELTRAN}
META_MP_METHODS = {MP}


def is_meta_mp(method):
    return method in META_MP_METHODS


def get_default_mp_method():
    return DOWNPASS


def is_parsimonious(method):
    return method in MP_METHODS | {MP}


# ---------------------------------------------------------------------------------------------------------------------
class _Levels(object):
    """Children of the internal nodes of a flat forest as one index list with segment starts, per height level."""

    def __init__(self, flat):
        self.flat = flat
        order = flat.bu_order                       # internal nodes by height
        counts = flat.n_children[order].astype(np.int64)
        starts = np.concatenate(([0], np.cumsum(counts)))
        # children of order[q] are child_ids[starts[q]:starts[q+1]]
        self.child_ids = (np.repeat(flat.first_child[order].astype(np.int64) - starts[:-1], counts)
                          + np.arange(starts[-1]))
        self.starts = starts
        self.order = order


def _levels(flat):
    lv = getattr(flat, '_parsimony_levels', None)
    if lv is None:
        lv = _Levels(flat)
        flat._parsimony_levels = lv
    return lv


def _children_sum(lv, values, a, b):
    """Sum over the children of the internal nodes order[a:b] of the rows of ``values``: [b - a, k]."""
    s0, s1 = lv.starts[a], lv.starts[b]
    rows = values[lv.child_ids[s0:s1]]
    return np.add.reduceat(rows, lv.starts[a:b] - s0, axis=0)


def _most_common(counts):
    """Rows of counts -> 0/1 rows marking the entries that reach the row maximum (parsimony.py:73-87)."""
    return (counts == counts.max(axis=1, keepdims=True)).astype(np.int32)


def _restrict(current, wanted):
    """current & wanted where that is not empty, current elsewhere."""
    both = current & wanted
    keep = both.any(axis=1, keepdims=True)
    return np.where(keep, both, current)


def uppass(flat, initial):
    """
    Bottom-up pass (parsimony.py:90-122): a parent keeps those of its states that are the most common among its
    children's sets, or all of its states if none is.  Returns the bottom-up sets [N, k].
    """
    lv = _levels(flat)
    bu = initial.copy()
    for l in range(flat.n_bu_levels):
        a, b = flat.bu_offsets[l], flat.bu_offsets[l + 1]
        parents = lv.order[a:b]
        bu[parents] = _restrict(bu[parents], _most_common(_children_sum(lv, bu, a, b)))
    return bu


def acctran(flat, bu):
    """Top-down pass that moves changes towards the root (parsimony.py:125-159)."""
    out = bu.copy()
    for d in range(1, flat.n_td_levels):
        a, b = flat.td_offsets[d], flat.td_offsets[d + 1]
        out[a:b] = _restrict(bu[a:b], out[flat.parent[a:b]])
    return out


def downpass(flat, bu, initial):
    """
    Top-down pass combining, for every node, what its supertree and its subtree say (parsimony.py:162-213): the
    node's "up" set is the most common among its parent's up set and its siblings' bottom-up sets; its final set the
    most common among its up set and its children's bottom-up sets, restricted to its initial states if possible.
    """
    lv = _levels(flat)
    N, k = bu.shape
    kids = np.zeros((N, k), dtype=np.int32)      # sum of the children's bottom-up sets
    if len(lv.order):
        kids[lv.order] = _children_sum(lv, bu, 0, len(lv.order))
    up = np.ones((N, k), dtype=np.int32)
    out = initial.copy()
    internal = flat.n_children > 0
    for d in range(flat.n_td_levels):
        a, b = flat.td_offsets[d], flat.td_offsets[d + 1]
        if d > 0:
            p = flat.parent[a:b]
            up[a:b] = _most_common(up[p] + kids[p] - bu[a:b])
        both = np.where(internal[a:b, None], _most_common(up[a:b] + kids[a:b]), up[a:b])
        out[a:b] = _restrict(initial[a:b], both)
    return out


def deltran(flat, pars):
    """Top-down pass that moves changes towards the tips (parsimony.py:216-246); after downpass."""
    out = pars.copy()
    for d in range(1, flat.n_td_levels):
        a, b = flat.td_offsets[d], flat.td_offsets[d + 1]
        both = out[a:b] & out[flat.parent[a:b]]
        keep = both.any(axis=1, keepdims=True)
        out[a:b] = np.where(keep, both, out[a:b])
    return out


def num_parsimonious_steps(flat, sets):
    """
    Minimal number of state changes compatible with the sets (parsimony.py:361-380): cost[n][s] = sum over the children
    of min(cost[c][s], 1 + min cost[c]) for the states s of n; summed over the trees of the forest.
    """
    lv = _levels(flat)
    big = np.iinfo(np.int64).max // 4
    cost = np.where(sets > 0, 0, big).astype(np.int64)
    for l in range(flat.n_bu_levels):
        a, b = flat.bu_offsets[l], flat.bu_offsets[l + 1]
        s0, s1 = lv.starts[a], lv.starts[b]
        rows = cost[lv.child_ids[s0:s1]]
        rows = np.minimum(rows, 1 + rows.min(axis=1, keepdims=True))
        total = np.add.reduceat(rows, lv.starts[a:b] - s0, axis=0)
        parents = lv.order[a:b]
        cost[parents] = np.where(sets[parents] > 0, total, big)
    return int(cost[flat.roots].min(axis=1).sum())


# ---------------------------------------------------------------------------------------------------------------------
def parsimonious_acr(forest, character, prediction_method, states, num_nodes, num_tips):
    """
    Parsimonious states on the trees, stored as the node feature ``character`` (``character_<METHOD>`` for the
    meta-method MP); returns the list of result dictionaries (pastml/parsimony.py:249-333).
    """
    from pastml_amd.batch import annotation_words, masks_from_words, words_from_masks, count_scenarios
    from pastml_amd.tree import TreeNode, get_flat_forest, StateSetColumn
    if isinstance(forest, TreeNode):
        forest = [forest]
    logger = logging.getLogger('pastml')
    flat = get_flat_forest(forest)
    states = np.asarray(states)
    k = len(states)
    words, _ = annotation_words(flat, character, states)
    given = masks_from_words(words, k).astype(np.int32)
    initial = np.where(given.any(axis=1, keepdims=True), given, 1).astype(np.int32)
    bu = uppass(flat, initial)

    results = []
    result = {STATES: states, NUM_NODES: num_nodes, NUM_TIPS: num_tips}

    def report(method, sets):
        name = character if prediction_method == method else get_personalized_feature_name(character, method)
        flat.set_column(name, StateSetColumn(words_from_masks(sets, k), states))
        sizes = sets.sum(axis=1)
        scenarios = count_scenarios(sizes)
        res = result.copy()
        res[NUM_SCENARIOS] = scenarios
        res[NUM_UNRESOLVED_NODES] = int((sizes > 1).sum())
        res[NUM_STATES_PER_NODE] = int(sizes.sum()) / num_nodes
        res[PERC_UNRESOLVED] = res[NUM_UNRESOLVED_NODES] * 100 / num_nodes
        logger.debug('{} node{} unresolved ({:.2f}%) for {} by {}, i.e. {:.4f} state{} per node in average.'
                     .format(res[NUM_UNRESOLVED_NODES], 's are' if res[NUM_UNRESOLVED_NODES] != 1 else ' is',
                             res[PERC_UNRESOLVED], character, method, res[NUM_STATES_PER_NODE],
                             's' if res[NUM_STATES_PER_NODE] > 1 else ''))
        res[CHARACTER], res[METHOD] = name, method
        results.append(res)

    if prediction_method in (ACCTRAN, MP):
        sets = acctran(flat, bu)
        result[STEPS] = num_parsimonious_steps(flat, sets)
        report(ACCTRAN, sets)
    if prediction_method != ACCTRAN:
        sets = downpass(flat, bu, initial)
        result[STEPS] = 0
        if prediction_method in (DOWNPASS, MP):
            result[STEPS] = num_parsimonious_steps(flat, sets)
            report(DOWNPASS, sets)
        result[STEPS] = 0
        if prediction_method in (DELTRAN, MP):
            sets = deltran(flat, sets)
            result[STEPS] = num_parsimonious_steps(flat, sets)
            report(DELTRAN, sets)
    logger.debug("Parsimonious reconstruction for {} requires {} state changes.".format(character, result[STEPS]))
    return results
