"""
Deterministic synthetic workloads of BASELINE.json's configs, generated identically on every
box: balanced binary trees with uniform(0.01, 0.2) branch lengths (seed 42), tip states ``integers(0, k)`` with seed
1000 + character, F81 frequencies ``dirichlet(ones(k))`` with seed 2000 + character.
"""
import numpy as np

from pastml_amd.tree import FlatForest


def balanced_forest(n_levels, seed=42):
    """Perfectly balanced binary tree with 2**n_levels tips as a FlatForest."""
    return FlatForest.balanced(n_levels, seed=seed)


def tip_states(n_tips, k, character):
    """State index of every tip (in tip-id order) for the given character index."""
    return np.random.default_rng(1000 + character).integers(0, k, size=n_tips).astype(np.int32)


def f81_frequencies(k, character):
    return np.random.default_rng(2000 + character).dirichlet(np.ones(k))


def state_names(k):
    """Sorted, fixed-width state names."""
    width = len(str(k - 1))
    return np.array(['s{:0{w}d}'.format(i, w=width) for i in range(k)])


def one_hot_masks(flat, k, states_of_tips):
    """int8 [N, k] masks: tips one-hot, internal nodes all ones."""
    masks = np.ones((flat.n_nodes, k), dtype=np.int8)
    masks[flat.tips] = 0
    masks[flat.tips, states_of_tips] = 1
    return masks
