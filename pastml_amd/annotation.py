"""
Forest statistics and tip annotation (the part of pastml/annotation.py that feeds the likelihood path).
"""
import numpy as np
import pandas as pd

from pastml_amd.tree import FlatForest, TreeNode, get_flat_forest


class ForestStats(object):
    """
    avg_nonzero_brlen (default sf is its inverse), num_nodes, num_tips, forest_length (sum of branch lengths) and
    num_trees (reference: pastml/annotation.py:29-101).  Accepts a list of TreeNode roots or a FlatForest.
    """

    def __init__(self, forest):
        if isinstance(forest, FlatForest):
            flat, num_trees = forest, len(forest.roots)
        else:
            if isinstance(forest, TreeNode):
                forest = [forest]
            flat, num_trees = get_flat_forest(forest), len(forest)
        self.avg_nonzero_brlen, self.num_nodes, self.num_tips, self.forest_length = get_forest_stats(flat)
        self.num_trees = num_trees


def get_forest_stats(flat):
    """
    [avg non-zero branch length, #nodes, #tips, total length].  The reference accumulates tip and internal lengths
    separately, node by node in level order (annotation.py:36-101); we reproduce that summation order so that the
    default scaling factor is bit-identical.
    """
    dist = flat.dist
    is_tip = flat.is_tip
    # per-tree level order == ascending id inside one tree; the reference loops tree by tree
    if len(flat.roots) > 1:
        order = np.lexsort((np.arange(flat.n_nodes), flat.tree_id))
        dist, is_tip = dist[order], is_tip[order]
    nonzero = dist != 0
    num_zero = int((~nonzero).sum())

    def seq_sum(x):
        # np.cumsum adds strictly left to right, like the reference's running "+="
        return float(np.cumsum(x)[-1]) if len(x) else 0

    len_ext = seq_sum(dist[nonzero & is_tip])
    len_int = seq_sum(dist[nonzero & ~is_tip])
    n = flat.n_nodes
    total = len_ext + len_int
    avg_len = total / (n - num_zero) if n > num_zero else 0
    return [avg_len, n, int(flat.n_tips), total]


def df2gdf(df):
    """
    One row per node name holding, for every column, the set of its non-empty values over all rows of that name
    (what annotation.py:104-110 builds with a pandas group-by/apply: one Python call per name and column there, one pass
    over the column arrays here).  Rows come out sorted by name, as a group-by leaves them; empty cells of ``df`` become
    '' in place, as in the reference.
    """
    df.fillna('', inplace=True)
    names = df.index.to_numpy()
    order = sorted(set(names))
    position = {name: i for i, name in enumerate(order)}
    rows = np.fromiter((position[name] for name in names), dtype=np.int64, count=len(names))
    gdf = pd.DataFrame(index=pd.Index(order, name=df.index.name), columns=df.columns)
    for c in df.columns:
        cells = [set() for _ in order]
        for r, v in zip(rows, df[c].to_numpy()):
            if v != '' and not pd.isnull(v):
                cells[r].add(v)
        gdf[c] = pd.Series(cells, index=gdf.index, dtype=object)
    return gdf


def preannotate_forest(forest, df=None, gdf=None):
    """Sets ``node.<column> = set(states)`` for annotated nodes, removes the feature elsewhere (annotation.py:113-123)."""
    if gdf is None:
        gdf = df2gdf(df)
    index = set(gdf.index)
    columns = list(gdf.columns)
    records = gdf.to_dict(orient='index')
    for tree in forest:
        for node in tree.traverse('postorder'):
            if node.name in index:
                node.add_features(**records[node.name])
            else:
                for c in columns:
                    node.del_feature(c)
    return gdf.columns, gdf
