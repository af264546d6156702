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


def annotation_columns(flat, df):
    """
    The annotation table as columns over the nodes of ``flat``: {column: AnnotationColumn}.  One pass per table column
    over arrays (codes of the distinct values), no per-node Python objects: a node named in the table gets the set of
    its non-empty values (several rows of one name: their union), other nodes have no such attribute -- what
    pastml/annotation.py:104-123 leaves on the tree.  Empty cells of ``df`` become '' in place, as in the reference.
    """
    from pastml_amd.tree import AnnotationColumn
    df.fillna('', inplace=True)
    N = flat.n_nodes
    node_names = getattr(flat, '_node_names', None)
    if node_names is None:
        node_names = pd.Index([n.name for n in flat.nodes])
        flat._node_names = node_names
    if node_names.is_unique and df.index.is_unique:
        rows_of_node = node_names.get_indexer(df.index)        # df row -> node id (-1: not in the tree)
        row_ids = np.flatnonzero(rows_of_node >= 0)
        node_ids = rows_of_node[row_ids]
        pairs = None
    else:
        # names shared by several nodes and / or several rows per name: explicit (row, node) pairs
        where = {}
        for i, name in enumerate(node_names):
            where.setdefault(name, []).append(i)
        pairs = [(r, i) for r, name in enumerate(df.index) for i in where.get(name, ())]
        row_ids = np.array([r for r, _ in pairs], dtype=np.int64)
        node_ids = np.array([i for _, i in pairs], dtype=np.int64)
    out = {}
    for c in df.columns:
        col = df[c].to_numpy()
        present = ~pd.isnull(col)
        if col.dtype == object:
            present &= np.asarray(col != '', dtype=bool)
        if present.any():
            # hash-based codes, distinct values in sorted order (what np.unique gives, without sorting every cell)
            inverse, values = pd.factorize(col[present], sort=True)
            values = np.asarray(values, dtype=object)
        else:
            values, inverse = np.zeros(0, dtype=object), np.zeros(0, int)
        code_of_row = np.full(len(col), -1, dtype=np.int64)
        code_of_row[present] = inverse
        codes = np.full(N, -2, dtype=np.int64)
        multi = {}
        if pairs is None:
            codes[node_ids] = code_of_row[row_ids]
        else:
            for r, i in zip(row_ids, node_ids):
                cr = code_of_row[r]
                if cr < 0:                      # a row without a value: the node is in the table, nothing more
                    if codes[i] == -2:
                        codes[i] = -1
                elif codes[i] < 0:
                    codes[i] = cr
                elif cr != codes[i]:            # a second, different value for the same node
                    multi.setdefault(int(i), {int(codes[i])}).add(int(cr))
        out[c] = AnnotationColumn(codes, values, multi)
    return out


def preannotate_forest(forest, df=None, gdf=None):
    """
    ``node.<column>`` = set of the node's states for the nodes named in the table, no such attribute elsewhere
    (pastml/annotation.py:113-123).  With ``df`` the annotation becomes columnar features of the flattened forest
    (pastml_amd.tree, no per-node work); a ready-made ``gdf`` (one row of sets per name) is put on the nodes one by one.
    """
    if gdf is None:
        if isinstance(forest, TreeNode):
            forest = [forest]
        flat = get_flat_forest(forest)
        for c, column in annotation_columns(flat, df).items():
            flat.set_column(c, column)
        return df.columns, None
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
