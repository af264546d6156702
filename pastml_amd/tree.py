"""
Tree containers for the ACR hot path.

Two representations live here:

* :class:`TreeNode` -- a minimal pointer tree that duck-types the subset of ``ete3.Tree`` that the reference's
  ``acr()`` / ``ml.py`` touch (``traverse``, ``children``, ``up``, ``dist``, ``name``,
  ``is_root``, ``is_leaf``, ``add_feature(s)``, ``del_feature``, ``features``, leaf iteration).  ete3 itself is not
  installed on our boxes, so callers build trees with :func:`read_tree` / :func:`read_forest`
  (reference: ``pastml/tree.py:176-222``).  Only newick reading/naming is provided; tree editing, dates, polytomy
  resolution and nexus (``pastml/tree.py`` rest) are out of scope.

* :class:`FlatForest` -- the structure-of-arrays form that is uploaded to the GPU: nodes are renumbered in
  breadth-first (level) order over the whole forest so that (i) the children of a node are contiguous,
  (ii) every depth level is a contiguous id range (top-down sweep), and (iii) the row order of the posterior
  table equals ``tree.traverse()`` order (``pastml/ml.py:498-502``).  Bottom-up levels group the internal nodes by
  height above their deepest tip.
"""
from collections import deque, Counter

import re

import numpy as np

DEFAULT_DIST = 1.0


class NewickError(ValueError):
    pass


# ---------------------------------------------------------------------------------------------------------------------
# Columnar node features.  The reference keeps everything it knows about a node in attributes of the ete3 node object
# (annotations, allowed-state arrays, likelihood vectors, selected states: pastml/ml.py throughout) and walks the tree
# to read or write them, once per character and feature.  Here a feature of ALL nodes of a forest is one column -- an
# array (or a packed bit mask per node) owned by the FlatForest -- and a TreeNode only resolves ``node.<feature>`` to
# its row when somebody asks: writing the results of a character is O(1) per feature, not O(nodes) Python calls.
# A value set on one node with add_feature() lives in that node's __dict__ and hides the column's row for that node.
# ---------------------------------------------------------------------------------------------------------------------
ABSENT = object()
# names that add_feature() ever put into some node's __dict__: FlatForest.set_column() only has to clear shadowing
# entries for these
_DICT_FEATURE_NAMES = set()
_BASE_FEATURES = frozenset(('dist', 'name', 'support'))


class NodeColumn(object):
    """One feature of all nodes.  get(i) returns the value of node i or ABSENT."""

    absent = None  # optional bool array: rows deleted with del_feature()

    def get(self, i):
        raise NotImplementedError

    def mark_absent(self, i, n_nodes):
        if self.absent is None:
            self.absent = np.zeros(n_nodes, dtype=bool)
        self.absent[i] = True


class ArrayColumn(NodeColumn):
    """values[i] (a scalar or a row of a 2-d array); ``convert`` is applied to what is handed out."""

    def __init__(self, values, convert=None):
        self.values = values
        self.convert = convert

    def get(self, i):
        if self.absent is not None and self.absent[i]:
            return ABSENT
        v = self.values[i]
        return self.convert(v) if self.convert is not None else v


class MaskColumn(NodeColumn):
    """Packed allowed-state words [N, W] handed out as the reference's 0/1 int arrays of length k."""

    def __init__(self, words, k):
        self.words = words
        self.k = k

    def get(self, i):
        if self.absent is not None and self.absent[i]:
            return ABSENT
        b = np.ascontiguousarray(self.words[i]).view(np.uint8)
        return np.unpackbits(b, bitorder='little')[:self.k].astype(int)


class StateSetColumn(NodeColumn):
    """Packed state words [N, W] handed out as sets of state names (pastml/ml.py:923-928)."""

    def __init__(self, words, states):
        self.words = words
        self.states = np.asarray(states)

    def get(self, i):
        if self.absent is not None and self.absent[i]:
            return ABSENT
        b = np.ascontiguousarray(self.words[i]).view(np.uint8)
        bits = np.unpackbits(b, bitorder='little')[:len(self.states)].astype(bool)
        return set(self.states[bits])


class AnnotationColumn(NodeColumn):
    """
    A column of the annotation table on the nodes (pastml/annotation.py:113-123): codes[i] = -2 the node is not in the
    table (no such attribute), -1 it is there without a value (empty set), >= 0 index into ``values``; nodes with
    several values (several table rows of one name) are in ``multi`` {node: set of value indices}.
    """

    def __init__(self, codes, values, multi=None):
        self.codes = codes
        self.values = values
        self.multi = multi or {}

    def get(self, i):
        if self.absent is not None and self.absent[i]:
            return ABSENT
        c = self.codes[i]
        if c == -2:
            return ABSENT
        if c == -1:
            return set()
        if i in self.multi:
            return {self.values[j] for j in self.multi[i]}
        return {self.values[c]}


class TreeNode(object):
    """ete3-like tree node. The node *is* the (sub)tree rooted at it."""

    __slots__ = ('children', 'up', 'dist', 'name', 'support', '_features', '__dict__', '_flat_cache', '_cols', '_idx')

    def __init__(self, newick=None, format=None, name=None, dist=None, support=None, quoted_node_names=False):
        self.children = []
        self.up = None
        self.dist = DEFAULT_DIST if dist is None else dist
        self.name = name if name is not None else ''
        self.support = 1.0 if support is None else support
        self._features = _BASE_FEATURES   # shared until the node gets a feature of its own (add_feature)
        self._flat_cache = None
        self._cols = None   # the FlatForest whose columns hold this node's columnar features, and the node's row
        self._idx = -1
        if newick is not None:
            _parse_newick(newick, self)

    # --- features -------------------------------------------------------------------------------------------------
    def __getattr__(self, name):
        # reached only when the attribute is neither a slot nor in __dict__: a columnar feature, or nothing
        if not name.startswith('__'):
            try:
                flat = object.__getattribute__(self, '_cols')
            except AttributeError:
                flat = None
            if flat is not None:
                col = flat.columns.get(name)
                if col is not None:
                    v = col.get(object.__getattribute__(self, '_idx'))
                    if v is not ABSENT:
                        return v
        raise AttributeError(name)

    @property
    def features(self):
        flat = self._cols
        if flat is None or not flat.columns:
            return self._features
        return self._features | {n for n, c in flat.columns.items() if c.get(self._idx) is not ABSENT}

    def add_feature(self, pr_name, pr_value):
        setattr(self, pr_name, pr_value)
        if self._features is _BASE_FEATURES:
            self._features = set(_BASE_FEATURES)
        self._features.add(pr_name)
        _DICT_FEATURE_NAMES.add(pr_name)

    def add_features(self, **features):
        for k, v in features.items():
            self.add_feature(k, v)

    def del_feature(self, pr_name):
        if pr_name in self.__dict__:
            del self.__dict__[pr_name]
        if pr_name in self._features:
            if self._features is _BASE_FEATURES:
                self._features = set(_BASE_FEATURES)
            self._features.discard(pr_name)
        flat = self._cols
        if flat is not None:
            col = flat.columns.get(pr_name)
            if col is not None:
                col.mark_absent(self._idx, flat.n_nodes)

    # --- topology ---------------------------------------------------------------------------------------------------
    def is_root(self):
        return self.up is None

    def is_leaf(self):
        return not self.children

    def add_child(self, child=None, name=None, dist=None, support=None):
        if child is None:
            child = TreeNode()
        if name is not None:
            child.name = name
        if dist is not None:
            child.dist = dist
        if support is not None:
            child.support = support
        self.children.append(child)
        child.up = self
        return child

    def remove_child(self, child):
        self.children.remove(child)
        child.up = None
        return child

    def get_tree_root(self):
        n = self
        while n.up is not None:
            n = n.up
        return n

    def set_outgroup(self, outgroup):
        """
        Re-roots the tree (self must be its root) on the branch above ``outgroup``, at the branch's midpoint, like
        ete3's ``set_outgroup`` that the reference's re-rooting test relies on
        (tests/ACRParameterOptimisationMPPAF81Test.py:20-37): self stays the root object, its first child becomes
        ``outgroup``; what used to hang under the old root moves into the sister subtree (a root with two children
        dissolves: its two branches merge; otherwise an unnamed zero-length connector holds the remaining children).
        """
        if outgroup is self:
            raise ValueError('Cannot set the root as outgroup')
        path = []  # outgroup.up, ..., self
        n = outgroup
        while n.up is not None:
            path.append(n.up)
            n = n.up
        if n is not self:
            raise ValueError('The outgroup is not in this tree')
        self._flat_cache = None
        below_root = path[-2] if len(path) > 1 else outgroup
        rest = [c for c in self.children if c is not below_root]
        if len(rest) == 1:
            connector = rest[0]
        else:
            connector = TreeNode(dist=0.0)
            for c in rest:
                connector.add_child(c)
        if len(path) == 1:
            sister = connector
        else:
            # turn the path outgroup.up -> ... -> below_root upside down: every node on it becomes the parent of its
            # old parent, and the branch lengths move with the branches
            sister = path[0]
            sister.children.remove(outgroup)
            carried = sister.dist
            for child, parent in zip(path[:-2], path[1:-1]):
                parent.children.remove(child)
                carried, parent.dist = parent.dist, carried
                child.add_child(parent)
            path[-2].add_child(connector, dist=connector.dist + carried)
            sister.dist = 0.0
        half = (outgroup.dist + sister.dist) / 2.
        self.children = []
        self.add_child(outgroup, dist=half)
        self.add_child(sister, dist=half)
        return self

    # --- traversal (orders identical to ete3: children are visited left to right) ---------------------------------------
    def traverse(self, strategy='levelorder'):
        if strategy == 'preorder':
            return self._iter_preorder()
        if strategy == 'postorder':
            return self._iter_postorder()
        if strategy == 'levelorder':
            return self._iter_levelorder()
        raise ValueError('Unknown traversal strategy {}'.format(strategy))

    def _iter_preorder(self):
        stack = [self]
        while stack:
            n = stack.pop()
            yield n
            stack.extend(reversed(n.children))

    def _iter_postorder(self):
        stack = [(self, False)]
        while stack:
            n, expanded = stack.pop()
            if expanded or not n.children:
                yield n
            else:
                stack.append((n, True))
                stack.extend((c, False) for c in reversed(n.children))

    def _iter_levelorder(self):
        queue = deque([self])
        while queue:
            n = queue.popleft()
            yield n
            queue.extend(n.children)

    def iter_leaves(self):
        for n in self._iter_preorder():
            if not n.children:
                yield n

    def get_leaves(self):
        return list(self.iter_leaves())

    def iter_descendants(self, strategy='levelorder'):
        for n in self.traverse(strategy):
            if n is not self:
                yield n

    def __iter__(self):
        return self.iter_leaves()

    def __len__(self):
        return sum(1 for _ in self.iter_leaves())

    def __bool__(self):
        return True

    def __repr__(self):
        return 'TreeNode({!r})'.format(self.name)

    def copy(self):
        """Deep copy of topology, names, lengths and features (feature values are shared)."""
        mapping = {}
        for n in self._iter_preorder():
            c = TreeNode(name=n.name, dist=n.dist, support=n.support)
            for f in sorted(n.features):
                if f not in ('dist', 'name', 'support'):
                    c.add_feature(f, getattr(n, f))
            mapping[id(n)] = c
            if n is not self:
                mapping[id(n.up)].add_child(c)
        return mapping[id(self)]

    def write(self, format=3):
        """Newick with all names and branch lengths."""
        out = {}
        for n in self._iter_postorder():
            label = '{}:{}'.format(n.name, repr(float(n.dist)) if n.dist != int(n.dist) else '{:g}'.format(n.dist))
            if n.children:
                out[id(n)] = '({}){}'.format(','.join(out.pop(id(c)) for c in n.children), label)
            else:
                out[id(n)] = label
        return out[id(self)] + ';'


Tree = TreeNode


# one token of a newick string: a structural character, a [comment], a :length, or a label (an optional quoted part
# followed by anything up to the next delimiter; it starts at a non-blank)
_NEWICK_TOKEN = re.compile(r"""\s*(?:(?P<p>[(),;])|\[(?P<c>[^\]]*)\]|:(?P<d>[^,();\[]*)|(?P<l>(?:'[^']*')?[^:,();\[\s]?[^:,();\[]*))""")


def _bare_node():
    # TreeNode() without the argument handling of __init__ (a million-tip tree is two million of these)
    node = TreeNode.__new__(TreeNode)
    node.children = []
    node.up = None
    node.dist = DEFAULT_DIST
    node.name = ''
    node.support = 1.0
    node._features = _BASE_FEATURES
    node._flat_cache = None
    node._cols = None
    node._idx = -1
    return node


def _parse_newick(text, root):
    """
    Iterative newick reader (no recursion, so million-tip trees are fine), one regular-expression token at a time.
    Labels may be quoted with single quotes; ``[...]`` comments are kept on the node they follow; missing branch lengths
    get ete3's default (1.0), a missing root length is 0.
    """
    if '\n' not in text and not text.lstrip().startswith('(') and not text.rstrip().endswith(';'):
        # a path
        with open(text, 'r') as f:
            text = f.read()
    s = text.strip()
    if not s.endswith(';'):
        raise NewickError('Newick string must end with ";"')
    node = root
    root.dist = 0.0
    depth = 0
    expecting_node = True
    pos, n = 0, len(s)
    match = _NEWICK_TOKEN.match
    while pos < n:
        m = match(s, pos)
        if m is None or m.end() == pos:
            if s[pos:].strip() == '':
                break
            raise NewickError('Unexpected "{}" at position {}'.format(s[pos], pos))
        pos = m.end()
        kind = m.lastgroup
        if kind == 'p':
            ch = m.group('p')
            if ch == '(':
                if not expecting_node:
                    raise NewickError('Unexpected "(" at position {}'.format(m.start('p')))
                child = _bare_node()
                node.children.append(child)
                child.up = node
                node = child
                depth += 1
            elif ch == ',':
                if depth == 0:
                    raise NewickError('Unexpected "," at top level')
                child = _bare_node()
                parent = node.up
                parent.children.append(child)
                child.up = parent
                node = child
                expecting_node = True
            elif ch == ')':
                if depth == 0:
                    raise NewickError('Unbalanced ")"')
                node = node.up
                depth -= 1
                expecting_node = False
            else:  # ';'
                break
        elif kind == 'l':
            label = m.group('l')
            if label[:1] == "'":
                close = label.index("'", 1)
                label = label[1:close] + label[close + 1:]
            node.name = label.strip()
            expecting_node = False
        elif kind == 'd':
            try:
                node.dist = float(m.group('d'))
            except ValueError:
                raise NewickError('Bad branch length "{}"'.format(m.group('d')))
            expecting_node = False
        else:  # a comment: kept on the node whose label / length it sits next to
            previous = node.__dict__.get('comment')
            node.__dict__['comment'] = m.group('c') if previous is None else previous + ' ' + m.group('c')
    if depth != 0:
        raise NewickError('Unbalanced parentheses')
    return root


def read_tree(tree_path, columns=None):
    """Reads one newick tree from a path or a string (reference: pastml/tree.py:202-222)."""
    try:
        tree = TreeNode(tree_path)
    except (NewickError, OSError) as e:
        raise ValueError('Could not read the tree {}. Is it a valid newick? ({})'.format(tree_path, e))
    if columns:
        for n in tree.traverse():
            for c in columns:
                vs = set(getattr(n, c).split('|')) if hasattr(n, c) else set()
                if vs:
                    n.add_feature(c, vs)
    return tree


def read_forest(tree_path, columns=None):
    """Reads all the newick trees of a file; negative branches are set to zero (reference: pastml/tree.py:176-199)."""
    with open(tree_path, 'r') as f:
        nwks = f.read().replace('\n', '').split(';')
    if not nwks or not nwks[:-1]:
        raise ValueError('Could not find any trees (in newick format) in the file {}.'.format(tree_path))
    roots = [read_tree(nwk + ';', columns) for nwk in nwks[:-1]]
    for root in roots:
        for _ in root.traverse():
            if _.dist < 0:
                _.dist = 0
    return roots


def name_tree(tree, suffix=""):
    """
    Gives unique names to unnamed / non-uniquely named nodes (reference: pastml/tree.py:76-105).
    """
    existing_names = Counter()
    n_nodes = 0
    for _ in tree.traverse():
        n_nodes += 1
        if _.name:
            existing_names[_.name] += 1
    if n_nodes == len(existing_names):
        return
    i = 0
    new_existing_names = Counter()
    for node in tree.traverse('preorder'):
        name_prefix = node.name if node.name and existing_names[node.name] < 10 \
            else '{}{}{}'.format('t' if node.is_leaf() else 'n', i, suffix)
        name = 'root{}'.format(suffix) if node.is_root() else name_prefix
        while name is None or name in new_existing_names:
            name = '{}{}{}'.format(name_prefix, i, suffix)
            i += 1
        node.name = name
        new_existing_names[name] += 1


# =====================================================================================================================
# Flat (structure-of-arrays) forest
# =====================================================================================================================

class FlatForest(object):
    """
    Level-ordered structure-of-arrays forest.

    Node ids are breadth-first over the whole forest: all roots, then all depth-1 nodes (in parent order, children
    left to right), etc.  Hence ``first_child[p] .. first_child[p] + n_children[p] - 1`` are p's children and depth
    levels are contiguous ranges ``td_offsets[d] .. td_offsets[d+1]``.

    Attributes (numpy arrays, N = number of nodes):
        parent       int32[N]   parent id, -1 for roots
        first_child  int32[N]   id of the first child (undefined for tips)
        n_children   int32[N]
        dist         float64[N] branch length above the node
        depth        int32[N]   0 for roots
        height       int32[N]   0 for tips, 1 + max child height otherwise
        tree_id      int32[N]   index of the tree in the forest
        roots        int32[R]
        tips         int32[T]   ids of the tips, ascending
        bu_order     int32[N-T] internal nodes sorted by (height, id); bu_offsets int32[H+1] delimits height 1..H
        td_parents   int32[N-T] internal nodes sorted by id (== by depth);  td_parent_offsets int32[D+1] per depth
        td_offsets   int32[D+2] id range of each depth level
        post_rank    int32[N]   position of the node in the reference's processing order
                                (trees one after another, each in post-order; pastml/ml.py:109,206)
    ``nodes`` (optional) is the list of TreeNode objects in id order.
    """

    def __init__(self, parent, n_children, first_child, dist, roots, nodes=None):
        self.parent = np.ascontiguousarray(parent, dtype=np.int32)
        self.n_children = np.ascontiguousarray(n_children, dtype=np.int32)
        self.first_child = np.ascontiguousarray(first_child, dtype=np.int32)
        self.dist = np.ascontiguousarray(dist, dtype=np.float64)
        self.roots = np.ascontiguousarray(roots, dtype=np.int32)
        self.nodes = nodes
        self.n_nodes = len(self.parent)
        self.columns = {}   # columnar node features: name -> NodeColumn (see the top of this module)
        self._derive()
        if nodes is not None:
            self.adopt_nodes()

    # ------------------------------------------------------------------------------------------------------------------
    def adopt_nodes(self):
        """Points the TreeNode objects at this forest: ``node.<feature>`` then resolves through ``self.columns``."""
        for i, n in enumerate(self.nodes):
            old = n._cols
            if old is not None and old is not self and old.columns:
                # the tree was edited and flattened again: what the old forest's columns held for this node moves onto
                # the node itself
                for name, col in old.columns.items():
                    if name not in n.__dict__:
                        v = col.get(n._idx)
                        if v is not ABSENT:
                            n.add_feature(name, v)
            n._cols = self
            n._idx = i

    def set_column(self, name, column):
        """
        Sets a columnar feature for all nodes.  Values that add_feature() put on individual nodes under the same name
        would hide the column: they are removed (only names ever used with add_feature need the walk).
        """
        if self.nodes is not None and name in _DICT_FEATURE_NAMES:
            for n in self.nodes:
                n.__dict__.pop(name, None)
        self.columns[name] = column

    def del_column(self, name):
        self.columns.pop(name, None)

    # ------------------------------------------------------------------------------------------------------------------
    def _derive(self):
        N = self.n_nodes
        parent = self.parent
        depth = np.zeros(N, dtype=np.int32)
        tree_id = np.zeros(N, dtype=np.int32)
        # ids are BFS ordered, so depth is non-decreasing in id and a parent's depth is known before its children's
        # level boundaries: roots are the first R ids
        R = len(self.roots)
        if not np.array_equal(self.roots, np.arange(R, dtype=np.int32)):
            raise ValueError('FlatForest expects roots to be ids 0..R-1 (breadth-first numbering)')
        tree_id[:R] = np.arange(R)
        offsets = [0, R]
        lo, hi = 0, R
        d = 0
        while hi < N:
            # children of level [lo, hi) are the next contiguous block
            cnt = int(self.n_children[lo:hi].sum())
            if cnt == 0:
                raise ValueError('Disconnected nodes in FlatForest')
            d += 1
            depth[hi:hi + cnt] = d
            tree_id[hi:hi + cnt] = tree_id[parent[hi:hi + cnt]]
            lo, hi = hi, hi + cnt
            offsets.append(hi)
        self.depth = depth
        self.tree_id = tree_id
        self.td_offsets = np.asarray(offsets, dtype=np.int32)
        n_depths = len(offsets) - 1

        is_tip = self.n_children == 0
        self.is_tip = is_tip
        self.tips = np.flatnonzero(is_tip).astype(np.int32)
        self.n_tips = len(self.tips)

        height = np.zeros(N, dtype=np.int32)
        for lvl in range(n_depths - 1, 0, -1):
            a, b = offsets[lvl], offsets[lvl + 1]
            np.maximum.at(height, parent[a:b], height[a:b] + 1)
        self.height = height

        internal = np.flatnonzero(~is_tip).astype(np.int32)
        order = np.lexsort((internal, height[internal]))
        self.bu_order = internal[order].astype(np.int32)
        H = int(height.max()) if N else 0
        counts = np.bincount(height[internal], minlength=H + 1)[1:] if H > 0 else np.zeros(0, dtype=np.int64)
        self.bu_offsets = np.concatenate(([0], np.cumsum(counts))).astype(np.int32)

        self.td_parents = internal
        pd_counts = np.bincount(depth[internal], minlength=n_depths) if len(internal) else np.zeros(n_depths, int)
        self.td_parent_offsets = np.concatenate(([0], np.cumsum(pd_counts))).astype(np.int32)

        # subtree sizes -> preorder -> postorder ranks (post = pre - depth + size - 1 within a tree)
        size = np.ones(N, dtype=np.int64)
        for lvl in range(n_depths - 1, 0, -1):
            a, b = offsets[lvl], offsets[lvl + 1]
            np.add.at(size, parent[a:b], size[a:b])
        pre = np.zeros(N, dtype=np.int64)
        tree_sizes = size[:R]
        tree_offset = np.concatenate(([0], np.cumsum(tree_sizes)))[:-1]
        for lvl in range(1, n_depths):
            a, b = offsets[lvl], offsets[lvl + 1]
            # exclusive cumsum of sibling sizes within each parent group
            sz = size[a:b]
            cs = np.cumsum(sz) - sz
            p = parent[a:b]
            first = self.first_child[p] - a  # index of the first sibling inside this level
            pre[a:b] = pre[p] + 1 + (cs - cs[first])
        post = pre - depth + size - 1
        self.post_rank = (post + tree_offset[tree_id]).astype(np.int32)
        self.subtree_size = size

    # ------------------------------------------------------------------------------------------------------------------
    @property
    def n_bu_levels(self):
        return len(self.bu_offsets) - 1

    @property
    def n_td_levels(self):
        return len(self.td_offsets) - 1

    @property
    def forest_length(self):
        return float(self.dist.sum())

    def to_tree_nodes(self, names=None):
        """Builds TreeNode objects (and remembers them in ``self.nodes``); names default to ROOT / n<i> / t<i>."""
        nodes = []
        for i in range(self.n_nodes):
            if names is not None:
                name = names[i]
            elif self.parent[i] < 0:
                name = 'ROOT' if len(self.roots) == 1 else 'ROOT{}'.format(i)
            else:
                name = ('t{}' if self.n_children[i] == 0 else 'n{}').format(i)
            node = TreeNode(name=name, dist=float(self.dist[i]))
            nodes.append(node)
            if self.parent[i] >= 0:
                nodes[self.parent[i]].add_child(node)
        self.nodes = nodes
        self.adopt_nodes()
        return [nodes[r] for r in self.roots]

    # ------------------------------------------------------------------------------------------------------------------
    @classmethod
    def from_trees(cls, forest):
        """Flattens TreeNode trees (one tree or a list). Node ids follow forest-wide level order."""
        if isinstance(forest, TreeNode):
            forest = [forest]
        nodes = list(forest)
        parent = [-1] * len(nodes)
        i = 0
        first_child, n_children, dist = [], [], []
        while i < len(nodes):
            n = nodes[i]
            first_child.append(len(nodes))
            n_children.append(len(n.children))
            dist.append(n.dist)
            for c in n.children:
                nodes.append(c)
                parent.append(i)
            i += 1
        return cls(parent, n_children, first_child, dist, np.arange(len(forest)), nodes=nodes)

    @classmethod
    def balanced(cls, n_levels, seed=42, lo=0.01, hi=0.2):
        """
        Perfectly balanced binary tree with 2**n_levels tips: ids in level order, root
        dist 0, branch lengths uniform(lo, hi) drawn in id order from default_rng(seed).
        """
        T = 1 << n_levels
        N = 2 * T - 1
        ids = np.arange(N, dtype=np.int64)
        parent = ((ids - 1) // 2).astype(np.int32)
        parent[0] = -1
        n_children = np.where(ids < T - 1, 2, 0).astype(np.int32)
        first_child = (2 * ids + 1).astype(np.int32)
        rng = np.random.default_rng(seed)
        dist = rng.uniform(lo, hi, size=N)
        dist[0] = 0.0
        return cls(parent, n_children, first_child, dist, np.array([0]))

    @classmethod
    def random(cls, n_tips, seed=0, max_arity=2, zero_frac=0.0, lo=0.001, hi=0.3, n_trees=1):
        """
        Random (unbalanced) forest for tests: grows by splitting random tips; arity in [2, max_arity];
        a fraction zero_frac of the branches gets length 0.
        """
        rng = np.random.default_rng(seed)
        roots = []
        for _ in range(n_trees):
            root = TreeNode(name='', dist=0.0)
            leaves = [root]
            while len(leaves) < max(2, n_tips // n_trees):
                idx = int(rng.integers(len(leaves)))
                leaf = leaves.pop(idx)
                for _c in range(int(rng.integers(2, max_arity + 1))):
                    d = 0.0 if rng.random() < zero_frac else float(rng.uniform(lo, hi))
                    leaves.append(leaf.add_child(dist=d))
            roots.append(root)
        for ti, root in enumerate(roots):
            for i, n in enumerate(root.traverse('preorder')):
                n.name = 't{}_{}'.format(ti, i) if n.is_leaf() else 'n{}_{}'.format(ti, i)
        return cls.from_trees(roots)

    def renumbered(self, new_of_old):
        """
        The same forest under another breadth-first numbering (new_of_old[i] = the new id of node i; the order of the sibling
        groups inside a depth may differ, a node's own children keep their order) -- e.g. the numbering the device library
        works in, ``Engine.node_order()``.  Arrays only: ``nodes`` and columns stay with this forest.
        """
        new_of_old = np.asarray(new_of_old, dtype=np.int64)
        old_of_new = np.empty_like(new_of_old)
        old_of_new[new_of_old] = np.arange(self.n_nodes)
        parent = self.parent[old_of_new].astype(np.int64)
        parent = np.where(parent >= 0, new_of_old[np.maximum(parent, 0)], -1)
        n_children = self.n_children[old_of_new]
        first_child = np.where(n_children > 0, new_of_old[np.minimum(self.first_child[old_of_new], self.n_nodes - 1)], 0)
        return FlatForest(parent, n_children, first_child, self.dist[old_of_new], np.arange(len(self.roots)))

    def children_of(self, i):
        a = self.first_child[i]
        return range(a, a + self.n_children[i])

    def postorder_ids(self):
        """Node ids in the reference's processing order."""
        return np.argsort(self.post_rank, kind='stable').astype(np.int32)


_TRUST_FLAT_CACHE = [0]


class trusted_flat_cache(object):
    """
    ``with trusted_flat_cache():`` -- inside, get_flat_forest hands out a cached forest without walking the tree to see
    whether somebody edited it since.  For code that owns its trees for the duration (pastml_pipeline between reading
    and writing: the walk is a third of a second per call at 5e5 nodes, and the stages ask five times).
    """

    def __enter__(self):
        _TRUST_FLAT_CACHE[0] += 1
        return self

    def __exit__(self, *exc):
        _TRUST_FLAT_CACHE[0] -= 1


def get_flat_forest(forest):
    """
    Returns the (cached) FlatForest of a list of TreeNode roots. The cache lives on the first root and is keyed by
    the identity of the roots, the number of nodes and the branch lengths, so that editing a tree invalidates it.
    """
    if isinstance(forest, TreeNode):
        forest = [forest]
    holder = forest[0]
    key_roots = tuple(id(t) for t in forest)
    cache = getattr(holder, '_flat_cache', None)
    if cache is not None and cache[0] == key_roots:
        flat = cache[1]
        if _TRUST_FLAT_CACHE[0] > 0:
            return flat
        # validation: same node objects, same child lists, same branch lengths
        nodes = flat.nodes
        ok = all(nodes[r] is t for r, t in enumerate(forest))
        if ok:
            dist, n_children, first_child = flat.dist, flat.n_children, flat.first_child
            for i, n in enumerate(nodes):
                ch = n.children
                if n.dist != dist[i] or len(ch) != n_children[i]:
                    ok = False
                    break
                fc = first_child[i]
                for j, c in enumerate(ch):
                    if nodes[fc + j] is not c:
                        ok = False
                        break
                if not ok:
                    break
        if ok:
            return flat
    flat = FlatForest.from_trees(forest)
    # ids must follow per-tree level order for single trees; for forests the order is forest-wide level order
    holder._flat_cache = (key_roots, flat)
    return flat
