"""
File-to-file entry point around the likelihood path: ``pastml_pipeline`` (pastml/acr.py:316-674) without its
visualisation half.  Reads the tree(s) (newick, or the trees block of a nexus file) and the annotation table
(pastml/acr.py:695-826, pastml/tree.py:176-222), runs the batched ``acr()`` on the GPU and writes, in the reference's
formats and under the reference's file names, so that the outputs are interchangeable with PastML's:

* ``params.character_<c>.method_<m>.model_<M>.tab``        statistics and model parameters per character
* ``marginal_probabilities.character_<c>.model_<M>.tab``   marginal posteriors per node (marginal methods)
* ``combined_ancestral_states.tab``                          the selected states of every node, all characters
* ``named.tree_<tree>.nwk``                                  the input tree(s) with every node named

HTML maps, iTOL upload, dates / timelines and polytomy resolution belong to PastML's visualisation and tree-editing
layers (out of scope): asking for them raises NotImplementedError.  (The parsimony methods and COPY
are host-side array passes of acr(); they run here too.)
"""
import logging
import os
import re

import numpy as np
import pandas as pd

from pastml_amd import CHARACTER, STATES
from pastml_amd.acr import acr, _serialize_acr, COPY
from pastml_amd.annotation import preannotate_forest
from pastml_amd.file import col_name2cat
from pastml_amd.ml import MPPA
from pastml_amd.models.F81Model import F81
from pastml_amd.tree import StateSetColumn, _DICT_FEATURE_NAMES, read_tree, name_tree, get_flat_forest, trusted_flat_cache

PASTML_WORK_DIR = '{tree}_pastml'
COMBINED_ANCESTRAL_STATE_TAB = 'combined_ancestral_states.tab'
NAMED_TREE_NWK = 'named.tree_{tree}.nwk'


def get_pastml_work_dir(tree):
    """pastml/file.py:68-75."""
    return PASTML_WORK_DIR.format(tree=os.path.splitext(tree)[0])


def get_named_tree_file(tree):
    """pastml/file.py:78-87."""
    name = os.path.splitext(os.path.basename(tree))[0]
    return NAMED_TREE_NWK.format(tree=name if name else 'tree')


def get_combined_ancestral_state_file():
    """pastml/file.py:58-65."""
    return COMBINED_ANCESTRAL_STATE_TAB


# ---------------------------------------------------------------------------------------------------------------------
# trees
# ---------------------------------------------------------------------------------------------------------------------
_NEXUS_TREE = re.compile(r'^\s*tree\s+[^=]+=\s*(?:\[[^\]]*\]\s*)*(.*;)\s*$', re.I)


def _nexus_newicks(text):
    """Newick strings of the ``trees`` block of a nexus file, taxon numbers replaced through its translate table."""
    if not text.lstrip().lower().startswith('#nexus'):
        return None
    block = re.search(r'begin\s+trees\s*;(.*?)end\s*;', text, re.I | re.S)
    if not block:
        return []
    body = block.group(1)
    translate = {}
    table = re.search(r'translate(.*?);', body, re.I | re.S)
    if table:
        for entry in table.group(1).split(','):
            parts = entry.split()
            if len(parts) >= 2:
                translate[parts[0]] = ' '.join(parts[1:]).strip("'\"")
        body = body[:table.start()] + body[table.end():]
    newicks = []
    for statement in body.split(';'):
        m = _NEXUS_TREE.match(statement.replace('\n', ' ') + ';')
        if not m:
            continue
        nwk = m.group(1)
        if translate:
            nwk = re.sub(r'(?<=[(,])\s*([^\s:(),;\[\]]+)(?=[:,)\[])',
                         lambda t: translate.get(t.group(1), t.group(1)), nwk)
        newicks.append(nwk)
    return newicks


def read_forest(tree_path, columns=None):
    """
    All the trees of a newick or nexus file; negative branch lengths become zero (pastml/tree.py:176-199).  With
    ``columns`` the annotations are looked for in the tree itself: NHX-style comments ``[&&NHX:column=a|b]`` / nexus
    comments ``[&column="a|b"]`` behind a node give ``node.<column> = {'a', 'b'}``.
    """
    logger = logging.getLogger('pastml')
    with open(tree_path, 'r') as f:
        text = f.read()
    newicks = _nexus_newicks(text)
    if newicks is None:
        newicks = [nwk + ';' for nwk in text.replace('\n', '').split(';')[:-1]]
    if not newicks:
        raise ValueError('Could not find any trees (in newick or nexus format) in the file {}.'.format(tree_path))
    roots = []
    for nwk in newicks:
        root = read_tree(nwk)
        if columns:
            annotate_from_comments(root, columns)
        roots.append(root)
    negative = 0
    for root in roots:
        for n in root.traverse():
            if n.dist < 0:
                negative += 1
                n.dist = 0
    if negative:
        logger.warning('Input tree{} contained {} negative branches: we put them to zero.'
                       .format('s' if len(roots) > 1 else '', negative))
    logger.debug('Read the tree{} {}.'.format('s' if len(roots) > 1 else '', tree_path))
    return roots


def annotate_from_comments(root, columns):
    """``[&&NHX:column=a|b]`` / ``[&column="a|b"]`` comments kept by the newick reader -> ``node.<column> = {a, b}``."""
    patterns = {c: re.compile(r'(?:^|[:,&\s]){}="?([^,"\]:]*)"?'.format(re.escape(c))) for c in columns}
    for node in root.traverse():
        comment = node.__dict__.pop('comment', None)
        if not comment:
            continue
        for c, pattern in patterns.items():
            values = set()
            for m in pattern.finditer(comment):
                values |= {v for v in m.group(1).split('|') if v != ''}
            if values:
                node.add_feature(c, values)


# ---------------------------------------------------------------------------------------------------------------------
# input validation
# ---------------------------------------------------------------------------------------------------------------------
def _quote(names):
    return ', '.join('"{}"'.format(_) for _ in names) if names is not None else ''


def read_annotation_table(data, data_sep='\t', id_index=0, columns=None):
    """The annotation table as strings, indexed by node name, columns renamed to feature-safe names (acr.py:712-727)."""
    df = pd.read_csv(data, sep=data_sep, index_col=id_index, header=0, dtype=str)
    df.index = df.index.map(str)
    if columns:
        unknown = set(columns) - set(df.columns)
        if unknown:
            raise ValueError('{} of the specified columns ({}) {} not found among the annotation columns: {}.'
                             .format('One' if len(unknown) == 1 else 'Some', _quote(unknown),
                                     'is' if len(unknown) == 1 else 'are', _quote(df.columns)))
        df = df[columns]
    df.columns = [col_name2cat(c) for c in df.columns]
    return df


def validate_input(tree_nwk, columns=None, data=None, data_sep='\t', id_index=0, copy_only=False, parameters=None,
                   rates=None):
    """
    Reads and checks tree(s) and annotations (pastml/acr.py:695-826 without the date handling): returns
    (roots, columns, column2states, parameters, rates) with the trees annotated and every node named.
    """
    logger = logging.getLogger('pastml')
    logger.debug('\n=============INPUT DATA VALIDATION=============')
    if not columns and data is None:
        raise ValueError("If you don't provide the metadata file, "
                         "you need to provide an annotated tree and specify the columns argument, "
                         "which will be used to look for character annotations in your input tree.")
    if columns and isinstance(columns, str):
        columns = [columns]
    roots = read_forest(tree_nwk, columns=columns if data is None else None)
    column2states = {}
    flat = get_flat_forest(roots)
    names = np.array([n.name for n in flat.nodes], dtype=object)
    if data:
        df = read_annotation_table(data, data_sep, id_index, columns)
        logger.debug('Read the annotation file {}.'.format(data))
        columns = list(df.columns)
        ids = set(df.index)
        if not ids & set(names[names != '']):
            stripped = [name.strip("'").strip('"') for name in names]
            if ids & set(stripped):
                for node, name in zip(flat.nodes, stripped):
                    node.name = name
                names = np.array(stripped, dtype=object)
                flat._node_names = None
        if not ids & set(names[names != '']):
            tips = [flat.nodes[i].name for i in flat.tips[:3]]
            raise ValueError('Your tree tip names (e.g. {}) do not correspond to annotation id column values (e.g. {}). '
                             'Check your annotation file.'.format(', '.join(tips), ', '.join(list(ids)[:3])))
        logger.debug('Checked that (at least some of) tip names correspond to annotation file index.')
        preannotate_forest(roots, df=df)
        for c in columns:
            column2states[c] = {_ for _ in df[c].unique() if pd.notnull(_) and _ != ''}
    else:
        columns = [col_name2cat(c) for c in columns]
        column2states = {c: set() for c in columns}
    # how well are the tips annotated, and is the character discrete enough (acr.py:744-781)
    from pastml_amd.batch import annotation_words, popcount
    n_tips = flat.n_tips
    annotated_tips, annotated_states = {}, {}
    for c in columns:
        if not data:
            for node in flat.nodes:
                column2states[c] |= getattr(node, c, set())
        states = np.array(sorted(column2states[c]))
        words, _ = annotation_words(flat, c, states)
        given = words.any(axis=-1)
        # annotated nodes of any kind count (acr.py:747-754), against the number of tips
        annotated_tips[c] = int(given.sum())
        present = np.bitwise_or.reduce(words[given], axis=0) if given.any() else np.zeros(words.shape[1], np.uint64)
        annotated_states[c] = int(popcount(present).sum())
    # the least annotated column among those with any annotation (acr.py:758-761: columns without one are not in the
    # reference's counter); the first column, with none, if no column has any
    some = [_ for _ in columns if annotated_tips[_] > 0]
    c = min(some, key=lambda _: annotated_tips[_]) if some else columns[0]
    unknown = (n_tips - annotated_tips[c]) / n_tips
    if unknown >= (.9 if not copy_only else 1):
        raise ValueError('{:.1f}% of tip annotations for character "{}" are unknown, '
                         'not enough data to infer ancestral states. {}'
                         .format(unknown * 100, c,
                                 'Check your annotation file and if its ids correspond to the tree tip/node names.'
                                 if data else 'You tree file should contain character state annotations, '
                                              'otherwise consider specifying a metadata file.'))
    c = min(columns, key=lambda _: annotated_states[_])
    if annotated_states[c] > n_tips * .75 and not copy_only:
        raise ValueError('Character "{}" has {} unique states annotated in this tree: {}, '
                         'which is too much to infer on a {} with only {} tips. '
                         'Make sure the character you are analysing is discrete, and if yes use a larger tree.'
                         .format(c, annotated_states[c], column2states[c], 'tree' if len(roots) == 1 else 'forest',
                                 n_tips))
    logger.debug('Finished input validation.')
    column2states = {c: np.array(sorted(states)) for c, states in column2states.items()}

    def per_column(what, label):
        if not what:
            return {}
        if isinstance(what, str):
            what = [what]
        if isinstance(what, list):
            return dict(zip(columns, what))
        if isinstance(what, dict):
            return {col_name2cat(col): v for col, v in what.items()}
        raise ValueError('{} should be either a list or a dict, got {}.'.format(label, type(what)))

    parameters = per_column(parameters, 'Parameters')
    rates = per_column(rates, 'Rate matrices')
    for i, tree in enumerate(roots):
        name_tree(tree, suffix='' if len(roots) == 1 else '_{}'.format(i))
    if getattr(flat, '_node_names', None) is not None:
        flat._node_names = None   # the names may just have changed
    return roots, columns, column2states, parameters, rates


# ---------------------------------------------------------------------------------------------------------------------
# outputs
# ---------------------------------------------------------------------------------------------------------------------
def serialize_predicted_states(columns, out_data, roots):
    """
    ``combined_ancestral_states.tab`` (pastml/acr.py:831-858): one line per node with its state in every column; a node
    with several states in some column takes several lines, the columns' states listed in ascending order and the
    exhausted columns left empty.  Rows follow the trees one after another, each in level order.
    Columns that live in the flat forest as packed state sets (what acr() leaves) are expanded array-wise: at 10^5 tips
    the per-node attribute loop of the reference is the longest stage of the pipeline.
    """
    from pastml_amd.hip import unpack_masks
    flat = get_flat_forest(roots)
    N = flat.n_nodes
    order = np.arange(N) if len(flat.roots) == 1 else np.lexsort((np.arange(N), flat.tree_id))
    names = np.array([n.name for n in flat.nodes], dtype=object)
    # per column: number of states of every node and, line by line, the state names
    counts = np.zeros((len(columns), N), dtype=np.int64)
    cells = []   # per column: function line -> object array [N] of the line's entries ('' where exhausted)
    for ci, c in enumerate(columns):
        col = flat.columns.get(c) if hasattr(flat, 'columns') else None
        # (a value put on a single node with add_feature hides the column there: such names take the per-node path)
        if isinstance(col, StateSetColumn) and c not in _DICT_FEATURE_NAMES and col.absent is None:
            states = np.array([str(x) for x in col.states], dtype=object)
            ascending = np.argsort(states, kind='stable')           # lines list a node's states in ascending order
            bits = unpack_masks(col.words, len(states))[:, ascending].astype(bool)
            rank = np.cumsum(bits, axis=1)
            counts[ci] = rank[:, -1]
            sorted_states = states[ascending]

            def line_cells(line, bits=bits, rank=rank, sorted_states=sorted_states):
                hit = bits & (rank == line + 1)
                out = np.full(N, '', dtype=object)
                has = hit.any(axis=1)
                out[has] = sorted_states[hit[has].argmax(axis=1)]
                return out
        else:
            values = [sorted(getattr(node, c, set())) for node in flat.nodes]
            counts[ci] = [len(v) for v in values]

            def line_cells(line, values=values):
                return np.array([str(v[line]) if line < len(v) else '' for v in values], dtype=object)
        cells.append(line_cells)
    lines = counts.max(axis=0) if len(columns) else np.zeros(N, dtype=np.int64)
    per_line = [[fn(line) for fn in cells] for line in range(int(lines.max()) if N else 0)]
    with open(out_data, 'w+') as f:
        f.write('node\t{}\n'.format('\t'.join(columns)))
        if N and per_line:
            ordered_lines = lines[order]
            node_of_row = np.repeat(order, ordered_lines)
            starts = np.cumsum(ordered_lines) - ordered_lines
            line_of_row = np.arange(len(node_of_row)) - np.repeat(starts, ordered_lines)
            table = np.empty((len(node_of_row), len(columns) + 1), dtype=object)
            table[:, 0] = names[node_of_row]
            for line, row_cells in enumerate(per_line):
                sel = line_of_row == line
                for ci, values in enumerate(row_cells):
                    table[sel, ci + 1] = values[node_of_row[sel]]
            f.write('\n'.join('\t'.join(row) for row in table.tolist()))
            f.write('\n')
    logging.getLogger('pastml').debug('Serialized reconstructed states to {}.'.format(out_data))


def pastml_pipeline(tree, data=None, data_sep='\t', id_index=0, columns=None, prediction_method=MPPA, model=F81,
                    parameters=None, rate_matrix=None, name_column=None, root_date=None, timeline_type=None,
                    tip_size_threshold=None, colours=None, out_data=None, html_compressed=None, html=None,
                    html_mixed=None, work_dir=None, verbose=False, forced_joint=False, upload_to_itol=False,
                    itol_id=None, itol_project=None, itol_tree_name=None, offline=False, threads=0, reoptimise=False,
                    focus=None, resolve_polytomies=False, smoothing=False, frequency_smoothing=False, pajek=None,
                    pajek_timing=None, recursion_limit=0):
    """
    Reads tree(s) and annotations, reconstructs the ancestral states of all the characters in one batched ``acr()`` call
    on the GPU and writes the result tables into ``work_dir`` (default ``<tree>_pastml``).  Arguments as in
    pastml/acr.py:316-327; those of the visualisation layer (html*, iTOL, colours, focus, timeline, pajek) and
    ``resolve_polytomies`` are not available here.  Returns the list of result dictionaries.
    """
    logger = logging.getLogger('pastml')
    if verbose:
        logging.basicConfig(level=logging.DEBUG, format='%(asctime)s: %(message)s', datefmt='%H:%M:%S')
        logger.setLevel(logging.DEBUG)
    asked = [name for name, value in (('html', html), ('html_compressed', html_compressed), ('html_mixed', html_mixed),
                                       ('upload_to_itol', upload_to_itol), ('pajek', pajek), ('root_date', root_date),
                                       ('resolve_polytomies', resolve_polytomies)) if value]
    if asked:
        raise NotImplementedError('{}: visualisation, dating and tree editing are PastML\'s own layers; this pipeline '
                                  'covers tree + table -> reconstruction -> result tables'.format(', '.join(asked)))
    copy_only = COPY == prediction_method or (isinstance(prediction_method, list)
                                              and all(COPY == _ for _ in prediction_method))
    with trusted_flat_cache():   # the trees are this function's own from reading to writing
        return _pipeline(tree, data, data_sep, id_index, columns, prediction_method, model, parameters, rate_matrix,
                         out_data, work_dir, forced_joint, threads, reoptimise, smoothing, frequency_smoothing, copy_only)


def _pipeline(tree, data, data_sep, id_index, columns, prediction_method, model, parameters, rate_matrix, out_data,
              work_dir, forced_joint, threads, reoptimise, smoothing, frequency_smoothing, copy_only):
    roots, columns, column2states, parameters, rates = \
        validate_input(tree, columns, data, data_sep, id_index, copy_only=copy_only, parameters=parameters,
                       rates=rate_matrix)
    if not work_dir:
        work_dir = get_pastml_work_dir(tree)
    os.makedirs(work_dir, exist_ok=True)
    results = acr(forest=roots, columns=columns, column2states=column2states, prediction_method=prediction_method,
                  model=model, column2parameters=parameters, column2rates=rates, force_joint=forced_joint,
                  threads=threads, reoptimise=reoptimise, tau=None if smoothing else 0,
                  frequency_smoothing=frequency_smoothing)
    characters = sorted({r[CHARACTER]: r[STATES] for r in results}.keys())
    # Several processes (one per GPU, pastml_amd.sharding): acr() returned this rank's block of the characters.  The
    # per-character tables are one file each and do not meet; the combined table holds this rank's columns and carries
    # the rank in its name (joining them is a column-wise paste: every rank writes the same nodes in the same order);
    # the named tree is the same everywhere and written by rank 0.
    from pastml_amd import sharding
    comm = sharding.communicator()
    rank, world = (comm.rank, comm.world) if comm is not None else (0, 1)
    combined = out_data or os.path.join(work_dir, get_combined_ancestral_state_file())
    if world > 1:
        stem, ext = os.path.splitext(combined)
        combined = '{}.rank{}of{}{}'.format(stem, rank, world, ext)
    serialize_predicted_states(characters, combined, roots)
    if rank == 0:
        with open(os.path.join(work_dir, get_named_tree_file(tree)), 'w+') as f:
            f.write('\n'.join(root.write() for root in roots))
    for r in results:
        _serialize_acr((r, work_dir))
    return results
