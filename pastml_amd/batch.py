"""
Characters as columns: the host side of the maximum-likelihood reconstruction for SEVERAL characters of one forest at
once.

The reference analyses one character at a time (pastml/acr.py:213-231 hands them to a pool one by one; pastml/ml.py
keeps everything in per-node attributes).  The device sweeps have a column axis with independent masks and parameters
per column, so here the characters of a run that share the number of states and the model family become the columns
of ONE device context and every step of ``ml_acr`` (pastml/ml.py:640-750) -- every likelihood evaluation of every
character's L-BFGS-B run (:174-237), the joint sweep, the marginal pass, the state selections, the restricted
likelihoods -- is one batched launch sequence for all of them:

* :class:`CharacterBatch` -- allowed-state masks of m characters as packed bit words ``uint64[m, N, W]`` with the
  zero-branch alteration / restoration rules of ml.py:321-405 as array operations over all characters and zero-branch
  clusters at once, plus the device context holding the m columns;
* :class:`SweepServer` -- the optimisers of the m characters run concurrently (one scipy L-BFGS-B instance each, own
  thread); a likelihood request blocks until every running optimiser has posted one, then a single bottom-up sweep over
  sum(n_params + 1) columns serves them all.  Columns are computed independently and deterministically, so each
  character sees exactly the numbers -- and takes exactly the iterates -- of a run on its own;
* :func:`reconstruct` -- the rest of ``ml_acr`` in lock-step.

Node features the reference leaves on the tree are written as columnar features of the flat forest
(pastml_amd.tree), O(1) per feature instead of a tree walk.
"""
import logging
import os
import threading

import numpy as np
import pandas as pd

from pastml_amd import hip
from pastml_amd.models import PointBlock
from pastml_amd.tree import (TreeNode, get_flat_forest, AnnotationColumn, StateSetColumn, MaskColumn, ArrayColumn,
                             _DICT_FEATURE_NAMES)

ONE = np.uint64(1)


# =====================================================================================================================
# packed state sets
# =====================================================================================================================
def n_words(k):
    return (k + 63) // 64


def full_words(k):
    """All k states allowed: uint64[W]."""
    W = n_words(k)
    w = np.full(W, ~np.uint64(0), dtype=np.uint64)
    if k % 64:
        w[-1] = (ONE << np.uint64(k % 64)) - ONE
    return w


def words_from_masks(masks, k):
    """0/1 array [..., k] -> uint64 [..., W]."""
    return hip.pack_masks(masks, k)


def masks_from_words(words, k):
    """uint64 [..., W] -> int8 0/1 [..., k]."""
    return hip.unpack_masks(words, k)


def popcount(words):
    """Number of set bits per entry of a uint64 array."""
    b = np.ascontiguousarray(words).view(np.uint8)
    return _POP8[b].reshape(words.shape + (8,)).sum(axis=-1)


_POP8 = np.array([bin(i).count('1') for i in range(256)], dtype=np.int64)


def one_hot_words(index, k):
    """State indices [...] -> words [..., W] with that one bit set."""
    index = np.asarray(index, dtype=np.int64)
    W = n_words(k)
    out = np.zeros(index.shape + (W,), dtype=np.uint64)
    np.put_along_axis(out, (index >> 6)[..., None], (ONE << (index & 63).astype(np.uint64))[..., None], axis=-1)
    return out


# =====================================================================================================================
# zero-branch clusters (tree level)
# =====================================================================================================================
class ZeroClusters(object):
    """
    Nodes joined by zero-length branches (pastml/ml.py:321-349), for clusters of at least two nodes: ``nodes`` lists
    their members cluster by cluster, ``starts`` where each cluster begins in that list, ``cluster_of`` the cluster of
    every listed member.
    """

    def __init__(self, flat):
        N = flat.n_nodes
        top = np.arange(N, dtype=np.int64)
        zero = (flat.dist == 0) & (flat.parent >= 0)
        # parents precede children in id order, level by level
        for lvl in range(1, flat.n_td_levels):
            a, b = flat.td_offsets[lvl], flat.td_offsets[lvl + 1]
            ids = np.arange(a, b)[zero[a:b]]
            top[ids] = top[flat.parent[ids]]
        self.top = top
        size = np.bincount(top, minlength=N)
        members = np.flatnonzero(size[top] >= 2)
        order = np.argsort(top[members], kind='stable')
        self.nodes = members[order]
        tops = top[self.nodes]
        self.starts = np.flatnonzero(np.concatenate(([True], tops[1:] != tops[:-1]))) if len(tops) else \
            np.zeros(0, dtype=np.int64)
        self.cluster_of = np.cumsum(np.concatenate(([0], (tops[1:] != tops[:-1]).astype(np.int64)))) if len(tops) else \
            np.zeros(0, dtype=np.int64)


def zero_clusters(flat):
    zc = getattr(flat, '_zero_clusters', None)
    if zc is None:
        zc = ZeroClusters(flat)
        flat._zero_clusters = zc
    return zc


# =====================================================================================================================
# annotation of a character as packed words
# =====================================================================================================================
def annotation_words(flat, character, states):
    """
    (words uint64[N, W], annotated bool[N]) of a character: the node's states as given (0 words: none given) and
    whether the node "has a state" in the sense of pastml/ml.py:329-331 (the attribute exists and is not '': an empty
    set counts).  Columnar annotations (pastml_amd.annotation.preannotate_forest, an earlier reconstruction) are read as
    arrays; attributes set node by node are collected with one walk.
    """
    states = np.asarray(states)
    k = len(states)
    N = flat.n_nodes
    W = n_words(k)
    col = flat.columns.get(character)
    shadowed = character in _DICT_FEATURE_NAMES and flat.nodes is not None and \
        any(character in n.__dict__ for n in flat.nodes)
    if col is not None and not shadowed and col.absent is None:
        if isinstance(col, AnnotationColumn):
            state2index = {s: i for i, s in enumerate(states)}
            lut = np.array([state2index.get(v, -1) for v in col.values] + [-1], dtype=np.int64)
            codes = col.codes
            annotated = codes > -2
            idx = lut[np.where(codes >= 0, codes, len(col.values))]
            words = np.zeros((N, W), dtype=np.uint64)
            has = idx >= 0
            words[has] = one_hot_words(idx[has], k)
            for i, vs in col.multi.items():
                words[i] = 0
                for j in vs:
                    if lut[j] >= 0:
                        words[i, lut[j] >> 6] |= ONE << np.uint64(lut[j] & 63)
            return words, annotated
        if isinstance(col, StateSetColumn) and len(col.states) == k and np.array_equal(col.states, states):
            return np.array(col.words, dtype=np.uint64, copy=True), np.ones(N, dtype=bool)
    if flat.nodes is None:
        return np.zeros((N, W), dtype=np.uint64), np.zeros(N, dtype=bool)
    state2index = {s: i for i, s in enumerate(states)}
    words = np.zeros((N, W), dtype=np.uint64)
    annotated = np.zeros(N, dtype=bool)
    for i, node in enumerate(flat.nodes):
        value = getattr(node, character, None)
        if value is not None and value != '':
            annotated[i] = True
        if value:
            for s in value:
                j = state2index[s]
                words[i, j >> 6] |= ONE << np.uint64(j & 63)
    return words, annotated


# =====================================================================================================================
# the batch
# =====================================================================================================================
class LikelihoodError(Exception):
    """Zero likelihood in a column: (column, parent id, child id); turned into PastMLLikelihoodError by the caller."""

    def __init__(self, column, parent, child):
        Exception.__init__(self, 'zero likelihood in column {} between nodes {} and {}'.format(column, parent, child))
        self.column, self.parent, self.child = column, parent, child


class CharacterBatch(object):
    """m characters with k states each on one flat forest: masks on the host, columns on the device."""

    def __init__(self, flat, k, m, device=None):
        self.flat = flat
        self.k, self.m = k, m
        self.N = flat.n_nodes
        self.W = n_words(k)
        self.full = full_words(k)
        self._device = device
        self._engine = None
        self._opt = None
        self.ann = np.zeros((m, self.N, self.W), dtype=np.uint64)
        self.annotated = np.zeros((m, self.N), dtype=bool)
        self.masks = np.broadcast_to(self.full, (m, self.N, self.W)).copy()
        self.init_masks = np.zeros((m, self.N, self.W), dtype=np.uint64)
        self.has_init = np.zeros((m, self.N), dtype=bool)
        self._uploaded = None
        self._uploaded_models = None
        self.n_sweeps = 0

    # ------------------------------------------------------------------------------------------------ resources
    @property
    def engine(self):
        """The device context of the m columns, created on first use (mask bookkeeping alone needs no GPU)."""
        if self._engine is None:
            self._engine = hip.acquire_engine(self.flat, self.m, self.k, device=self._device)
            self._uploaded = None
            self._uploaded_models = None
        return self._engine

    def close(self):
        if self._engine is not None:
            hip.release_engine(self._engine)
            self._engine = None
        if self._opt is not None:
            hip.release_engine(self._opt['engine'])
            self._opt = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------------------------------------ masks (host)
    def set_annotation(self, c, words, annotated):
        self.ann[c] = words
        self.annotated[c] = annotated

    def initialize_allowed_states(self, rows=None):
        """Annotated states where there are any, everything allowed elsewhere (pastml/ml.py:293-318)."""
        rows = slice(None) if rows is None else rows
        given = self.ann[rows].any(axis=-1, keepdims=True)
        self.masks[rows] = np.where(given, self.ann[rows], self.full)

    def alter(self, rows):
        """
        Zero-branch alteration (pastml/ml.py:352-387) for the characters whose ``rows`` flag is set: in every cluster of
        nodes joined by zero-length branches, the annotated members, if they are at least two and share no allowed
        state, all get the union of their masks; the masks they had are remembered.  All characters and clusters at
        once.  Returns the altered (character, node) pairs as a bool array [m, N].
        """
        altered = np.zeros((self.m, self.N), dtype=bool)
        zc = zero_clusters(self.flat)
        rows = np.asarray(rows, dtype=bool)
        if not len(zc.nodes) or not rows.any():
            return altered
        r = np.flatnonzero(rows)
        a = self.annotated[np.ix_(r, zc.nodes)]                          # [r, n_zc]
        mk = self.masks[np.ix_(r, zc.nodes)]                             # [r, n_zc, W]
        count = np.add.reduceat(a.astype(np.int64), zc.starts, axis=1)   # annotated members per cluster
        common = np.bitwise_and.reduceat(np.where(a[..., None], mk, ~np.uint64(0)), zc.starts, axis=1)
        union = np.bitwise_or.reduceat(np.where(a[..., None], mk, np.uint64(0)), zc.starts, axis=1)
        clash = (count >= 2) & ~common.any(axis=-1)                      # [r, n_clusters]
        hit = clash[:, zc.cluster_of] & a                                # [r, n_zc]
        if not hit.any():
            return altered
        ri, ni = np.nonzero(hit)
        chars, nodes = r[ri], zc.nodes[ni]
        self.init_masks[chars, nodes] = mk[ri, ni]
        self.has_init[chars, nodes] = True
        self.masks[chars, nodes] = union[ri, zc.cluster_of[ni]]
        altered[chars, nodes] = True
        return altered

    def unalter(self, altered):
        """masks & saved masks, or the saved ones where nothing is left (pastml/ml.py:390-405)."""
        if not altered.any():
            return
        both = self.masks[altered] & self.init_masks[altered]
        self.masks[altered] = np.where(both.any(axis=-1, keepdims=True), both, self.init_masks[altered])

    # ------------------------------------------------------------------------------------------------ device
    def _upload_models(self, engine, models, col_begin=0, cache_attr='_uploaded_models'):
        keys = []
        for mdl in models:
            spec = mdl.kernel_spec()
            keys.append((spec['kind'], mdl.rate_params(),
                         tuple(np.asarray(v).tobytes() if isinstance(v, np.ndarray) else v
                               for _, v in sorted(spec.items()))))
        if getattr(self, cache_attr) != keys:
            engine.set_models(models, col_begin=col_begin)
            setattr(self, cache_attr, keys)

    def _upload_masks(self):
        eng = self.engine
        if self._uploaded is None:
            eng.set_mask_words(self.masks)
            self._uploaded = self.masks.copy()
            return
        changed = np.flatnonzero((self.masks != self._uploaded).any(axis=(1, 2)))
        # contiguous runs of changed columns go up together
        if len(changed):
            runs = np.split(changed, np.flatnonzero(np.diff(changed) > 1) + 1)
            for run in runs:
                a, b = int(run[0]), int(run[-1]) + 1
                eng.set_mask_words(self.masks[a:b], col_begin=a)
                self._uploaded[a:b] = self.masks[a:b]

    def joint_pass(self, models):
        """Joint sweep (with alteration, ml.py:688-691) and the back-trace behind it, one host round trip
        (``pml_joint_pass``): (ln L [m], joint states int64 [m, N])."""
        return self.bottom_up(models, is_marginal=False, alter=True, backtrace=True)

    def bottom_up(self, models, is_marginal=True, alter=True, errors=None, backtrace=False):
        """
        Bottom-up log-likelihood of every character (pastml/ml.py:82-121 summed over the trees): optional alteration of
        the masks (characters with tau == 0), ONE device sweep over the m columns, restoration of the masks (marginal);
        after a joint sweep the arg-max tables of the altered nodes are rewritten on the device instead
        (ml.py:115-119).  Returns ln L [m]; raises LikelihoodError for the first column without likelihood -- or, if a
        dict is given as ``errors``, records the failing columns there ({column: LikelihoodError}) and goes on.
        """
        rows = np.array([alter and 0 == mdl.tau for mdl in models], dtype=bool)
        before = self.masks.copy() if (rows.any() and not is_marginal) else None
        altered = self.alter(rows) if rows.any() else np.zeros((self.m, self.N), dtype=bool)
        eng = self.engine
        self._upload_models(eng, models)
        self._upload_masks()
        if not is_marginal and altered.any():
            eng.set_initial_mask_words(before)
        else:
            eng.set_initial_masks(None)
        self.n_sweeps += self.m
        joint = None
        try:
            if backtrace:
                lnl, joint = eng.joint_pass()
            else:
                lnl = eng.bottom_up(is_marginal)
        except hip.ZeroLikelihoodError as e:
            failed = np.flatnonzero(e.err_child >= 0)
            if errors is None:
                first = int(failed[0])
                raise LikelihoodError(first, int(e.err_parent[first]), int(e.err_child[first]))
            for c in failed.tolist():
                errors[c] = LikelihoodError(c, int(e.err_parent[c]), int(e.err_child[c]))
            lnl = e.loglik
        if is_marginal and altered.any():
            self.unalter(altered)
        return (lnl, joint.astype(np.int64)) if backtrace else lnl

    def top_down_marginals(self):
        """After a marginal sweep: (posterior [m, N, k], lh_sum [m, N], lh_sf [m, N]) (ml.py:240-290, 431-502)."""
        return self.engine.top_down_marginals()

    def marginal_pass(self, models):
        """
        Bottom-up sweep with the masks as they are (no alteration: the caller has done it, ml.py:700-703) and the
        top-down sweep + marginals + posteriors behind it, one host round trip (``pml_marginal_pass``).
        Returns (ln L [m], posterior [m, N, k], lh_sum [m, N], lh_sf [m, N]).
        """
        eng = self.engine
        self._upload_models(eng, models)
        self._upload_masks()
        eng.set_initial_masks(None)
        self.n_sweeps += self.m
        try:
            return eng.marginal_pass()
        except hip.ZeroLikelihoodError as e:
            first = int(np.flatnonzero(e.err_child >= 0)[0])
            raise LikelihoodError(first, int(e.err_parent[first]), int(e.err_child[first]))

    def joint_states(self):
        """Joint state of every node after a joint sweep (ml.py:598-622): int64 [m, N]."""
        return self.engine.joint_backtrace().astype(np.int64)

    def select(self, method, force_joint=False):
        """
        MAP / MPPA selection (ml.py:505-595) on the device from the posteriors of the last marginal pass; the marginal
        likelihoods of nodes with saved ('.initial') masks are restricted to them first.  The selected masks become both
        the device's and this object's masks.  Returns the number of selected states per node, int64 [m, N].
        """
        lh_masks = None
        if self.has_init.any():
            lh_masks = np.where(self.has_init[..., None], self.init_masks, self.full)
        sel, nsel = self.engine.select_states(method, force_joint=force_joint, lh_masks=lh_masks, packed=True)
        self.masks = sel
        self._uploaded = sel.copy()
        return nsel.astype(np.int64)

    # ------------------------------------------------------------------------------------------------ optimiser columns
    def open_optimiser(self, widths):
        """
        A second context whose columns are blocks, one block of widths[c] columns per character: the points of one
        finite-difference gradient of character c go into block c.  All columns of a block carry the character's masks
        -- as they are for points with tau > 0, altered (pastml/ml.py:101-103) for points with tau == 0.  The
        alteration depends on the masks only, not on the parameters, so it is computed once per character, when its
        first point with tau == 0 comes, for the hundreds of evaluations of an optimisation; the saved '.initial'
        masks stay, as after a sequence of single evaluations.
        """
        offsets = np.concatenate(([0], np.cumsum(widths))).astype(np.int64)
        total = int(offsets[-1])
        eng = hip.acquire_engine(self.flat, total, self.k, device=self._device)
        self._opt = dict(engine=eng, offsets=offsets, total=total, variant=np.full(total, -1, dtype=np.int64),
                         altered={}, models=[None] * total, all_set=False)
        return self._opt

    def _altered_variant(self, c):
        """Masks of character c after alteration, or None if the alteration changes nothing for it."""
        opt = self._opt
        if c not in opt['altered']:
            plain = self.masks[c].copy()
            rows = np.zeros(self.m, dtype=bool)
            rows[c] = True
            changed = self.alter(rows)[c].any()
            opt['altered'][c] = self.masks[c].copy() if changed else None
            self.masks[c] = plain   # a marginal evaluation leaves the masks as they were (ml.py:115-117)
        return opt['altered'][c]

    def evaluate_points(self, requests):
        """
        requests: {character: [(spec, rates), ...]} -- kernel descriptions of the parameter vectors to evaluate, at
        most as many as the character's block is wide.  One bottom-up sweep over all blocks; returns
        {character: ln L array, or a LikelihoodError if one of its points has no likelihood}.
        """
        self.submit_points(requests)
        return self.collect_points()

    def submit_points(self, requests):
        """First half of evaluate_points: parameters and masks to the device, the sweep on the stream; no waiting."""
        opt = self._opt
        eng, offsets, models = opt['engine'], opt['offsets'], opt['models']
        lo, hi = opt['total'], 0
        if opt['all_set'] and opt.get('staged') and all(type(p) is PointBlock for p in requests.values()):
            # the array path (F81 family, every column staged before): a character's points go into the engine's staging
            # arrays as slices, the masks are looked at once per block
            for c, block in requests.items():
                a, n = int(offsets[c]), len(block)
                if n > int(offsets[c + 1]) - a:
                    raise ValueError('{} points for a block of {} columns'.format(n, int(offsets[c + 1]) - a))
                eng.stage_f81(a, block.pi, block.sf, block.tau, block.tf)
                plain = block.tau != 0
                words = None if plain.all() else self._altered_variant(c)
                want = np.zeros(n, dtype=np.int64) if words is None else (~plain).astype(np.int64)
                for j in np.flatnonzero(opt['variant'][a:a + n] != want):
                    eng.set_mask_words(self.masks[c] if want[j] == 0 else words, col_begin=a + int(j))
                    opt['variant'][a + j] = want[j]
                lo, hi = min(lo, a), max(hi, a + n)
            eng.commit_f81(lo, hi)
            self.n_sweeps += sum(len(p) for p in requests.values())
            # only the blocks that asked: every other column keeps the parameters and the results of an earlier sweep
            # (most characters of a group are done long before the last one: a sweep of 246 columns for the six that
            # still matter is the group's whole tail otherwise)
            active = opt.get('active')
            if active is None:
                active = opt['active'] = np.zeros(opt['total'], dtype=np.uint8)
            active[:] = 0
            for c, block in requests.items():
                a = int(offsets[c])
                active[a:a + len(block)] = 1
            eng.bottom_up_submit(True, active=active if SWEEP_ACTIVE_COLUMNS_ONLY else None)
            opt['in_flight'] = requests
            return
        for c, points in requests.items():
            a = int(offsets[c])
            if len(points) > int(offsets[c + 1]) - a:
                raise ValueError('{} points for a block of {} columns'.format(len(points), int(offsets[c + 1]) - a))
            for j, (spec, rates) in enumerate(points):
                models[a + j] = (spec, rates)
                words = self._altered_variant(c) if 0 == rates[1] else None
                variant = 0 if words is None else 1
                if opt['variant'][a + j] != variant:
                    eng.set_mask_words(self.masks[c] if words is None else words, col_begin=a + j)
                    opt['variant'][a + j] = variant
            lo, hi = min(lo, a), max(hi, a + len(points))
        # every column of the context is swept: columns that were never given a model (blocks of characters that have
        # not asked yet, the unused tail of a block) repeat a valid one; their results are not looked at
        if not opt['all_set']:
            filler = next(mm for mm in models if mm is not None)
            for i in range(opt['total']):
                if models[i] is None:
                    models[i] = filler
                if opt['variant'][i] < 0:
                    c = int(np.searchsorted(offsets, i, side='right') - 1)
                    eng.set_mask_words(self.masks[c], col_begin=i)
                    opt['variant'][i] = 0
            eng.set_models(models)
            opt['all_set'] = True
            opt['staged'] = eng.kind == hip.KIND_F81   # (every column's parameters are in the engine's staging arrays now)
        else:
            eng.set_models(models[lo:hi], col_begin=lo)
        self.n_sweeps += sum(len(p) for p in requests.values())
        eng.bottom_up_submit(True)
        opt['in_flight'] = requests

    def collect_points(self):
        """Second half of evaluate_points: waits for the sweep of submit_points and sorts its results by character."""
        opt = self._opt
        eng, offsets = opt['engine'], opt['offsets']
        requests = opt.pop('in_flight')
        failed = None
        try:
            values = eng.bottom_up_collect(True)
        except hip.ZeroLikelihoodError as e:
            values, failed = e.loglik, e
        out = {}
        for c, points in requests.items():
            a = int(offsets[c])
            out[c] = values[a:a + len(points)].copy()
            if failed is not None:
                bad = np.flatnonzero(failed.err_child[a:a + len(points)] >= 0)
                if len(bad):
                    j = a + int(bad[0])
                    out[c] = LikelihoodError(c, int(failed.err_parent[j]), int(failed.err_child[j]))
        return out


# =====================================================================================================================
# lock-step likelihood service for concurrent optimisers
# =====================================================================================================================
class SweepServer(object):
    """
    Rendezvous of the optimiser threads: ``evaluate(c, points)`` blocks until every *running* client has a request
    pending, then one of the waiting threads runs the batched sweep for all and wakes the others.  ``finish(c)`` takes a
    client out of the rendezvous.
    """

    def __init__(self, batch, clients):
        self.batch = batch
        self.cond = threading.Condition()
        self.running = set(clients)
        self.pending = {}
        self.results = {}
        self.rounds = 0

    def _round_if_complete(self):
        # caller holds the condition
        if self.pending and set(self.pending) >= self.running:
            requests, self.pending = self.pending, {}
            try:
                out = self.batch.evaluate_points(requests)
            except Exception as e:  # a failure of the sweep itself reaches every waiting client
                out = {c: e for c in requests}
            self.rounds += 1
            self.results.update(out)
            self.cond.notify_all()

    def evaluate(self, c, points):
        with self.cond:
            self.pending[c] = points
            self._round_if_complete()
            while c not in self.results:
                self.cond.wait()
            res = self.results.pop(c)
        if isinstance(res, Exception):
            raise res
        return res

    def finish(self, c):
        with self.cond:
            self.running.discard(c)
            self._round_if_complete()


# scipy's finite-difference helper (the one L-BFGS-B itself uses when no gradient is given): with it a whole gradient
# is evaluated as one batch and the iterates stay those of scipy's own numerical differentiation.  It is a private
# module: without it the optimisers fall back to letting scipy difference the likelihood point by point (same iterates,
# one sweep per point).
try:
    from scipy.optimize._numdiff import approx_derivative as _approx_derivative
except Exception:  # pragma: no cover - depends on the SciPy build
    _approx_derivative = None
try:
    from scipy.optimize._numdiff import _adjust_scheme_to_bounds
except Exception:  # pragma: no cover - depends on the SciPy build
    _adjust_scheme_to_bounds = None


def two_point_scheme(x0, lower, upper, step=1e-8):
    """
    The points of scipy's forward-difference gradient at x0 -- approx_derivative(method='2-point', abs_step=1e-8,
    bounds=...), the scheme L-BFGS-B applies when it is given no gradient -- and the steps to divide by:
    rows x0 + h_i e_i with the step mirrored where it would leave the bounds, steps recomputed as (x0_i + h_i) - x0_i.
    The gradient is then (f(row_i) - f(x0)) / step_i: scipy's _dense_difference, without calling its function wrapper
    per coordinate (a third of the host time of an optimiser round went there).
    """
    x0 = np.asarray(x0, dtype=np.float64)
    h = np.full(x0.shape, float(step))   # (1e-8: scipy's; the polish of search_parameters_steps takes a larger one)
    zero = ((x0 + h) - x0) == 0
    if zero.any():   # (a step below the spacing of x0: scipy falls back to a relative one)
        sign = (x0 >= 0).astype(float) * 2 - 1
        h = np.where(zero, np.finfo(np.float64).eps ** 0.5 * sign * np.maximum(1.0, np.abs(x0)), h)
    x = x0 + h
    if ((x < lower) | (x > upper)).any():   # (otherwise scipy's helper returns h as it is)
        h, _ = _adjust_scheme_to_bounds(x0, h, 1, '1-sided', lower, upper)
    points = np.repeat(x0[None, :], len(x0), axis=0)
    idx = np.arange(len(x0))
    points[idx, idx] += h
    return points, points[idx, idx] - x0


def batched_gradients_available():
    return _approx_derivative is not None and os.environ.get('PASTML_AMD_BATCHED_OPTIMISER', '1') != '0'


def block_width(model):
    """Columns a character needs in the optimiser context: one per point of its largest finite-difference gradient."""
    if not batched_gradients_available():
        return 1
    fixed = model.extra_params_fixed()
    model.unfix_extra_params()
    n = model.get_num_params()
    if fixed:
        model.fix_extra_params()
    return n + 1


# ---------------------------------------------------------------------------------------------------------------------
# L-BFGS-B by reverse communication.  scipy.optimize.minimize(method='L-BFGS-B') is a Python loop around the routine
# ``setulb``, which returns to its caller whenever it wants the function and gradient at a point
# (scipy/optimize/_lbfgsb_py.py, _minimize_lbfgsb).  Calling ``setulb`` from a *generator* that yields those points gives
# the same iterates -- same routine, same workspace, same options -- without scipy's callback holding the thread: all
# characters of a group then advance in ONE loop (optimise_group), a sweep round per step, no thread per character and
# no hand-offs through the interpreter lock.  The routine is private to scipy; its argument list is checked once and the
# thread-per-character driver (scipy's own minimize) stays as the fallback.
# ---------------------------------------------------------------------------------------------------------------------
def _probe_setulb():
    if os.environ.get('PASTML_AMD_LBFGSB_THREADS'):
        return None
    try:
        from scipy.optimize import _lbfgsb
        doc = (_lbfgsb.setulb.__doc__ or '').replace(' ', '')
        if 'setulb(m,x,l,u,nbd,f,g,factr,pgtol,wa,iwa,task,lsave,isave,dsave,maxls,ln_task)' in doc:
            return _lbfgsb.setulb
    except Exception:  # pragma: no cover - depends on the SciPy build
        pass
    return None


_setulb = _probe_setulb()


def single_loop_optimiser_available():
    return _setulb is not None and batched_gradients_available()


class _Found(object):
    """What the search looks at in scipy's OptimizeResult."""
    __slots__ = ('x', 'fun', 'success', 'nit', 'nfev', 'reason', 'continued_at')

    def __init__(self, x, fun, success, nit, nfev, reason=0, continued_at=None):
        self.x, self.fun, self.success, self.nit, self.nfev, self.reason = x, fun, success, nit, nfev, reason
        self.continued_at = continued_at   # (iteration count, f) where a continued run would have stopped, or None


# The end of a many-parameter search (round 5; HIV1C 'Year', k = 30, was the one column of 91 whose optimum fell short of the
# reference's by more than 1e-6 relative: 0.033 in ln L).  Two things make where such a search ends a matter of luck
# (profiles/r04e_year_optimiser_path.txt, profiles/r05b_year_polish.txt):
#   * scipy's forward differences take a step of 1e-8, and ln L ~ -8e3 is computed to ~2e-10 -- 2e-2 of rounding noise in a
#     slope that goes to zero at the optimum (the reference's own numpy arithmetic has the same);
#   * L-BFGS-B's second convergence test, "RELATIVE REDUCTION OF F <= FACTR*EPSMCH" (task_messages[402] of
#     scipy/optimize/_lbfgsb_py.py), fires when ONE step gains less than 1.8e-5 -- in a flat valley that happens on
#     plateaus well short of the optimum.
# The reference's procedure is kept as it is (same routine, options, differencing, acceptance rule: pastml/ml.py:174-237).
# What is added, for searches of at least CONTINUE_MIN_PARAMETERS free parameters only: the ACCEPTED optimum is polished by
# one more L-BFGS-B run from it whose differences take a step of POLISH_STEP = 1e-6 (noise 2e-4 instead of 2e-2) and which,
# when it ends on the relative-reduction test, is continued once with a tenfold tighter tolerance (the routine keeps its
# tolerance in its workspace: the continued run makes the iterates of a run started with ftol / 10; a FRESH run from the end
# point, without the limited-memory matrix, stalls on the same plateau -- measured).  The better of the two ends is kept: ln L
# can only rise.  Measured on the five HIV1C columns with k >= 30: Year -4.2e-6 -> +1.6e-8 relative to the reference's
# optimum, the others +4e-9 .. +5e-8; the step alone or the continuation alone leave Year at -3.9e-6.  INTEGRATION.md,
# "Differences".  PASTML_AMD_CONTINUE: 0 never continue, 1 continue every run of such a search, 2 (default) its polish only;
# PASTML_AMD_POLISH_STEP=0: no polish.
RELATIVE_REDUCTION = 402
CONTINUE_FTOL_FACTOR = 0.1
CONTINUE = int(os.environ.get('PASTML_AMD_CONTINUE', '2'))
POLISH_STEP = float(os.environ.get('PASTML_AMD_POLISH_STEP', '1e-6'))
CONTINUE_MIN_PARAMETERS = int(os.environ.get('PASTML_AMD_CONTINUE_MIN_PARAMETERS', '20'))


# PASTML_AMD_FD_IN_LIBRARY=0: the finite-difference points of the F81 family through numpy (two_point_scheme + kernel_points)
# instead of the library's helper -- the same numbers; tests compare the two
# the optimiser's sweeps compute the columns of the characters that asked only (Engine.bottom_up_submit(active=...))
SWEEP_ACTIVE_COLUMNS_ONLY = os.environ.get('PASTML_AMD_SWEEP_ACTIVE_ONLY', '1') != '0'
FD_IN_LIBRARY = os.environ.get('PASTML_AMD_FD_IN_LIBRARY', '1') != '0'

# tests / diagnostics: a dict {character: [one record per L-BFGS-B run: start, iterates (x_k, f(x_k)), end, counts]} that
# the searches fill in when it is not None (tests/test_gpu_hiv1c.py compares them with the reference's own runs)
TRACE = None


_WARNED = set()


def _warn_once(text):
    if text not in _WARNED:
        _WARNED.add(text)
        logging.getLogger('pastml').warning(text)


def lbfgsb_steps(x0, bounds, iterates=None, continue_factor=None):
    """
    Generator form of ``minimize(fun, x0, method='L-BFGS-B', bounds=bounds, jac=True)`` with scipy's default options
    (maxcor 10, ftol 2.22e-9, gtol 1e-5, maxfun = maxiter = 15000, maxls 20): yields the points at which it needs
    (f, gradient) and is sent the pair; its return value (StopIteration.value) carries x, fun, success.
    As in scipy, the function is evaluated once at the (clipped) starting point before the routine is entered, and a
    request for the point evaluated last is served from memory (ScalarFunction / MemoizeJac).
    continue_factor: a run that ends on the relative-reduction test goes on, once, with its tolerance multiplied by this.
    """
    m, factr, pgtol, maxfun, maxiter, maxls = 10, 2.220446049250313e-09 / np.finfo(float).eps, 1e-5, 15000, 15000, 20
    bounds = np.asarray(bounds, dtype=np.float64)
    low, up = bounds[:, 0], bounds[:, 1]
    x0 = np.clip(np.asarray(x0, dtype=np.float64).ravel(), low, up)
    n = len(x0)
    nbd = np.zeros(n, np.int32)
    low_bnd = np.zeros(n, np.float64)
    upper_bnd = np.zeros(n, np.float64)
    for i in range(n):
        has_low, has_up = not np.isinf(low[i]), not np.isinf(up[i])
        if has_low:
            low_bnd[i] = low[i]
        if has_up:
            upper_bnd[i] = up[i]
        nbd[i] = 2 if (has_low and has_up) else 1 if has_low else 3 if has_up else 0
    at = np.array(x0, dtype=np.float64)            # the point whose values are in f, g
    f, g = yield at.copy()
    nfev = 1
    g = np.asarray(g, dtype=np.float64)
    x = np.array(x0, dtype=np.float64)
    wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
    iwa = np.zeros(3 * n, dtype=np.int32)
    task = np.zeros(2, dtype=np.int32)
    ln_task = np.zeros(2, dtype=np.int32)
    lsave = np.zeros(4, dtype=np.int32)
    isave = np.zeros(44, dtype=np.int32)
    dsave = np.zeros(29, dtype=np.float64)
    fx, gx = np.array(0.0, dtype=np.int32), np.zeros((n,), dtype=np.int32)   # (scipy enters with these placeholders)
    n_iterations = 0
    continued = False
    while True:
        gx = np.asarray(gx).astype(np.float64)
        _setulb(m, x, low_bnd, upper_bnd, nbd, fx, gx, factr, pgtol, wa, iwa, task, lsave, isave, dsave, maxls, ln_task)
        if task[0] == 3:
            if not np.array_equal(x, at):
                at = np.array(x, dtype=np.float64)
                f, g = yield at.copy()
                g = np.asarray(g, dtype=np.float64)
                nfev += 1
            fx, gx = f, g
        elif task[0] == 1:
            n_iterations += 1
            if iterates is not None:
                iterates.append((np.array(x, dtype=np.float64), float(fx)))
            if n_iterations >= maxiter:
                task[0], task[1] = 5, 504
            elif nfev > maxfun:
                task[0], task[1] = 5, 502
        elif task[0] == 4 and task[1] == RELATIVE_REDUCTION and continue_factor and not continued:
            # the routine computes its tolerance (factr * epsmch) when it starts and keeps it in dsave[2]; entering again with
            # NEW_X repeats the test that just fired, now at the tighter level, and goes on from there.  That is SciPy's private
            # workspace layout (checked with SciPy 1.14.1 and 1.15.3): where the word is not what it should be the run ends
            # here, as the reference's does -- said once, because the end of a many-parameter search then falls a few 1e-6
            # short of what the continuation reaches (profiles/r05b_year_polish.txt)
            if dsave[2] != factr * np.finfo(float).eps:
                _warn_once('L-BFGS-B keeps its tolerance elsewhere in this SciPy build (dsave[2] = {!r}, expected {!r}): '
                           'runs that stop on the relative-reduction test are not continued'
                           .format(float(dsave[2]), factr * np.finfo(float).eps))
                break
            continued = (n_iterations, float(fx))
            dsave[2] *= continue_factor
            task[0], task[1] = 1, 0
        else:
            break
    return _Found(x, fx, task[0] == 4, n_iterations, nfev, int(task[1]), continued or None)


def _drive(steps, evaluate):
    """Runs a generator of likelihood requests to its end against a synchronous evaluator; returns its value."""
    try:
        request = next(steps)
        while True:
            request = steps.send(evaluate(request))
    except StopIteration as stop:
        return stop.value


def search_parameters_steps(model, observed_frequencies, rng, trace=None):
    """
    One L-BFGS-B search over the model's currently free parameters (the procedure of pastml/ml.py:174-237): it starts
    from the current values, then -- if frequencies are free -- from the observed frequencies, then from up to 98
    uniform draws inside the bounds, and stops at the first start whose optimum is at least as good as the better of the
    first two starting likelihoods; if none is, the better starting point is kept.
    Generator: yields lists of parameter vectors, is sent their ln L (one batched sweep each); returns ln L at the
    optimum, where the model is left.
    """
    bounds = model.get_bounds()
    lower, upper = bounds[:, 0], bounds[:, 1]
    lower_c, upper_c = np.ascontiguousarray(lower, dtype=np.float64), np.ascontiguousarray(upper, dtype=np.float64)
    fd_block = getattr(model, 'fd_block', None) if _adjust_scheme_to_bounds is not None and FD_IN_LIBRARY else None

    def negative(values):
        values = np.asarray(values, dtype=np.float64)   # (ln L values or NaN: what pd.isnull would flag)
        return np.where(np.isnan(values), np.inf, -values)

    def objective(ps):
        if np.isnan(np.asarray(ps, dtype=np.float64)).any():
            return np.nan
        values = yield [np.asarray(ps, dtype=np.float64)]
        return negative(values)[0]

    def objective_and_gradient(ps, step=1e-8):
        # the 2-point scheme scipy would apply itself (absolute step 1e-8, steps mirrored at the bounds): a first pass
        # of its own helper records the points it asks for, the batch evaluates them together, a second pass runs on
        # the table of their values -- points and arithmetic are scipy's
        ps = np.asarray(ps, dtype=np.float64)
        if np.isnan(ps).any():
            return np.nan, np.full(len(ps), np.nan)
        if fd_block is not None:
            # F81 family: the points of the gradient, decoded, in one call into the library (host arithmetic; the numbers of
            # two_point_scheme + kernel_points)
            made = fd_block(ps, lower_c, upper_c, step)
            if made is not None:
                block, steps = made
                values = negative((yield block))
                return values[0], (values[1:] - values[0]) / steps
        if _adjust_scheme_to_bounds is not None:
            if ((ps < lower) | (ps > upper)).any():
                raise ValueError("`x0` violates bound constraints.")
            points, steps = two_point_scheme(ps, lower, upper, step)
            values = negative((yield np.vstack((ps[None, :], points))))
            return values[0], (values[1:] - values[0]) / steps
        asked = []
        _approx_derivative(lambda x: asked.append(np.array(x, dtype=np.float64)) or 0.0, ps, method='2-point',
                           abs_step=step, f0=0.0, bounds=(lower, upper))
        values = yield [ps] + asked
        values = negative(values)
        table = {x.tobytes(): v for x, v in zip(asked, values[1:])}
        gradient = _approx_derivative(lambda x: table[np.asarray(x, dtype=np.float64).tobytes()], ps,
                                      method='2-point', abs_step=step, f0=values[0], bounds=(lower, upper))
        return values[0], gradient

    from pastml_amd.models import ModelWithFrequencies
    frequencies_free = isinstance(model, ModelWithFrequencies) and model._optimise_frequencies
    start_current = model.get_optimised_parameters()
    start_observed = start_current
    if frequencies_free:
        model.frequencies = np.maximum(observed_frequencies, 1e-10) if np.any(observed_frequencies <= 0) \
            else observed_frequencies
        start_observed = model.get_optimised_parameters()
    def minimise(x0, step=1e-8):
        # one L-BFGS-B run from x0: (what it found, its trace record or None)
        iterates = [] if trace is not None else None
        search = lbfgsb_steps(x0, bounds, iterates,
                              CONTINUE_FTOL_FACTOR if len(x0) >= CONTINUE_MIN_PARAMETERS
                              and (CONTINUE == 1 or (CONTINUE == 2 and step != 1e-8)) else None)
        try:
            point = next(search)
            while True:
                point = search.send((yield from objective_and_gradient(point, step)))
        except StopIteration as stop:
            found = stop.value
        record = None
        if trace is not None:
            record = dict(x0=np.array(x0, dtype=np.float64), x=np.array(found.x), fun=float(found.fun),
                          success=bool(found.success), nit=found.nit, nfev=found.nfev, reason=found.reason,
                          continued_at=found.continued_at,
                          iterates=np.array([it[0] for it in iterates]), values=np.array([it[1] for it in iterates]))
        return found, record

    lnl_current = -(yield from objective(start_current))
    lnl_observed = -(yield from objective(start_observed)) if frequencies_free else lnl_current
    to_beat = max(lnl_current, lnl_observed)
    for attempt in range(100):
        if attempt == 0:
            x0 = start_current
        elif attempt == 1 and frequencies_free:
            x0 = start_observed
        else:
            x0 = rng.uniform(lower, upper)
        found, record = yield from minimise(x0)
        if trace is not None:
            trace.append(record)
        if found.success and not np.any(np.isnan(found.x)) and -found.fun >= to_beat:
            if POLISH_STEP and len(found.x) >= CONTINUE_MIN_PARAMETERS:
                polished, record2 = yield from minimise(found.x, POLISH_STEP)
                if record is not None:
                    record['polish'] = record2
                if not np.any(np.isnan(polished.x)) and polished.fun < found.fun:
                    found = polished
            model.set_params_from_optimised(found.x)
            return -found.fun
    model.set_params_from_optimised(start_current if lnl_current >= lnl_observed else start_observed)
    return to_beat


def search_parameters(model, observed_frequencies, evaluate, rng):
    """
    Synchronous form of search_parameters_steps: ``evaluate(list of parameter vectors)`` -> list of ln L.  Without the
    reverse-communication routine (other SciPy builds) the search runs through scipy's own minimize.
    """
    if single_loop_optimiser_available():
        return _drive(search_parameters_steps(model, observed_frequencies, rng), evaluate)
    return _search_parameters_scipy(model, observed_frequencies, evaluate, rng)


def _search_parameters_scipy(model, observed_frequencies, evaluate, rng):
    """The search of search_parameters_steps with scipy.optimize.minimize as the inner loop (fallback)."""
    from scipy.optimize import minimize
    from pastml_amd.models import ModelWithFrequencies
    bounds = model.get_bounds()
    lower, upper = bounds[:, 0], bounds[:, 1]

    def negative(values):
        return [np.inf if pd.isnull(v) else -v for v in values]

    def objective(ps):
        if np.any(pd.isnull(ps)):
            return np.nan
        return negative(evaluate([np.asarray(ps, dtype=np.float64)]))[0]

    def objective_and_gradient(ps, step=1e-8):
        ps = np.asarray(ps, dtype=np.float64)
        if np.any(pd.isnull(ps)):
            return np.nan, np.full(len(ps), np.nan)
        asked = []
        _approx_derivative(lambda x: asked.append(np.array(x, dtype=np.float64)) or 0.0, ps, method='2-point',
                           abs_step=step, f0=0.0, bounds=(lower, upper))
        values = negative(evaluate([ps] + asked))
        table = {x.tobytes(): v for x, v in zip(asked, values[1:])}
        gradient = _approx_derivative(lambda x: table[np.asarray(x, dtype=np.float64).tobytes()], ps,
                                      method='2-point', abs_step=step, f0=values[0], bounds=(lower, upper))
        return values[0], gradient

    frequencies_free = isinstance(model, ModelWithFrequencies) and model._optimise_frequencies
    start_current = model.get_optimised_parameters()
    start_observed = start_current
    if frequencies_free:
        model.frequencies = np.maximum(observed_frequencies, 1e-10) if np.any(observed_frequencies <= 0) \
            else observed_frequencies
        start_observed = model.get_optimised_parameters()
    lnl_current = -objective(start_current)
    lnl_observed = -objective(start_observed) if frequencies_free else lnl_current
    to_beat = max(lnl_current, lnl_observed)
    batched = batched_gradients_available()
    for attempt in range(100):
        if attempt == 0:
            x0 = start_current
        elif attempt == 1 and frequencies_free:
            x0 = start_observed
        else:
            x0 = rng.uniform(lower, upper)
        def minimise(x0, step=1e-8):
            # one L-BFGS-B run from x0 -- the same rule as search_parameters_steps.minimise: a run of a many-parameter search
            # that stops on the relative-reduction test is continued (CONTINUE = 1: every run, 2: the polish run only), in
            # the only form scipy's own driver offers: the run again from its start with the tighter ftol -- the same
            # iterates up to where the first run stopped, then the ones the continuation makes
            def run(options=None):
                if batched:
                    return minimize(lambda ps: objective_and_gradient(ps, step), x0=x0, method='L-BFGS-B', bounds=bounds,
                                    jac=True, options=options)
                return minimize(objective, x0=x0, method='L-BFGS-B', bounds=bounds,
                                options=dict(options or {}, eps=step))
            found = run()
            goes_on = len(x0) >= CONTINUE_MIN_PARAMETERS and (CONTINUE == 1 or (CONTINUE == 2 and step != 1e-8))
            if goes_on and found.success and 'RELATIVE REDUCTION OF F' in str(found.message):
                found = run(dict(ftol=2.220446049250313e-09 * CONTINUE_FTOL_FACTOR))
            return found
        found = minimise(x0)
        if found.success and not np.any(np.isnan(found.x)) and -found.fun >= to_beat:
            if POLISH_STEP and len(found.x) >= CONTINUE_MIN_PARAMETERS:   # (the polish of search_parameters_steps)
                polished = minimise(found.x, POLISH_STEP)
                if not np.any(np.isnan(polished.x)) and polished.fun < found.fun:
                    found = polished
            model.set_params_from_optimised(found.x)
            return -found.fun
    model.set_params_from_optimised(start_current if lnl_current >= lnl_observed else start_observed)
    return to_beat


def fit_parameters_steps(character, model, observed_frequencies, rng, search=None):
    """
    Likelihood at the given parameters, then -- if anything is free -- the scaling / smoothing factors alone, then all
    free parameters together (the two stages of pastml/ml.py:865-920).  Generator: yields lists of
    (kernel description, rate parameters), is sent their ln L; returns ln L at the optimum.
    ``search``: a synchronous replacement for the inner search (the fallback driver), given the point evaluator.
    """
    from pastml_amd.ml import PastMLLikelihoodError
    logger = logging.getLogger('pastml')
    trouble = 'Failed to {} the likelihood for your tree, please check that you do not have contradicting {} ' \
              'states specified for internal tree nodes, ' \
              'and if not - submit a bug at https://github.com/evolbioinfo/pastml/issues'

    def report(title, text, lnl):
        logger.debug('{} for {}:\n{}{}'.format(title, character, text, '\tlog likelihood:\t{:.6f}'.format(lnl)))

    def points_of(vectors):
        return vectors if type(vectors) is PointBlock else model.kernel_points(vectors)

    def run_search():
        if search is not None:
            return search(points_of)
        steps = search_parameters_steps(model, observed_frequencies, rng,
                                        None if TRACE is None else TRACE.setdefault(character, []))
        try:
            vectors = next(steps)
            while True:
                vectors = steps.send((yield points_of(vectors)))
        except StopIteration as stop:
            return stop.value

    lnl = float((yield [(model.kernel_spec(), model.rate_params())])[0])
    if np.isnan(lnl):
        raise PastMLLikelihoodError(trouble.format('calculate', character))
    if not model.get_num_params():
        report('All the parameters are fixed', model._print_parameters(), lnl)
        return lnl
    report('Initial values for parameter optimisation', model._print_parameters(), lnl)
    if not model.basic_params_fixed():
        model.fix_extra_params()
        try:
            lnl = yield from run_search()
        finally:
            model.unfix_extra_params()
        if np.isnan(lnl) or lnl == -np.inf:
            raise PastMLLikelihoodError(trouble.format('optimise', character))
        if not model.extra_params_fixed():
            report('Pre-optimised basic parameters', model._print_basic_parameters(), lnl)
    if not model.extra_params_fixed():
        lnl = yield from run_search()
        if np.isnan(lnl) or lnl == -np.inf:
            raise PastMLLikelihoodError(trouble.format('calculate', character))
    report('Optimised parameters', model._print_parameters(), lnl)
    return lnl


def fit_parameters(character, model, observed_frequencies, evaluate, rng):
    """Synchronous form of fit_parameters_steps: ``evaluate(list of (kernel description, rates))`` -> ln L array."""
    search = None
    if not single_loop_optimiser_available():
        def search(points_of):
            return _search_parameters_scipy(model, observed_frequencies, lambda vectors: evaluate(points_of(vectors)), rng)
    return _drive(fit_parameters_steps(character, model, observed_frequencies, rng, search), evaluate)


# =====================================================================================================================
# ml_acr for a group of characters
# =====================================================================================================================
class Task(object):
    """One character of a run: what pastml.ml.ml_acr takes as arguments."""

    def __init__(self, character, method, model, observed_frequencies):
        self.character, self.method, self.model = character, method, model
        self.observed_frequencies = observed_frequencies

    @property
    def group_key(self):
        return len(self.model.states), self.model.kernel_spec()['kind'], self.method


def likelihood_error(flat, e):
    """The reference's message for a zero likelihood (pastml/ml.py:139-145) from the node ids the device reports."""
    from pastml_amd.ml import PastMLLikelihoodError
    name = (lambda i: flat.nodes[i].name) if flat.nodes is not None else str
    return PastMLLikelihoodError("The parent node {} and its child node {} have non-intersecting states, "
                                 "and are connected by a zero-length ({:g}) branch. "
                                 "This creates a zero likelihood value. "
                                 "To avoid this issue check the restrictions on these node states "
                                 "and/or use a smoothing factor (tau).".format(name(e.parent), name(e.child),
                                                                               flat.dist[e.child]))


class GroupSearch(object):
    """
    The parameter searches of one group of characters (one CharacterBatch): every character's fit_parameters_steps
    generator, the requests they are waiting on, and the two halves of a sweep round -- ``submit`` sends the pending
    requests of all characters to the device in one batched sweep without waiting, ``collect`` waits for it and steps
    every character to its next request.
    """

    def __init__(self, batch, tasks, seeds):
        self.batch, self.tasks = batch, tasks
        m = len(tasks)
        self.lnl = np.full(m, np.nan)
        self.errors = {}
        self.pending = {}
        self.rounds = 0
        batch.open_optimiser([block_width(t.model) for t in tasks])
        self.searches = [fit_parameters_steps(t.character, t.model, t.observed_frequencies,
                                              np.random.RandomState(seeds[c])) for c, t in enumerate(tasks)]
        for c in range(m):
            self._advance(c, lambda: next(self.searches[c]))

    def _advance(self, c, step):
        # runs character c's search up to its next request (or its end / failure)
        try:
            self.pending[c] = step()
        except StopIteration as stop:
            self.lnl[c] = stop.value
        except BaseException as e:  # delivered to the caller after all searches are done
            self.errors[c] = e

    def submit(self):
        self._failed = None
        try:
            self.batch.submit_points(self.pending)
        except Exception as e:  # a failure of the sweep itself reaches every character that asked
            self._failed = e

    def collect(self):
        requests, self.pending = self.pending, {}
        if self._failed is None:
            try:
                out = self.batch.collect_points()
            except Exception as e:
                self._failed = e
        if self._failed is not None:
            out = {c: self._failed for c in requests}
        self.rounds += 1
        for c, res in out.items():
            if isinstance(res, Exception):
                self._advance(c, lambda: self.searches[c].throw(res))
            else:
                self._advance(c, lambda: self.searches[c].send(res))

    def close(self):
        hip.release_engine(self.batch._opt['engine'])
        self.batch._opt = None

    def error(self):
        if not self.errors:
            return None
        e = self.errors[min(self.errors)]
        return likelihood_error(self.batch.flat, e) if isinstance(e, LikelihoodError) else e


def optimise_groups(groups):
    """
    ONE loop for all searches of all groups, each group on its own context (= stream).  Every group's first sweep is
    submitted before any is waited for; from then on a group's next sweep goes out as soon as its last one has been
    collected and its characters stepped to their next requests -- while the loop does that for one group the sweeps of
    all the others are in flight, so the device is not idle during the host's share of a round (round 3; before, every
    round submitted all groups, then collected all: the device waited while the points of the next round were made).
    A group's sequence of sweeps is what it would be alone.  No interpreter threads, so the groups do not take the
    interpreter lock from each other (with a thread per group an acr() over the 91 HIV1C columns took as long as its
    groups one after the other).
    """
    live = [g for g in groups if g.pending]
    for g in live:
        g.submit()
    while live:
        still = []
        for g in live:
            g.collect()
            if g.pending:
                g.submit()
                still.append(g)
        live = still
    for g in groups:
        g.close()


def optimise_group(batch, tasks, seeds=None):
    """
    Parameters of every character of the batch, all optimisers advancing together: ONE loop steps every character's
    search (fit_parameters_steps) to its next likelihood request, a single batched sweep serves them all, and so on until
    the last search has ended -- a run costs the rounds of its slowest character, and no interpreter thread per
    character.  Returns (ln L [m], sweep rounds).
    """
    m = len(tasks)
    # restart points come from per-character generators seeded, in character order, from numpy's global one: the
    # optimisers advance together, a shared generator would hand its draws out in an order that depends on the batch
    if seeds is None:
        seeds = np.random.randint(0, 2 ** 31 - 1, size=m)
    if not single_loop_optimiser_available():
        return _optimise_group_threads(batch, tasks, seeds)
    group = GroupSearch(batch, tasks, seeds)
    optimise_groups([group])
    if group.error() is not None:
        raise group.error()
    return group.lnl, group.rounds


def _optimise_group_threads(batch, tasks, seeds):
    """Fallback driver (scipy's own minimize, which holds its thread): one thread per character, met in SweepServer."""
    m = len(tasks)
    batch.open_optimiser([block_width(t.model) for t in tasks])
    server = SweepServer(batch, range(m))
    lnl = np.full(m, np.nan)
    errors = {}

    def work(c):
        t = tasks[c]
        try:
            lnl[c] = fit_parameters(t.character, t.model, t.observed_frequencies,
                                    lambda points: server.evaluate(c, points), np.random.RandomState(seeds[c]))
        except BaseException as e:  # delivered to the caller after all optimisers are done
            errors[c] = e
        finally:
            server.finish(c)

    if m == 1:
        work(0)
    else:
        threads = [threading.Thread(target=work, args=(c,), name='pastml-opt-{}'.format(c)) for c in range(m)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    hip.release_engine(batch._opt['engine'])
    batch._opt = None
    if errors:
        e = errors[min(errors)]
        raise likelihood_error(batch.flat, e) if isinstance(e, LikelihoodError) else e
    return lnl, server.rounds


def count_scenarios(sizes):
    """
    Product of the numbers of states kept per node (ml.py:567-568), as an exact integer.  As powers per distinct size:
    the node-by-node product of the reference multiplies an ever longer integer N times (seconds at 10^5 nodes).
    """
    values, counts = np.unique(np.asarray(sizes)[np.asarray(sizes) > 1], return_counts=True)
    scenarios = 1
    for value, count in zip(values.tolist(), counts.tolist()):
        scenarios *= value ** count
    return scenarios


def reconstruct(batch, tasks, lnl, force_joint=True):
    """
    Everything of pastml/ml.py:640-750 after the parameters are known, for all characters of the batch at once (they
    share the prediction method): joint sweep + back-trace, marginal pass, MAP and MPPA selections with their
    restricted likelihoods; results and node features as the reference leaves them.
    Returns one list of result dictionaries per character.
    """
    from pastml_amd import ml
    from pastml_amd import get_personalized_feature_name as feature_name, CHARACTER, METHOD, NUM_SCENARIOS, \
        NUM_UNRESOLVED_NODES, NUM_STATES_PER_NODE, PERC_UNRESOLVED, STATES
    logger = logging.getLogger('pastml')
    flat, m, k = batch.flat, batch.m, batch.k
    method = tasks[0].method
    models = [t.model for t in tasks]
    current = [{ml.LOG_LIKELIHOOD: float(lnl[c]), CHARACTER: t.character, METHOD: method, ml.MODEL: t.model,
                STATES: t.model.states} for c, t in enumerate(tasks)]
    results = [[] for _ in tasks]

    def emit(which):
        # the selected states become the character's node feature; a copy of the result as it stands is reported
        if which != method and not ml.is_meta_ml(method):
            return
        for c, t in enumerate(tasks):
            name = t.character if which == method else feature_name(t.character, which)
            flat.set_column(name, StateSetColumn(batch.masks[c].copy(), t.model.states))
            res = current[c].copy()
            res[CHARACTER], res[METHOD] = name, which
            results[c].append(res)

    def note_restricted(which, values):
        for c, t in enumerate(tasks):
            logger.debug('Log likelihood for {} after {} state selection:\t{:.6f}'.format(t.character, which, values[c]))
            current[c][ml.RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(which)] = float(values[c])

    try:
        if method != ml.MAP:
            lnl_joint, joint = batch.joint_pass(models)
            note_restricted(ml.JOINT, lnl_joint)
            batch.masks = one_hot_words(joint, k)
            for c, t in enumerate(tasks):
                flat.set_column(feature_name(t.character, ml.JOINT_STATE), ArrayColumn(joint[c], convert=int))
            emit(ml.JOINT)

        if ml.is_marginal(method):
            batch.initialize_allowed_states()
            altered = batch.alter(np.array([0 == mdl.tau for mdl in models], dtype=bool))
            _, posterior, lh_sum, lh_sf = batch.marginal_pass(models)
            # (row order and row names: ONE index object for the tables of all characters of the group -- an index of 7 237
            # names built per table was a sixth of the time outside the optimiser on the HIV1C columns)
            order = None if len(flat.roots) == 1 else \
                np.lexsort((np.arange(flat.n_nodes), flat.tree_id))   # tree by tree, level order each (ml.py:498-502)
            ids = range(flat.n_nodes) if order is None else order
            names = pd.Index([flat.nodes[i].name for i in ids] if flat.nodes is not None else list(ids))
            for c, t in enumerate(tasks):
                current[c][ml.MARGINAL_PROBABILITIES] = pd.DataFrame(posterior[c] if order is None else posterior[c][order],
                                                                     index=names, columns=t.model.states)
            batch.unalter(altered)
            lh = posterior * lh_sum[:, :, None]

            def restrict_saved():
                # marginal likelihoods of nodes that were ever altered count only inside their own saved masks
                # (ml.py:541-542, 593-594)
                if batch.has_init.any():
                    lh[batch.has_init] *= masks_from_words(batch.init_masks[batch.has_init], k)

            restrict_saved()
            batch.select('MAP')
            note_restricted(ml.MAP, batch.bottom_up(models, is_marginal=True, alter=True))
            emit(ml.MAP)

            if method == ml.ALL:
                # the parsimonious reconstructions of the annotation, and the likelihood restricted to each of them
                # (ml.py:718-733); a selection that leaves some zero-length branch without a common state has none
                from pastml_amd.parsimony import parsimonious_acr, MP
                mp_results = [parsimonious_acr(batch.flat.nodes[:len(flat.roots)], t.character, MP, t.model.states,
                                               t.model.forest_stats.num_nodes, t.model.forest_stats.num_tips)
                              for t in tasks]
                for c in range(m):
                    results[c].extend(mp_results[c])
                for which in range(len(mp_results[0])):
                    for c, t in enumerate(tasks):
                        batch.masks[c] = flat.columns[mp_results[c][which][CHARACTER]].words
                    failures = {}
                    values = batch.bottom_up(models, is_marginal=True, alter=True, errors=failures)
                    name = mp_results[0][which][METHOD]
                    for c, t in enumerate(tasks):
                        if c in failures:
                            logger.error('{}\n{} parsimonious state selection is inconsistent in terms of ML.'
                                         .format(likelihood_error(flat, failures[c]).message, name))
                        else:
                            current[c][ml.RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(name)] = float(values[c])

            if method == ml.MPPA or ml.is_meta_ml(method):
                restrict_saved()   # the restricted-MAP sweep may have saved new masks (ml.py:675-680 before :541)
                kept = batch.select('MPPA', force_joint=force_joint)
                for c, t in enumerate(tasks):
                    n_nodes = t.model.forest_stats.num_nodes
                    scenarios = count_scenarios(kept[c])
                    unresolved = int((kept[c] > 1).sum())
                    per_node = int(kept[c].sum()) / n_nodes
                    current[c].update({NUM_SCENARIOS: scenarios, NUM_UNRESOLVED_NODES: unresolved,
                                       NUM_STATES_PER_NODE: per_node, PERC_UNRESOLVED: unresolved * 100 / n_nodes})
                    logger.debug('{} node{} unresolved ({:.2f}%) for {} by {}, i.e. {:.4f} state{} per node in average.'
                                 .format(unresolved, 's are' if unresolved != 1 else ' is', unresolved * 100 / n_nodes,
                                         t.character, ml.MPPA, per_node, 's' if per_node > 1 else ''))
                note_restricted(ml.MPPA, batch.bottom_up(models, is_marginal=True, alter=True))
                emit(ml.MPPA)

            for c, t in enumerate(tasks):
                flat.set_column(feature_name(t.character, ml.LH), ArrayColumn(lh[c]))
                flat.set_column(feature_name(t.character, ml.LH_SF), ArrayColumn(lh_sf[c], convert=float))
    except LikelihoodError as e:
        raise likelihood_error(flat, e)

    for c, t in enumerate(tasks):
        flat.set_column(feature_name(t.character, ml.ALLOWED_STATES), MaskColumn(batch.masks[c].copy(), k))
    return results


# bytes of device memory per character and node, generously: bottom-up vector, posterior, arg-max rows, scalars of the
# character's own column + what each of the (on average) `widths` columns of its optimiser block holds.  Eigen models beyond 64
# states read P(t) of every branch from HBM, k x k doubles per node and column (Limits): the joint sweep of the
# character's own column always (65 - 128 states: the sum sweeps are fused, so the optimiser's columns need none), every column
# beyond 128 states.
def _column_bytes(flat, k, widths, kind=hip.KIND_F81):
    ks = k + (k & 1)
    own, point = 17 * ks + 96, 8 * ks + 64
    if kind == hip.KIND_EIGEN and k > 64:
        own += 8 * k * ks
        if k > 128:
            point += 8 * k * ks
    return flat.n_nodes * (own + sum(widths) / max(1, len(widths)) * point)


def visible_devices(device=None):
    """
    The GPUs one process spreads its groups of characters over.  A process that is one rank of a multi-process launch
    (RANK / LOCAL_RANK / PASTML_HIP_DEVICE set: one process per GPU, pastml_amd.sharding) keeps to its own device; a
    plain call -- what pastml.acr.acr() with threads is in the reference (acr.py:226-231) -- takes every visible
    device.  PASTML_AMD_DEVICES="0,1,..." picks them explicitly (a device may be named twice: tests on one GPU).
    """
    if device is not None:
        return [int(device)]
    env = os.environ.get('PASTML_AMD_DEVICES')
    if env:
        return [int(x) for x in env.split(',') if x.strip() != '']
    if any(v in os.environ for v in ('LOCAL_RANK', 'PASTML_HIP_DEVICE')) or int(os.environ.get('WORLD_SIZE', 1)) > 1:
        return [hip.default_device()]
    return list(range(max(1, hip.device_count())))


def run_tasks(forest, tasks, force_joint=True, device=None, flat=None, seeds=None):
    """
    ml_acr for a list of Tasks on one forest.  Characters are grouped by (number of states, model family, prediction
    method); a group becomes one CharacterBatch -- or several, if the device memory does not hold all of its columns
    at once, or if there are several GPUs to spread a large group over (visible_devices).  Results do not depend on
    the placement: a column's bits depend on nothing but the column.  Returns one list of result dictionaries per
    task, in task order.
    """
    if isinstance(forest, TreeNode):
        forest = [forest]
    if flat is None:
        flat = get_flat_forest(forest)
    groups = {}
    for i, t in enumerate(tasks):
        groups.setdefault(t.group_key, []).append(i)
    out = [None] * len(tasks)
    stats = dict(groups=0, rounds=0, sweeps=0)
    devices = visible_devices(device)
    free = None
    for dev in sorted(set(devices)):
        with hip.BareContext(dev) as probe:
            _, f = probe.memory()
        free = f if free is None else min(free, f)
    if os.environ.get('PASTML_AMD_DEVICE_BYTES'):   # plan as if the device had this much free memory (tests)
        free = min(free, int(float(os.environ['PASTML_AMD_DEVICE_BYTES'])))
    # a group is split over the devices when it is worth a second context (nodes x characters; small forests are bound
    # by the host loop, which does not get shorter)
    split_min = float(os.environ.get('PASTML_AMD_SPLIT_MIN_WORK', 2e5))
    jobs, job_bytes = [], []
    for key, members in groups.items():
        k = key[0]
        per_char = _column_bytes(flat, k, [block_width(tasks[i].model) for i in members], kind=key[1])
        chunk = max(1, min(len(members), 4096, int(0.6 * free / max(1.0, per_char))))
        if len(devices) > 1 and len(members) >= 2 and flat.n_nodes * len(members) >= split_min:
            chunk = min(chunk, -(-len(members) // len(devices)))
        for a in range(0, len(members), chunk):
            jobs.append((k, members[a:a + chunk]))
            job_bytes.append(per_char * len(members[a:a + chunk]))
    # placement: largest job first, each on the device that holds the least so far (deterministic)
    load = [0.0] * len(devices)
    job_device = [devices[0]] * len(jobs)
    for j in sorted(range(len(jobs)), key=lambda q: (-job_bytes[q], q)):
        d = min(range(len(devices)), key=lambda q: (load[q], q))
        job_device[j] = devices[d]
        load[d] += job_bytes[j]
    jobs = [(k, part, job_device[j]) for j, (k, part) in enumerate(jobs)]
    total_bytes = max(load) if load else 0.0
    stats['devices'] = sorted(set(job_device)) if jobs else []

    # restart seeds of all characters, in task order (the groups advance together); acr() hands in the ones it drew for
    # the whole call when the characters are shared out over several processes
    if seeds is None:
        seeds = np.random.randint(0, 2 ** 31 - 1, size=len(tasks))
    seeds = np.asarray(seeds)

    per_group = []   # (diagnostics) where the time of a run goes: one entry per group of characters

    import time

    def prepare(job):
        k, part, dev = job
        group = [tasks[i] for i in part]
        batch = CharacterBatch(flat, k, len(group), device=dev)
        for c, t in enumerate(group):
            batch.set_annotation(c, *annotation_words(flat, t.character, t.model.states))
        batch.initialize_allowed_states()
        return batch, group

    def run(job):
        k, part = job[:2]
        t0 = time.perf_counter()
        batch, group = prepare(job)
        with batch:
            lnl, rounds = optimise_group(batch, group, seeds[part])
            t1 = time.perf_counter()
            res = reconstruct(batch, group, lnl, force_joint=force_joint)
            per_group.append(dict(k=k, characters=len(group), rounds=rounds, sweeps=batch.n_sweeps,
                                  optimise_s=round(t1 - t0, 3), reconstruct_s=round(time.perf_counter() - t1, 3)))
            return part, res, rounds, batch.n_sweeps

    # Groups are independent: when all of them fit the device together their searches advance in ONE loop
    # (optimise_groups), each group on its own contexts (= streams).  On small trees a sweep is a chain of latency-bound
    # launches, so the sweeps of different groups overlap on the GPU and with the host work of the other groups, and a
    # run costs about the host work of all groups, not the sum of their sweep latencies on top.
    concurrent = len(jobs) > 1 and total_bytes < 0.3 * free and os.environ.get('PASTML_AMD_CONCURRENT_GROUPS', '1') != '0' \
        and single_loop_optimiser_available()
    done = []
    if concurrent:
        t0 = time.perf_counter()
        prepared, searches = [], []
        try:
            for job in jobs:
                prepared.append(prepare(job))
            searches = [GroupSearch(batch, group, seeds[job[1]]) for job, (batch, group) in zip(jobs, prepared)]
            optimise_groups(searches)
            t1 = time.perf_counter()
            failures = [(job[1][0], g.error()) for job, g in zip(jobs, searches) if g.error() is not None]
            if failures:
                raise min(failures, key=lambda f: f[0])[1]
            for job, (batch, group), g in zip(jobs, prepared, searches):
                t2 = time.perf_counter()
                res = reconstruct(batch, group, g.lnl, force_joint=force_joint)
                per_group.append(dict(k=job[0], characters=len(group), rounds=g.rounds, sweeps=batch.n_sweeps,
                                      optimise_s=None, reconstruct_s=round(time.perf_counter() - t2, 3)))
                done.append((job[1], res, g.rounds, batch.n_sweeps))
            stats['optimise_all_groups_s'] = round(t1 - t0, 3)
        finally:
            for batch, _ in prepared:
                batch.__exit__(None, None, None)
    else:
        for job in jobs:
            done.append(run(job))
    for part, res, rounds, sweeps in done:
        stats['groups'] += 1
        stats['rounds'] += rounds
        stats['sweeps'] += sweeps
        for i, r in zip(part, res):
            out[i] = r
    stats['per_group'] = per_group
    run_tasks.last_stats = stats
    return out
