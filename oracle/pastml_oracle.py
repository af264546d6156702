"""
ORACLE -- test infrastructure, not product code.

CPU restatement (numpy, one node at a time) of the maximum-likelihood ACR hot path of evolbioinfo/pastml, written to
follow the reference's operation sequence so that it reproduces its numbers to ~1e-13 and its per-node cost profile:

    P(t)                    pastml/models/F81Model.py:28-46, HKYModel.py:44-82, generator.py:33-65,
                            branch transform models/__init__.py:39-42,269-270
    bottom-up sweep         pastml/ml.py:82-148 (calc_node_bu_likelihood), rescale_log ml.py:151-171
    top-down sweep          pastml/ml.py:240-290
    marginals / posteriors  pastml/ml.py:431-502
    joint back-trace        pastml/ml.py:598-622
    MAP / MPPA selection    pastml/ml.py:505-595

It works on plain arrays (parent / first_child / n_children / dist in forest-wide level order, see
pastml_amd/tree.py) and on a model description dict, so it does not depend on the product's host logic or on the HIP
library.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import it.

Parity status: PINNED -- checked in tests/test_oracle_golden.py against fixtures produced by the real reference
(imported from /root/reference with tests/golden/make_golden.py) and against the reference's own pinned test values
(tests/ACRParameterOptimisationMPPA{F81,JC,EFT}Test.py) through those fixtures.
"""
import numpy as np

MIN_VALUE = np.log10(np.finfo(np.float64).eps)
MAX_VALUE = np.log10(np.finfo(np.float64).max)

KIND_F81, KIND_HKY, KIND_EIGEN = 0, 1, 2


class OracleLikelihoodError(Exception):
    """Zero likelihood at (parent, child): the condition of pastml/ml.py:139-145."""

    def __init__(self, parent, child):
        Exception.__init__(self, 'zero likelihood at parent {} child {}'.format(parent, child))
        self.parent = parent
        self.child = child


# ---------------------------------------------------------------------------------------------------------------------
# P(t)
# ---------------------------------------------------------------------------------------------------------------------

def transform_t(t, sf, tau, tau_factor):
    """models/__init__.py:269-270."""
    return (t + tau) * tau_factor * sf


def tau_factor(tau, forest_length, num_nodes):
    """models/__init__.py:39-42."""
    return forest_length / (forest_length + tau * (num_nodes - 1)) if tau else 1


def normalised_generator(pi, rates=None):
    """generator.py:33-51."""
    k = len(pi)
    if rates is None:
        rates = np.ones((k, k)) - np.eye(k)
    q = rates * pi
    q -= np.diag(q.sum(axis=1))
    return q / (-q.diagonal().dot(pi))


def diagonalise(pi, rates=None):
    """generator.py:16-30: numpy eig + inv."""
    d, a = np.linalg.eig(normalised_generator(pi, rates))
    return d, a, np.linalg.inv(a)


def pij(spec, t, sf=1., tau=0., tf=1.):
    """
    k x k transition matrix for branch length t.
    spec: dict(kind, pi, [mu | kappa | d, A, Ainv]).
    """
    tt = transform_t(t, sf, tau, tf)
    pi = spec['pi']
    kind = spec['kind']
    if kind == KIND_F81:
        # F81Model.py:42-46
        with np.errstate(divide='ignore'):
            mu = 1. / (1. - pi.dot(pi))
        e = 0. if mu == np.inf else np.exp(-mu * tt)
        return (1 - e) * pi + np.eye(len(pi)) * e
    if kind == KIND_HKY:
        # HKYModel.py:55-82, states A C G T
        kappa = spec['kappa']
        a, c, g, tfreq = pi
        ag, ct = a + g, c + tfreq
        beta = .5 / (ag * ct + kappa * (a * g + c * tfreq))
        eb = np.exp(-beta * tt)
        e_ct = np.exp(-beta * tt * (1. + ct * (kappa - 1.))) / ct
        e_ag = np.exp(-beta * tt * (1. + ag * (kappa - 1.))) / ag
        s_ct = (ct + ag * eb) / ct
        s_ag = (ag + ct * eb) / ag
        p = np.ones((4, 4)) * (1 - eb)
        p *= pi
        p[3, 3] = tfreq * s_ct + c * e_ct
        p[3, 1] = c * s_ct - c * e_ct
        p[1, 3] = tfreq * s_ct - tfreq * e_ct
        p[1, 1] = c * s_ct + tfreq * e_ct
        p[0, 0] = a * s_ag + g * e_ag
        p[0, 2] = g * s_ag - g * e_ag
        p[2, 0] = a * s_ag - a * e_ag
        p[2, 2] = g * s_ag + a * e_ag
        return p
    if kind == KIND_EIGEN:
        # generator.py:54-65
        return spec['A'].dot(np.diag(np.exp(spec['d'] * tt))).dot(spec['Ainv'])
    raise ValueError('unknown model kind {}'.format(kind))


# ---------------------------------------------------------------------------------------------------------------------
# scaling
# ---------------------------------------------------------------------------------------------------------------------

def rescale_log(log_arr):
    """
    ml.py:151-171: shifts the (finite part of the) log10 array into the representable band, in place;
    returns the shift.
    """
    finite = log_arr[log_arr > -np.inf]
    lo = np.min(finite)
    hi = np.max(finite)
    shift = 0
    if hi > MAX_VALUE:
        shift = MAX_VALUE - hi - 1
    elif lo < MIN_VALUE:
        shift = min(MIN_VALUE - lo + 1, MAX_VALUE - hi - 1)
    log_arr += shift
    return shift


# ---------------------------------------------------------------------------------------------------------------------
# traversal helpers on the flat arrays
# ---------------------------------------------------------------------------------------------------------------------

def postorder(parent, first_child, n_children, roots):
    """Node ids tree by tree, each tree in post-order with children left to right (ete3 'postorder')."""
    out = []
    for r in roots:
        stack = [(int(r), False)]
        while stack:
            n, expanded = stack.pop()
            if expanded or n_children[n] == 0:
                out.append(n)
            else:
                stack.append((n, True))
                fc = first_child[n]
                for c in range(fc + n_children[n] - 1, fc - 1, -1):
                    stack.append((c, False))
    return out


def preorder(parent, first_child, n_children, roots):
    out = []
    for r in roots:
        stack = [int(r)]
        while stack:
            n = stack.pop()
            out.append(n)
            fc = first_child[n]
            stack.extend(range(fc + n_children[n] - 1, fc - 1, -1))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# sweeps
# ---------------------------------------------------------------------------------------------------------------------

def bottom_up(tree, masks, spec, sf=1., tau=0., tf=1., is_marginal=True):
    """
    Felsenstein pruning (ml.py:82-148).

    tree: object with parent, first_child, n_children, dist, roots (e.g. pastml_amd.tree.FlatForest)
    masks: int array [N, k] of allowed states
    Returns dict(bu[N,k], bu_sf[N], loglik (sum over trees), loglik_per_tree, joint_table[N,k] (joint only)).
    Raises OracleLikelihoodError like ml.py:139-145.
    """
    parent, first_child, n_children, dist = tree.parent, tree.first_child, tree.n_children, tree.dist
    N, k = masks.shape
    pi = spec['pi']
    bu = np.zeros((N, k))
    bu_sf = np.zeros(N)
    table = None if is_marginal else np.zeros((N, k), dtype=np.int64)
    with np.errstate(divide='ignore', invalid='ignore'):
        for n in postorder(parent, first_child, n_children, tree.roots):
            log_arr = np.log10(np.ones(k, dtype=np.float64) * masks[n])
            factors = 0
            fc = first_child[n]
            for c in range(fc, fc + n_children[n]):
                cl = pij(spec, dist[c], sf, tau, tf) * bu[c]
                if is_marginal:
                    cl = cl.sum(axis=1)
                else:
                    table[c] = cl.argmax(axis=1)
                    cl = cl.max(axis=1)
                cl = np.maximum(cl, 0)
                log_arr += np.log10(cl)
                if np.all(log_arr == -np.inf):
                    raise OracleLikelihoodError(n, c)
                factors += rescale_log(log_arr)
            bu[n] = np.power(10, log_arr)
            bu_sf[n] = factors + sum(bu_sf[fc + j] for j in range(n_children[n]))
        per_tree = []
        for r in tree.roots:
            rl = bu[r] * pi
            rl = rl.sum() if is_marginal else rl.max()
            per_tree.append(np.log(rl) - bu_sf[r] / np.log10(np.e))
    res = dict(bu=bu, bu_sf=bu_sf, loglik=sum(per_tree), loglik_per_tree=np.array(per_tree))
    if table is not None:
        res['joint_table'] = table
    return res


def top_down(tree, bu, bu_sf, spec, sf=1., tau=0., tf=1.):
    """ml.py:240-290. Returns td[N,k], td_sf[N]."""
    parent, first_child, n_children, dist = tree.parent, tree.first_child, tree.n_children, tree.dist
    N, k = bu.shape
    td = np.zeros((N, k))
    td_sf = np.zeros(N)
    with np.errstate(divide='ignore', invalid='ignore'):
        for n in preorder(parent, first_child, n_children, tree.roots):
            p = parent[n]
            if p < 0:
                td[n] = np.ones(k, np.float64)
                td_sf[n] = 0
                continue
            pt = np.transpose(pij(spec, dist[n], sf, tau, tf))
            contrib = bu[n].dot(pt)
            contrib[contrib <= 0] = 1
            plog = np.log10(td[p]) + np.log10(bu[p]) - np.log10(contrib)
            factors = td_sf[p] + bu_sf[p] - bu_sf[n]
            factors += rescale_log(plog)
            td[n] = np.maximum(np.power(10, plog).dot(pt), 0)
            td_sf[n] = factors
    return td, td_sf


def marginals(bu, bu_sf, td, td_sf, masks, pi):
    """ml.py:454-460, 498-500. Returns lh[N,k], lh_sf[N], posterior[N,k]."""
    N, k = bu.shape
    lh = np.zeros((N, k))
    lh_sf = np.zeros(N)
    with np.errstate(divide='ignore', invalid='ignore'):
        for n in range(N):
            ll = np.log10(bu[n]) + np.log10(td[n]) + np.log10(pi * masks[n])
            f = rescale_log(ll)
            lh[n] = np.power(10, ll)
            lh_sf[n] = f + td_sf[n] + bu_sf[n]
    return lh, lh_sf, lh / lh.sum(axis=1)[:, None]


def full_marginal_pass(tree, masks, spec, sf=1., tau=0., tf=1.):
    """BU + TD + marginals: one 'step' of the benchmark metric."""
    r = bottom_up(tree, masks, spec, sf, tau, tf, True)
    td, td_sf = top_down(tree, r['bu'], r['bu_sf'], spec, sf, tau, tf)
    lh, lh_sf, post = marginals(r['bu'], r['bu_sf'], td, td_sf, masks, spec['pi'])
    r.update(td=td, td_sf=td_sf, lh=lh, lh_sf=lh_sf, posterior=post)
    return r


def joint_backtrace(tree, bu, table, pi):
    """ml.py:598-622: root state = argmax(BU_root * pi); child state = table[child][parent state]."""
    N = len(bu)
    state = np.zeros(N, dtype=np.int64)
    for n in preorder(tree.parent, tree.first_child, tree.n_children, tree.roots):
        p = tree.parent[n]
        state[n] = (bu[n] * pi).argmax() if p < 0 else table[n][state[p]]
    return state


def unalter_joint_table(table, masks_initial, altered):
    """ml.py:408-428 on the argmax tables of the altered nodes (in place)."""
    for n in altered:
        init = masks_initial[n]
        allowed_index = np.argmax(init)
        if len(init[init > 0]) == 1:
            table[n] = np.ones(len(init), int) * allowed_index
        else:
            for i in range(len(init)):
                if not init[table[n][i]]:
                    table[n][i] = allowed_index


# ---------------------------------------------------------------------------------------------------------------------
# state selection
# ---------------------------------------------------------------------------------------------------------------------

def choose_map(lh):
    """ml.py:591-595 (lh already multiplied by the initial mask where needed)."""
    return lh.argmax(axis=1)


def choose_mppa(lh, joint_state=None):
    """
    ml.py:539-572 per node. lh: [N,k] marginal likelihoods (already multiplied by the initial masks where needed);
    joint_state: [N] or None (force_joint off).  Returns (selected masks [N,k] int, best_k [N]).
    """
    N, k = lh.shape
    out = np.zeros((N, k), dtype=int)
    best_ks = np.zeros(N, dtype=int)
    for n in range(N):
        ml = lh[n]
        probs = ml / ml.sum()
        if joint_state is not None:
            ji = joint_state[n]
            probs = np.hstack((np.sort(np.delete(probs, ji)), [probs[ji]]))
        else:
            probs = np.sort(probs)
        best_k, best_c = k, np.inf
        for m in range(1, k + 1):
            corr = np.hstack((np.zeros(k - m), np.ones(m) / m)) - probs
            corr = corr.dot(corr)
            if corr < best_c:
                best_c, best_k = corr, m
        # the reference's sort key has a constant first component (ml.py:562-563), i.e. stable sort by -lh
        sel = sorted(range(k), key=lambda _: -ml[_])[:best_k]
        out[n, sel] = 1
        best_ks[n] = best_k
    return out, best_ks
