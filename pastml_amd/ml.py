"""
Maximum-likelihood ancestral character reconstruction: the host side of the hot path of ``pastml/ml.py``.

Same public functions, arguments, result dictionaries, node features and errors as the reference
(``ml_acr`` ml.py:640, ``get_bottom_up_loglikelihood`` :82, ``calculate_top_down_likelihood`` :240,
``calculate_marginal_likelihoods`` :431, ``convert_likelihoods_to_probabilities`` :486, ``optimise_likelihood`` :865,
``optimize_likelihood_params`` :174, the three ``choose_ancestral_states_*`` :505-622), but the tree is flattened once
into level-ordered arrays (:class:`ForestProblem`) and every likelihood sweep runs on the GPU through
``pastml_amd.hip.Engine`` (C-ABI ``include/pastml_hip.h``).  There is no CPU implementation of the sweeps in this
module: without the HIP library / an MI355X the calls raise.

What stays on the host, as SURVEY.md section 8a prescribes: the allowed-state masks and their zero-branch alteration
(ml.py:293-428), the scipy L-BFGS-B driver (:174-237), the per-node state selection rules (:505-595) and the
bookkeeping of results.
"""
import numpy as np
import pandas as pd

from pastml_amd import get_personalized_feature_name
from pastml_amd import hip
from pastml_amd.tree import TreeNode, get_flat_forest

LOG_LIKELIHOOD = 'log_likelihood'
RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR = '{}_restricted_{{}}'.format(LOG_LIKELIHOOD)

JOINT = 'JOINT'
MPPA = 'MPPA'
MAP = 'MAP'
ALL = 'ALL'
ML = 'ML'

MARGINAL_PROBABILITIES = 'marginal_probabilities'

MODEL = 'model'

MIN_VALUE = np.log10(np.finfo(np.float64).eps)
MAX_VALUE = np.log10(np.finfo(np.float64).max)

MARGINAL_ML_METHODS = {MPPA, MAP}
ML_METHODS = MARGINAL_ML_METHODS | {JOINT}
META_ML_METHODS = {ML, ALL}

BU_LH = 'BOTTOM_UP_LIKELIHOOD'
TD_LH = 'TOP_DOWN_LIKELIHOOD'
LH = 'LIKELIHOOD'
LH_SF = 'LIKELIHOOD_SF'
BU_LH_SF = 'BOTTOM_UP_LIKELIHOOD_SF'
BU_LH_JOINT_STATES = 'BOTTOM_UP_LIKELIHOOD_JOINT_STATES'
TD_LH_SF = 'TOP_DOWM_LIKELIHOOD_SF'
ALLOWED_STATES = 'ALLOWED_STATES'
STATE_COUNTS = 'STATE_COUNTS'
JOINT_STATE = 'JOINT_STATE'


def is_marginal(method):
    """MAP, MPPA or one of the meta-methods (ALL, ML)."""
    return method in MARGINAL_ML_METHODS or method in META_ML_METHODS


def is_ml(method):
    """JOINT, a marginal method or a meta-method."""
    return method in ML_METHODS or method in META_ML_METHODS


def is_meta_ml(method):
    return method in META_ML_METHODS


def get_default_ml_method():
    return MPPA


class PastMLLikelihoodError(Exception):

    def __init__(self, *args):
        self.message = args[0] if args else None

    def __str__(self):
        if self.message:
            return 'PastMLLikelihoodError, {}'.format(self.message)
        return 'PastMLLikelihoodError has been raised.'


# =====================================================================================================================
# array-level problem: one character on one forest
# =====================================================================================================================

def zero_branch_tops(flat):
    """
    top[n] = the highest node reachable from n through zero-length branches (the representative of n's
    zero-distance cluster, ml.py:321-349).
    """
    from pastml_amd.batch import zero_clusters
    return zero_clusters(flat).top


class ForestProblem(object):
    """
    One character on one forest: a single-column view of :class:`pastml_amd.batch.CharacterBatch` (which holds the
    allowed-state masks of the reference's ``<character>_ALLOWED_STATES`` features as packed words, the masks saved by
    zero-branch alteration, and the device context) with the masks exposed as 0/1 arrays [N, k].  The stand-alone
    sweep functions of this module and ``marginal_counts`` work on it; ``ml_acr`` uses the batch directly.
    """

    def __init__(self, forest, character, states, flat=None, device=None):
        from pastml_amd.batch import CharacterBatch
        if isinstance(forest, TreeNode):
            forest = [forest]
        self.forest = forest
        self.character = character
        self.states = np.asarray(states)
        self.k = len(self.states)
        self.flat = flat if flat is not None else get_flat_forest(forest)
        self.nodes = self.flat.nodes
        self.N = self.flat.n_nodes
        self.batch = CharacterBatch(self.flat, self.k, 1, device=device)

    # the reference's 0/1 arrays, converted on access
    @property
    def masks(self):
        return hip.unpack_masks(self.batch.masks[0], self.k)

    @masks.setter
    def masks(self, value):
        self.batch.masks[0] = hip.pack_masks(value, self.k)

    @property
    def init_masks(self):
        return hip.unpack_masks(self.batch.init_masks[0], self.k)

    @property
    def has_init(self):
        return self.batch.has_init[0]

    @property
    def annotated(self):
        return self.batch.annotated[0]

    @annotated.setter
    def annotated(self, value):
        self.batch.annotated[0] = value

    @property
    def n_sweeps(self):
        return self.batch.n_sweeps

    @property
    def engine(self):
        return self.batch.engine

    def close(self):
        self.batch.close()

    # ------------------------------------------------------------------------------------------------ masks
    def initialize_allowed_states(self):
        """Masks from the ``character`` feature of the nodes (ml.py:293-318), and which nodes "have a state"."""
        from pastml_amd.batch import annotation_words
        self.batch.set_annotation(0, *annotation_words(self.flat, self.character, self.states))
        self.batch.initialize_allowed_states()

    def alter_zero_node_allowed_states(self):
        """Zero-branch alteration (ml.py:352-387); returns the ids of the altered nodes."""
        return np.flatnonzero(self.batch.alter(np.ones(1, dtype=bool))[0])

    def unalter_zero_node_allowed_states(self, altered):
        """masks & saved masks, or the saved ones if nothing is left (ml.py:390-405)."""
        flags = np.zeros((1, self.N), dtype=bool)
        flags[0, np.asarray(altered, dtype=np.int64)] = True
        self.batch.unalter(flags)

    # ------------------------------------------------------------------------------------------------ device
    def bottom_up_loglikelihood(self, model, is_marginal=True, alter=True):
        """Sum over the trees of the forest of get_bottom_up_loglikelihood (ml.py:82-121)."""
        from pastml_amd.batch import LikelihoodError, likelihood_error
        try:
            return float(self.batch.bottom_up([model], is_marginal=is_marginal, alter=alter)[0])
        except LikelihoodError as e:
            raise likelihood_error(self.flat, e)

    def select_on_device(self, method, force_joint=False):
        """MAP / MPPA selection (ml.py:505-595) on the device; returns the number of selected states per node."""
        return self.batch.select(method, force_joint=force_joint)[0]

    def joint_states(self):
        """Joint state of every node after a joint sweep (ml.py:598-622)."""
        return self.batch.joint_states()[0]

    def top_down_marginals(self):
        """
        After a marginal sweep: top-down sweep, marginal likelihoods and posteriors (ml.py:240-290, 431-502).
        Returns (posterior [N, k], lh [N, k], lh_sf [N]) where lh / lh_sf play the role of the reference's
        LIKELIHOOD / LIKELIHOOD_SF features: log10(lh.sum()) - lh_sf is the node's total log10-likelihood.
        """
        post, lh_sum, lh_sf = self.batch.top_down_marginals()
        return post[0], post[0] * lh_sum[0][:, None], lh_sf[0]

    def per_tree_order(self):
        """Node ids tree by tree, each in level order: the row order of the reference's probability table."""
        if len(self.flat.roots) == 1:
            return np.arange(self.N)
        return np.lexsort((np.arange(self.N), self.flat.tree_id))


# =====================================================================================================================
# state selection on arrays
# =====================================================================================================================

def select_map(lh):
    """One-hot masks at the arg-max of the marginal likelihoods (ml.py:591-595)."""
    N, k = lh.shape
    masks = np.zeros((N, k), dtype=np.int8)
    masks[np.arange(N), lh.argmax(axis=1)] = 1
    return masks


def select_mppa(lh, joint_state=None, chunk=1 << 16):
    """
    Marginal posterior probabilities approximation (ml.py:539-572) for all nodes at once.
    lh: [N, k] marginal likelihoods (already multiplied by the saved masks where the reference does so);
    joint_state: [N] to force the joint state into the selection (force_joint), or None.
    Returns (masks [N, k] int8, best_k [N]).

    Per node: the probabilities are sorted ascending (the joint state's one is moved last), the number m of kept
    states minimises sum((0,..,0,1/m,..,1/m) - sorted)^2 (first minimum), and the m states with the largest
    likelihoods are kept (stable order, i.e. ties go to the lower index -- the reference's sort key has a constant
    first component, ml.py:562-563).
    """
    N, k = lh.shape
    masks = np.zeros((N, k), dtype=np.int8)
    best_ks = np.zeros(N, dtype=np.int64)
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        ml = lh[a:b]
        probs = ml / ml.sum(axis=1)[:, None]
        if joint_state is not None:
            ji = joint_state[a:b]
            rows = np.arange(b - a)
            jp = probs[rows, ji]
            rest = probs.copy()
            rest[rows, ji] = np.inf  # sorts last, then replaced by the joint probability
            q = np.sort(rest, axis=1)
            q[:, -1] = jp
        else:
            q = np.sort(probs, axis=1)
        best_c = np.full(b - a, np.inf)
        best_k = np.full(b - a, k, dtype=np.int64)
        for m in range(1, k + 1):
            u = np.hstack((np.zeros(k - m), np.ones(m) / m))
            corr = u[None, :] - q
            corr = (corr * corr).sum(axis=1)
            better = corr < best_c
            best_c[better] = corr[better]
            best_k[better] = m
        order = np.argsort(-ml, axis=1, kind='stable')
        keep = np.arange(k)[None, :] < best_k[:, None]
        sel = np.zeros((b - a, k), dtype=np.int8)
        np.put_along_axis(sel, order, keep.astype(np.int8), axis=1)
        masks[a:b] = sel
        best_ks[a:b] = best_k
    return masks, best_ks


# =====================================================================================================================
# reference-shaped functions on trees
# =====================================================================================================================

def _problem_of(tree, character, model):
    """The ForestProblem cached on a tree for the stand-alone API functions."""
    cache = tree.__dict__.setdefault('_pastml_amd_problems', {})
    flat = get_flat_forest([tree])
    problem = cache.get(character)
    if problem is None or problem.flat is not flat or problem.k != len(model.states):
        if problem is not None:
            problem.close()
        problem = ForestProblem([tree], character, model.states, flat=flat)
        cache[character] = problem
    return problem


def initialize_allowed_states(tree, feature, states):
    """Adds the ``<feature>_ALLOWED_STATES`` arrays to the nodes (ml.py:293-318)."""
    allowed_states_feature = get_personalized_feature_name(feature, ALLOWED_STATES)
    n = len(states)
    state2index = dict(zip(states, range(n)))
    for node in tree.traverse():
        node_states = getattr(node, feature, set())
        if not node_states:
            allowed_states = np.ones(n, dtype=int)
        else:
            allowed_states = np.zeros(n, dtype=int)
            for state in node_states:
                allowed_states[state2index[state]] = 1
        node.add_feature(allowed_states_feature, allowed_states)


def _pull_masks_from_features(problem, tree, character):
    """Reads the node features into the problem's arrays (stand-alone API path)."""
    A = get_personalized_feature_name(character, ALLOWED_STATES)
    init = A + '.initial'
    N, k = problem.N, problem.k
    masks = np.zeros((N, k), dtype=np.int8)
    saved = np.zeros((N, k), dtype=np.int8)
    for i, node in enumerate(problem.nodes):
        masks[i] = getattr(node, A)
        value = getattr(node, character, None)
        problem.batch.annotated[0, i] = value is not None and value != ''
        if hasattr(node, init):
            problem.batch.has_init[0, i] = True
            saved[i] = getattr(node, init)
    problem.masks = masks
    problem.batch.init_masks[0] = hip.pack_masks(saved, k)


def _push_masks_to_features(problem, character):
    A = get_personalized_feature_name(character, ALLOWED_STATES)
    init = A + '.initial'
    masks, saved, has = problem.masks, problem.init_masks, problem.has_init
    for i, node in enumerate(problem.nodes):
        node.add_feature(A, masks[i].astype(int))
        if has[i]:
            node.add_feature(init, saved[i].astype(int))


def get_bottom_up_loglikelihood(tree, character, model, is_marginal=True, alter=True):
    """
    Bottom-up log-likelihood of one tree for the masks stored in the ``<character>_ALLOWED_STATES`` node features
    (API of ml.py:82-121).  The per-node vectors stay on the device; ``<character>_BOTTOM_UP_LIKELIHOOD`` and
    ``..._SF`` features are written back so that callers that read them keep working.
    """
    problem = _problem_of(tree, character, model)
    _pull_masks_from_features(problem, tree, character)
    res = problem.bottom_up_loglikelihood(model, is_marginal=is_marginal, alter=alter)
    _push_masks_to_features(problem, character)
    bu = problem.engine.download(hip.BUF_BU)
    bu_sf = problem.engine.download(hip.BUF_BU_SF)
    lh_feature = get_personalized_feature_name(character, BU_LH)
    lh_sf_feature = get_personalized_feature_name(character, BU_LH_SF)
    for i, node in enumerate(problem.nodes):
        node.add_feature(lh_feature, bu[i])
        node.add_feature(lh_sf_feature, bu_sf[i])
    if not is_marginal:
        table = problem.engine.download(hip.BUF_JOINT_TABLE).astype(np.int64)
        f = get_personalized_feature_name(character, BU_LH_JOINT_STATES)
        for i, node in enumerate(problem.nodes):
            if not node.is_root():
                node.add_feature(f, table[i])
    return res


def calculate_top_down_likelihood(tree, character, model):
    """
    Top-down likelihoods (API of ml.py:240-270); must follow a marginal get_bottom_up_loglikelihood on the same tree.
    The device computes the top-down vectors, the marginal likelihoods and the posteriors in one fused sweep.
    """
    problem = _problem_of(tree, character, model)
    post, lh, lh_sf = problem.top_down_marginals()
    problem.last_marginals = (post, lh, lh_sf)
    td = problem.engine.download(hip.BUF_TD)
    td_sf = problem.engine.download(hip.BUF_TD_SF)
    f, fsf = get_personalized_feature_name(character, TD_LH), get_personalized_feature_name(character, TD_LH_SF)
    for i, node in enumerate(problem.nodes):
        node.add_feature(f, td[i])
        node.add_feature(fsf, td_sf[i])


def calculate_marginal_likelihoods(tree, feature, frequencies, clean_up=True):
    """
    Stores ``<feature>_LIKELIHOOD`` / ``_LIKELIHOOD_SF`` on the nodes (API of ml.py:431-465) from the fused device
    sweep run by calculate_top_down_likelihood.
    """
    problem = tree.__dict__.get('_pastml_amd_problems', {}).get(feature)
    if problem is None or getattr(problem, 'last_marginals', None) is None:
        raise ValueError('calculate_top_down_likelihood must be called first')
    post, lh, lh_sf = problem.last_marginals
    lh_feature = get_personalized_feature_name(feature, LH)
    lh_sf_feature = get_personalized_feature_name(feature, LH_SF)
    for i, node in enumerate(problem.nodes):
        node.add_feature(lh_feature, lh[i].copy())
        node.add_feature(lh_sf_feature, lh_sf[i])
        if clean_up:
            for name in (BU_LH, BU_LH_SF, TD_LH, TD_LH_SF):
                node.del_feature(get_personalized_feature_name(feature, name))


def convert_likelihoods_to_probabilities(tree, feature, states):
    """DataFrame node name -> marginal probabilities, rows in tree.traverse() order (ml.py:486-502)."""
    lh_feature = get_personalized_feature_name(feature, LH)
    names, rows = [], []
    for node in tree.traverse():
        lh = getattr(node, lh_feature)
        names.append(node.name)
        rows.append(lh / lh.sum())
    return pd.DataFrame(np.array(rows), index=names, columns=states)


def convert_allowed_states2feature(tree, feature, states, out_feature=None):
    if out_feature is None:
        out_feature = feature
    allowed_states_feature = get_personalized_feature_name(feature, ALLOWED_STATES)
    for node in tree.traverse():
        node.add_feature(out_feature, set(states[getattr(node, allowed_states_feature).astype(bool)]))


# =====================================================================================================================
# parameter optimisation (API of ml.py:174-237, 865-920; the procedure itself lives in pastml_amd.batch)
# =====================================================================================================================

def _single_character_evaluator(problem):
    """evaluate(points) for the optimiser of one character: a bottom-up sweep with one column per point."""
    from pastml_amd.batch import LikelihoodError, likelihood_error
    batch = problem.batch

    def evaluate(points):
        out = batch.evaluate_points({0: points})[0]
        if isinstance(out, LikelihoodError):
            raise likelihood_error(problem.flat, out)
        return out
    return evaluate


def optimize_likelihood_params(forest, character, observed_frequencies, model, problem=None):
    """
    L-BFGS-B over the model's currently free parameters (API of ml.py:174-237); every function evaluation -- every
    whole finite-difference gradient -- is one bottom-up sweep on the GPU.  The model is left at the optimum.
    """
    from pastml_amd.batch import search_parameters, block_width
    own = problem is None
    if own:
        problem = ForestProblem(forest, character, model.states)
        problem.initialize_allowed_states()
    try:
        opened = problem.batch._opt is None
        if opened:
            problem.batch.open_optimiser([block_width(model)])
        evaluate = _single_character_evaluator(problem)

        def evaluate_vectors(vectors):
            points = []
            for ps in vectors:
                model.set_params_from_optimised(ps)
                points.append((model.kernel_spec(), model.rate_params()))
            return evaluate(points)
        try:
            return search_parameters(model, observed_frequencies, evaluate_vectors,
                                     np.random.RandomState(np.random.randint(0, 2 ** 31 - 1)))
        finally:
            if opened:
                hip.release_engine(problem.batch._opt['engine'])
                problem.batch._opt = None
    finally:
        if own:
            problem.close()


def optimise_likelihood(forest, character, model, observed_frequencies, problem=None):
    """Initial likelihood, then the scaling / smoothing factors, then all free parameters (API of ml.py:865-920)."""
    from pastml_amd.batch import fit_parameters, block_width
    own = problem is None
    if own:
        problem = ForestProblem(forest, character, model.states)
    try:
        problem.initialize_allowed_states()
        problem.batch.open_optimiser([block_width(model)])
        try:
            return fit_parameters(character, model, observed_frequencies, _single_character_evaluator(problem),
                                  np.random.RandomState(np.random.randint(0, 2 ** 31 - 1)))
        finally:
            hip.release_engine(problem.batch._opt['engine'])
            problem.batch._opt = None
    finally:
        if own:
            problem.close()


# =====================================================================================================================
# ml_acr
# =====================================================================================================================

def ml_acr(forest, character, prediction_method, model, observed_frequencies, force_joint=True):
    """
    ML states on the trees, stored in node features; returns the list of result dictionaries (ml.py:640-750).

    Node features written (as the reference leaves them, here as columnar features of the flattened forest):
    ``<character>`` (set of selected states), ``<character>_ALLOWED_STATES``, ``<character>_JOINT_STATE`` (unless
    MAP), ``<character>_LIKELIHOOD`` / ``_LIKELIHOOD_SF`` (marginal methods).  One character is a batch of one:
    ``pastml_amd.acr.acr`` hands all characters of a run to :func:`pastml_amd.batch.run_tasks` together.
    """
    from pastml_amd.batch import Task, run_tasks
    return run_tasks(forest, [Task(character, prediction_method, model, observed_frequencies)],
                     force_joint=force_joint)[0]


# =====================================================================================================================
# marginal_counts
# =====================================================================================================================

def marginal_counts(forest, character, model, n_repetitions=1_000, device_sampling=True):
    """
    Expected numbers of state changes i -> j along the trees, estimated by drawing ``n_repetitions`` ancestral
    scenarios from the marginal posterior (API and sampling scheme of pastml/ml.py:753-862, used by
    utilities/transition_counter.py).

    The likelihood part -- bottom-up and top-down sweeps, root posteriors, per-branch P(t) -- runs on the GPU, and so
    does the scenario sampling (draws of child states given the parent's state counts, ml.py:818-857:
    ``pml_marginal_counts``).  Where zero-branch handling altered some nodes the draws are still the device's
    (``pml_marginal_counts_altered``); the fractional counts the reference gives the (parent, child) pairs with an altered end
    (ml.py:806-812, 840-853) are formed here from the nodes' state counts.  More than 256 states, or
    ``device_sampling=False``: the host sampler with numpy's generator, line by line the reference's.  Either way only
    statistical parity with the reference is meaningful.

    :param device_sampling: False forces the host sampler (tests compare the two)
    :return: k x k array, entry [i, j] = average number of i -> j changes per scenario
    """
    if isinstance(forest, TreeNode):
        forest = [forest]
    problem = ForestProblem(forest, character, model.states)
    try:
        k = problem.k
        flat = problem.flat
        problem.initialize_allowed_states()
        altered = problem.alter_zero_node_allowed_states() if 0 == model.tau else np.zeros(0, dtype=np.int64)
        problem.bottom_up_loglikelihood(model, is_marginal=True, alter=False)
        posterior, _, _ = problem.top_down_marginals()
        if device_sampling and not len(altered) and k <= 256:   # (the device sampler's tables hold 256 states)
            # no node altered by the zero-branch handling: the scenarios are drawn on the device (same scheme, a
            # counter-based generator seeded from numpy's global one, so np.random.seed still fixes the result)
            return problem.engine.marginal_counts(n_repetitions, seed=int(np.random.randint(0, 2 ** 62, dtype=np.int64)))
        if device_sampling and k <= 256:
            # altered nodes: the draws are the device's all the same; the pairs with an altered end get their fractional counts
            # here (ml.py:806-812, 840-853), from the nodes' state counts, and their parents the diagonal correction (:857-858)
            is_altered = np.zeros(problem.N, dtype=bool)
            is_altered[altered] = True
            sums, counts, same_int = problem.engine.marginal_counts_altered(
                n_repetitions, int(np.random.randint(0, 2 ** 62, dtype=np.int64)), is_altered)
            initial = problem.init_masks

            def to_initial(cnt, n):
                c = cnt * initial[n]
                if np.count_nonzero(c):
                    return n_repetitions * c / c.sum()
                return n_repetitions * initial[n] / initial[n].sum()

            result = sums
            internal = np.flatnonzero(flat.n_children > 0)
            fcs, ncs = flat.first_child, flat.n_children
            has_altered_child = np.zeros(problem.N, dtype=bool)
            has_altered_child[flat.parent[altered][flat.parent[altered] >= 0]] = True
            for parent in internal[is_altered[internal] | has_altered_child[internal]]:
                pcounts = counts[parent].astype(np.float64)
                ps = to_initial(pcounts, parent) if is_altered[parent] else pcounts
                same = same_int[parent].astype(np.float64)
                pos = ps > 0
                for node in range(fcs[parent], fcs[parent] + ncs[parent]):
                    if not (is_altered[parent] or is_altered[node]):
                        continue
                    ccounts = counts[node].astype(np.float64)
                    ci = to_initial(ccounts, node) if is_altered[node] else ccounts
                    norm = ci / ci.sum()
                    result[pos] += ps[pos, None] * norm[None, :]
                    same[pos] += ps[pos] * norm[pos]
                result[np.arange(k), np.arange(k)] -= np.minimum(ps, same)
            return result / n_repetitions
        bu = problem.engine.download(hip.BUF_BU)
        # (an eigen model's P(0) carries +-1e-17 where the exact value is 0: clamped, as the device sampler and the sweeps do)
        P = np.maximum(problem.engine.pij_batch(copy_out=True)[0], 0.0)
        frequencies = np.asarray(model.frequencies, dtype=np.float64)
        is_altered = np.zeros(problem.N, dtype=bool)
        is_altered[altered] = True
        masks = problem.masks
        initial = problem.init_masks

        def restrict_to_initial(counts, n):
            """ml.py:806-812 / 840-846: counts of an altered node projected on its own (unaltered) states."""
            c = counts * initial[n]
            if np.count_nonzero(c):
                return n_repetitions * c / c.sum()
            return n_repetitions * initial[n] / initial[n].sum()

        result = np.zeros((k, k), dtype=float)
        state_counts = np.zeros((problem.N, k), dtype=np.int64)
        for parent in range(problem.N):
            if flat.parent[parent] < 0:
                state_counts[parent] = np.random.multinomial(n_repetitions, posterior[parent] / posterior[parent].sum())
            nc = flat.n_children[parent]
            if nc == 0:
                continue
            parent_counts = state_counts[parent]
            ps_counts_initial = restrict_to_initial(parent_counts, parent) if is_altered[parent] else parent_counts
            same_state_counts = np.zeros(k)
            fc = flat.first_child[parent]
            for node in range(fc, fc + nc):
                # p(child = b | parent = a)  ~  BU_node[b] * P[b, a] * pi_b * mask_b   (ml.py:819-824)
                weights = (bu[node] * frequencies * masks[node])[None, :] * P[node].T
                with np.errstate(divide='ignore', invalid='ignore'):  # rows of impossible parent states are unused
                    probs = weights / weights.sum(axis=1)[:, None]
                update_results = not is_altered[parent] and not is_altered[node]
                counts = np.zeros(k, dtype=np.int64)
                for j in np.flatnonzero(parent_counts):
                    drawn = np.random.multinomial(parent_counts[j], probs[j])
                    counts += drawn
                    if update_results:
                        result[j, :] += drawn
                        same_state_counts[j] += drawn[j]
                if not update_results:
                    counts_initial = restrict_to_initial(counts, node) if is_altered[node] else counts
                    norm_counts = counts_initial / counts_initial.sum()
                    for i in np.flatnonzero(ps_counts_initial > 0):
                        adjusted = norm_counts * ps_counts_initial[i]
                        result[i, :] += adjusted
                        same_state_counts[i] += adjusted[i]
                state_counts[node] = counts
            for i in range(k):
                result[i, i] -= min(ps_counts_initial[i], same_state_counts[i])
        return result / n_repetitions
    finally:
        problem.close()
