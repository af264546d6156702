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
import logging
import os

import numpy as np
import pandas as pd
from scipy.optimize import minimize
from scipy.optimize._numdiff import approx_derivative

from pastml_amd import get_personalized_feature_name, CHARACTER, METHOD, NUM_SCENARIOS, NUM_UNRESOLVED_NODES, \
    NUM_STATES_PER_NODE, PERC_UNRESOLVED, STATES
from pastml_amd import hip
from pastml_amd.models import ModelWithFrequencies
from pastml_amd.tree import TreeNode, FlatForest, get_flat_forest

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
    zero-distance cluster, ml.py:321-349).  Parents precede children in id order.
    """
    top = np.arange(flat.n_nodes, dtype=np.int64)
    zero = (flat.dist == 0) & (flat.parent >= 0)
    for lvl in range(1, flat.n_td_levels):
        a, b = flat.td_offsets[lvl], flat.td_offsets[lvl + 1]
        ids = np.arange(a, b)[zero[a:b]]
        top[ids] = top[flat.parent[ids]]
    return top


class ForestProblem(object):
    """
    Host mirror of what the reference keeps in node features for one character: allowed-state masks
    (``<character>_ALLOWED_STATES``), the masks saved by zero-branch alteration (``...ALLOWED_STATES.initial``),
    plus the device engine holding the flattened forest.
    """

    def __init__(self, forest, character, states, flat=None, device=None):
        if isinstance(forest, TreeNode):
            forest = [forest]
        self.forest = forest
        self.character = character
        self.states = np.asarray(states)
        self.k = len(self.states)
        self.flat = flat if flat is not None else get_flat_forest(forest)
        self.nodes = self.flat.nodes
        self.N = self.flat.n_nodes
        self._device = device
        self._engine = None
        self.masks = np.ones((self.N, self.k), dtype=np.int8)
        self.init_masks = np.zeros((self.N, self.k), dtype=np.int8)
        self.has_init = np.zeros(self.N, dtype=bool)
        self.annotated = np.zeros(self.N, dtype=bool)
        self._top = None
        self._uploaded_masks = None
        self._uploaded_model = None
        self.n_sweeps = 0

    @property
    def engine(self):
        """The device context, created on first use (mask bookkeeping alone needs no GPU)."""
        if self._engine is None:
            self._engine = hip.acquire_engine(self.flat, 1, self.k, device=self._device)
        return self._engine

    def close(self):
        if self._engine is not None:
            hip.release_engine(self._engine)
            self._engine = None
        for slot in self.__dict__.pop('_batch_engines', {}).values():
            hip.release_engine(slot['engine'])

    # ------------------------------------------------------------------------------------------------ masks
    def initialize_allowed_states(self):
        """
        Masks from the ``character`` feature of the nodes (a set of state names): annotated nodes allow their states,
        everything else (and empty annotations) allows all states (ml.py:293-318).  Also records which nodes
        "have a state" in the sense of ml.py:329-331.
        """
        state2index = dict(zip(self.states, range(self.k)))
        masks = np.ones((self.N, self.k), dtype=np.int8)
        annotated = np.zeros(self.N, dtype=bool)
        character = self.character
        for i, node in enumerate(self.nodes):
            value = getattr(node, character, None)
            if value is not None and value != '':
                annotated[i] = True
            if value:
                masks[i] = 0
                for state in value:
                    masks[i, state2index[state]] = 1
        self.masks = masks
        self.annotated = annotated

    def alter_zero_node_allowed_states(self):
        """
        Annotated nodes joined by zero-length branches whose masks have no common state all get the union of
        their masks; the previous masks are remembered (ml.py:352-387).  Returns the ids of the altered nodes.
        """
        if self._top is None:
            self._top = zero_branch_tops(self.flat)
        ids = np.flatnonzero(self.annotated)
        if len(ids) < 2:
            return np.zeros(0, dtype=np.int64)
        tops = self._top[ids]
        order = np.argsort(tops, kind='stable')
        ids, tops = ids[order], tops[order]
        bounds = np.flatnonzero(np.concatenate(([True], tops[1:] != tops[:-1], [True])))
        altered = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            if b - a < 2:
                continue
            members = ids[a:b]
            m = self.masks[members]
            if m.sum(axis=0).max() == len(members):
                continue
            union = (m.sum(axis=0) > 0).astype(np.int8)
            self.init_masks[members] = m
            self.has_init[members] = True
            self.masks[members] = union
            altered.extend(members.tolist())
        return np.array(altered, dtype=np.int64)

    def unalter_zero_node_allowed_states(self, altered):
        """masks & saved masks, or the saved ones if nothing is left (ml.py:390-405)."""
        for n in altered:
            both = self.masks[n] & self.init_masks[n]
            self.masks[n] = both if np.any(both > 0) else self.init_masks[n]

    # ------------------------------------------------------------------------------------------------ device
    def _sync_device(self, model, masks_before_alteration=None):
        spec = model.kernel_spec()
        key = (spec['kind'], model.rate_params(), tuple(np.asarray(v).tobytes() if isinstance(v, np.ndarray) else v
                                                         for k_, v in sorted(spec.items())))
        if key != self._uploaded_model:
            self.engine.set_models([model])
            self._uploaded_model = key
        if self._uploaded_masks is None or not np.array_equal(self._uploaded_masks, self.masks):
            self.engine.set_masks(self.masks)
            self._uploaded_masks = self.masks.copy()
        self.engine.set_initial_masks(masks_before_alteration)

    def _raise_likelihood_error(self, e):
        parent, child = int(e.err_parent[0]), int(e.err_child[0])
        raise PastMLLikelihoodError("The parent node {} and its child node {} have non-intersecting states, "
                                    "and are connected by a zero-length ({:g}) branch. "
                                    "This creates a zero likelihood value. "
                                    "To avoid this issue check the restrictions on these node states "
                                    "and/or use a smoothing factor (tau)."
                                    .format(self.nodes[parent].name, self.nodes[child].name, self.flat.dist[child]))

    def bottom_up_loglikelihood(self, model, is_marginal=True, alter=True):
        """
        Sum over the trees of the forest of get_bottom_up_loglikelihood (ml.py:82-121): optional alteration of the
        masks, one device sweep, restoration of the masks (marginal) -- for the joint sweep the arg-max tables of the
        altered nodes are rewritten on the device instead (ml.py:115-119).
        """
        altered = np.zeros(0, dtype=np.int64)
        before = None
        if 0 == model.tau and alter:
            before = self.masks.copy()
            altered = self.alter_zero_node_allowed_states()
        self._sync_device(model, before if (not is_marginal and len(altered)) else None)
        self.n_sweeps += 1
        try:
            lnl = self.engine.bottom_up(is_marginal)[0]
        except hip.ZeroLikelihoodError as e:
            self._raise_likelihood_error(e)
        if len(altered) and is_marginal:
            self.unalter_zero_node_allowed_states(altered)
        return float(lnl)

    def batch_loglikelihoods(self, model, parameter_vectors):
        """
        Marginal log-likelihoods (alter=True semantics) for several parameter vectors of the optimiser at once:
        one device column per vector, ONE bottom-up sweep for all of them.  This is what makes the finite-difference
        gradient of the optimiser (pastml/ml.py:231, scipy's 2-point scheme: n_params + 1 evaluations) cost one launch
        sequence instead of n_params + 1.  Every column is computed independently and deterministically, so the values
        are bit-identical to evaluating the vectors one by one.  The model is left at the last vector.
        """
        C = len(parameter_vectors)
        cache = self.__dict__.setdefault('_batch_engines', {})
        if C not in cache:
            cache[C] = dict(engine=hip.acquire_engine(self.flat, C, self.k, device=self._device), masks=[None] * C)
        slot = cache[C]
        engine = slot['engine']
        specs, variants = [], []
        for ps in parameter_vectors:
            model.set_params_from_optimised(ps)
            specs.append((model.kernel_spec(), model.rate_params()))
            variants.append(0 == model.tau)
        # masks: altered (tau == 0) or as they are; the alteration does not depend on the other parameters, only on
        # the masks themselves, so it is computed once per state of the masks (the optimiser calls this hundreds of
        # times between two changes of them) and the columns remember which version they hold
        plain = self.masks
        plain_key = hash(plain.tobytes())
        altered_masks, altered_key = None, None
        if any(variants):
            memo = self.__dict__.get('_alter_memo')
            if memo is not None and memo[0] == plain_key:
                altered_masks, altered_key = memo[1], memo[2]
            else:
                keep = (self.masks.copy(), self.init_masks.copy(), self.has_init.copy())
                altered = self.alter_zero_node_allowed_states()
                altered_masks = self.masks.copy()
                altered_key = hash(altered_masks.tobytes())
                # the evaluation itself leaves masks as they were (marginal sweeps un-alter, ml.py:115-117), but the
                # saved '.initial' masks stay, exactly as after a sequence of single evaluations
                self.masks = keep[0]
                if not len(altered):
                    self.init_masks, self.has_init = keep[1], keep[2]
                self._alter_memo = (plain_key, altered_masks, altered_key)
        for col, variant in enumerate(variants):
            wanted, wanted_key = (altered_masks, altered_key) if variant else (plain, plain_key)
            if slot['masks'][col] != wanted_key:
                engine.set_masks(wanted, col_begin=col)
                slot['masks'][col] = wanted_key
        engine.set_models(specs)
        self.n_sweeps += C
        try:
            return engine.bottom_up(True)
        except hip.ZeroLikelihoodError as e:
            first = int(np.flatnonzero(e.err_child >= 0)[0])
            e.err_parent, e.err_child = e.err_parent[first:first + 1], e.err_child[first:first + 1]
            self._raise_likelihood_error(e)

    def select_on_device(self, method, force_joint=False):
        """
        MAP / MPPA selection (ml.py:505-595) by ``pml_select_states`` from the posteriors of the last marginal pass;
        the marginal likelihoods of nodes with saved ('.initial') masks are restricted to them first.  The selected
        masks become both the device's and this object's masks.  Returns the number of selected states per node.
        """
        lh_masks = None
        if np.any(self.has_init):
            lh_masks = np.ones((1, self.N, self.k), dtype=np.int8)
            lh_masks[0, self.has_init] = self.init_masks[self.has_init]
        sel, nsel = self.engine.select_states(method, force_joint=force_joint, lh_masks=lh_masks)
        self.masks = sel[0]
        self._uploaded_masks = self.masks.copy()
        return nsel[0].astype(np.int64)

    def joint_states(self):
        """Joint state of every node after a joint sweep (ml.py:598-622)."""
        return self.engine.joint_backtrace()[0].astype(np.int64)

    def top_down_marginals(self):
        """
        After a marginal sweep: top-down sweep, marginal likelihoods and posteriors (ml.py:240-290, 431-502).
        Returns (posterior [N, k], lh [N, k], lh_sf [N]) where lh / lh_sf play the role of the reference's
        LIKELIHOOD / LIKELIHOOD_SF features: log10(lh.sum()) - lh_sf is the node's total log10-likelihood.
        """
        post, lh_sum, lh_sf = self.engine.top_down_marginals()
        return post[0], post[0] * lh_sum[0][:, None], lh_sf[0]

    # ------------------------------------------------------------------------------------------------ row order
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
    for i, node in enumerate(problem.nodes):
        problem.masks[i] = getattr(node, A)
        value = getattr(node, character, None)
        problem.annotated[i] = value is not None and value != ''
        if hasattr(node, init):
            problem.has_init[i] = True
            problem.init_masks[i] = getattr(node, init)


def _push_masks_to_features(problem, character):
    A = get_personalized_feature_name(character, ALLOWED_STATES)
    init = A + '.initial'
    for i, node in enumerate(problem.nodes):
        node.add_feature(A, problem.masks[i].astype(int))
        if problem.has_init[i]:
            node.add_feature(init, problem.init_masks[i].astype(int))


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
# parameter optimisation
# =====================================================================================================================

def optimize_likelihood_params(forest, character, observed_frequencies, model, problem=None):
    """
    L-BFGS-B over the free model parameters, two deterministic starting points (current values, observed
    frequencies), then random restarts (ml.py:174-237).  Every function evaluation is one bottom-up sweep on the GPU.
    """
    own = problem is None
    if own:
        problem = ForestProblem(forest, character, model.states)
        problem.initialize_allowed_states()
    try:
        bounds = model.get_bounds()

        def get_v(ps):
            if np.any(pd.isnull(ps)):
                return np.nan
            model.set_params_from_optimised(ps)
            res = problem.bottom_up_loglikelihood(model, is_marginal=True, alter=True)
            return np.inf if pd.isnull(res) else -res

        if np.any(observed_frequencies <= 0):
            observed_frequencies = np.maximum(observed_frequencies, 1e-10)

        x0_JC = model.get_optimised_parameters()
        optimise_frequencies = isinstance(model, ModelWithFrequencies) and model._optimise_frequencies
        x0_EFT = x0_JC
        if optimise_frequencies:
            model.frequencies = observed_frequencies
            x0_EFT = model.get_optimised_parameters()
        log_lh_JC = -get_v(x0_JC)
        log_lh_EFT = log_lh_JC if not optimise_frequencies else -get_v(x0_EFT)

        best_log_lh = max(log_lh_JC, log_lh_EFT)

        lower, upper = bounds[:, 0], bounds[:, 1]

        def get_v_and_gradient(ps):
            """
            Value and the 2-point finite-difference gradient scipy's L-BFGS-B would compute itself (abs_step 1e-8,
            steps flipped at the bounds), with all n_params + 1 likelihoods evaluated in one batched device sweep.
            scipy's own approx_derivative runs twice -- first to record the points it asks for, then on the table of
            their values -- so points and arithmetic are scipy's, and the iterates are those of the unbatched run.
            """
            ps = np.asarray(ps, dtype=np.float64)
            if np.any(pd.isnull(ps)):
                return np.nan, np.full(len(ps), np.nan)
            asked = []

            def record(x):
                asked.append(np.array(x, dtype=np.float64))
                return 0.0

            approx_derivative(record, ps, method='2-point', abs_step=1e-8, f0=0.0, bounds=(lower, upper))
            values = problem.batch_loglikelihoods(model, [ps] + asked)
            values = [np.inf if pd.isnull(v) else -v for v in values]
            table = {x.tobytes(): v for x, v in zip(asked, values[1:])}
            gradient = approx_derivative(lambda x: table[np.asarray(x, dtype=np.float64).tobytes()], ps,
                                         method='2-point', abs_step=1e-8, f0=values[0], bounds=(lower, upper))
            model.set_params_from_optimised(ps)
            return values[0], gradient

        batched = os.environ.get('PASTML_AMD_BATCHED_OPTIMISER', '1') != '0'

        for i in range(100):
            if i == 0:
                vs = x0_JC
            elif optimise_frequencies and i == 1:
                vs = x0_EFT
            else:
                vs = np.random.uniform(bounds[:, 0], bounds[:, 1])
            if batched:
                fres = minimize(get_v_and_gradient, x0=vs, method='L-BFGS-B', bounds=bounds, jac=True)
            else:
                fres = minimize(get_v, x0=vs, method='L-BFGS-B', bounds=bounds)
            if fres.success and not np.any(np.isnan(fres.x)):
                if -fres.fun >= best_log_lh:
                    model.set_params_from_optimised(fres.x)
                    return -fres.fun
        model.set_params_from_optimised(x0_JC if log_lh_JC >= log_lh_EFT else x0_EFT)
        return best_log_lh
    finally:
        if own:
            problem.close()


def optimise_likelihood(forest, character, model, observed_frequencies, problem=None):
    """Initial likelihood, then basic parameters (sf, tau), then all of them (ml.py:865-920)."""
    own = problem is None
    if own:
        problem = ForestProblem(forest, character, model.states)
    try:
        problem.initialize_allowed_states()
        logger = logging.getLogger('pastml')
        likelihood = problem.bottom_up_loglikelihood(model, is_marginal=True, alter=True)
        failure = 'Failed to {} the likelihood for your tree, please check that you do not have contradicting {} ' \
                  'states specified for internal tree nodes, ' \
                  'and if not - submit a bug at https://github.com/evolbioinfo/pastml/issues'
        if np.isnan(likelihood):
            raise PastMLLikelihoodError(failure.format('calculate', character))
        if not model.get_num_params():
            logger.debug('All the parameters are fixed for {}:\n{}{}.'
                         .format(character, model._print_parameters(), '\tlog likelihood:\t{:.6f}'.format(likelihood)))
        else:
            logger.debug('Initial values for {} parameter optimisation:\n{}{}.'
                         .format(character, model._print_parameters(), '\tlog likelihood:\t{:.6f}'.format(likelihood)))
            if not model.basic_params_fixed():
                model.fix_extra_params()
                likelihood = optimize_likelihood_params(forest=forest, character=character, model=model,
                                                        observed_frequencies=observed_frequencies, problem=problem)
                if np.any(np.isnan(likelihood) or likelihood == -np.inf):
                    raise PastMLLikelihoodError(failure.format('optimise', character))
                model.unfix_extra_params()
                if not model.extra_params_fixed():
                    logger.debug('Pre-optimised basic parameters for {}:\n{}{}.'
                                 .format(character, model._print_basic_parameters(),
                                         '\tlog likelihood:\t{:.6f}'.format(likelihood)))
            if not model.extra_params_fixed():
                likelihood = optimize_likelihood_params(forest=forest, character=character, model=model,
                                                        observed_frequencies=observed_frequencies, problem=problem)
                if np.any(np.isnan(likelihood) or likelihood == -np.inf):
                    raise PastMLLikelihoodError(failure.format('calculate', character))
            logger.debug('Optimised parameters for {}:\n{}{}'
                         .format(character, model._print_parameters(), '\tlog likelihood:\t{:.6f}'.format(likelihood)))
        return likelihood
    finally:
        if own:
            problem.close()


# =====================================================================================================================
# ml_acr
# =====================================================================================================================

def ml_acr(forest, character, prediction_method, model, observed_frequencies, force_joint=True):
    """
    ML states on the trees, stored in node features; returns the list of result dictionaries (ml.py:640-750).

    Node features written (as the reference leaves them): ``<character>`` (set of selected states),
    ``<character>_ALLOWED_STATES``, ``<character>_JOINT_STATE`` (unless MAP), ``<character>_LIKELIHOOD`` /
    ``_LIKELIHOOD_SF`` (marginal methods).
    """
    if ALL == prediction_method:
        raise NotImplementedError('The ALL meta-method additionally needs the parsimony methods '
                                  '(pastml/parsimony.py), which are outside the accelerated path; use ML, MPPA, '
                                  'MAP or JOINT.')
    if isinstance(forest, TreeNode):
        forest = [forest]
    logger = logging.getLogger('pastml')
    problem = ForestProblem(forest, character, model.states)
    try:
        likelihood = optimise_likelihood(forest=forest, character=character, model=model,
                                         observed_frequencies=observed_frequencies, problem=problem)
        result = {LOG_LIKELIHOOD: likelihood, CHARACTER: character, METHOD: prediction_method, MODEL: model,
                  STATES: model.states}
        results = []
        nodes = problem.nodes
        states = model.states
        A = get_personalized_feature_name(character, ALLOWED_STATES)

        def process_reconstructed_states(method):
            if method == prediction_method or is_meta_ml(prediction_method):
                method_character = get_personalized_feature_name(character, method) \
                    if prediction_method != method else character
                # convert_allowed_states2feature (ml.py:923-928)
                for i, node in enumerate(nodes):
                    node.add_feature(method_character, set(states[problem.masks[i].astype(bool)]))
                res = result.copy()
                res[CHARACTER] = method_character
                res[METHOD] = method
                results.append(res)

        def note_restricted_likelihood(method, restricted_likelihood):
            logger.debug('Log likelihood for {} after {} state selection:\t{:.6f}'
                         .format(character, method, restricted_likelihood))
            result[RESTRICTED_LOG_LIKELIHOOD_FORMAT_STR.format(method)] = restricted_likelihood

        def process_restricted_likelihood_and_states(method):
            restricted_likelihood = problem.bottom_up_loglikelihood(model, is_marginal=True, alter=True)
            note_restricted_likelihood(method, restricted_likelihood)
            process_reconstructed_states(method)

        joint_state = None
        if prediction_method != MAP:
            restricted_likelihood = problem.bottom_up_loglikelihood(model, is_marginal=False, alter=True)
            note_restricted_likelihood(JOINT, restricted_likelihood)
            joint_state = problem.joint_states()
            problem.masks = np.zeros((problem.N, problem.k), dtype=np.int8)
            problem.masks[np.arange(problem.N), joint_state] = 1
            f = get_personalized_feature_name(character, JOINT_STATE)
            for i, node in enumerate(nodes):
                node.add_feature(f, joint_state[i])
            process_reconstructed_states(JOINT)

        if is_marginal(prediction_method):
            problem.initialize_allowed_states()
            altered = np.zeros(0, dtype=np.int64)
            if 0 == model.tau:
                altered = problem.alter_zero_node_allowed_states()
            problem.bottom_up_loglikelihood(model, is_marginal=True, alter=False)
            posterior, lh, lh_sf = problem.top_down_marginals()
            order = problem.per_tree_order()
            result[MARGINAL_PROBABILITIES] = pd.DataFrame(posterior[order], index=[nodes[i].name for i in order],
                                                          columns=states)
            if len(altered):
                problem.unalter_zero_node_allowed_states(altered)
            # MAP (ml.py:577-595): likelihoods of nodes that were ever altered are masked by their saved masks;
            # the selection itself runs on the device and leaves the selected masks there for the restricted sweep
            lh[problem.has_init] *= problem.init_masks[problem.has_init]
            problem.select_on_device('MAP')
            process_restricted_likelihood_and_states(MAP)

            if MPPA == prediction_method or is_meta_ml(prediction_method):
                # the restricted-MAP sweep may have saved new masks (ml.py:541-542 after :675-680)
                lh[problem.has_init] *= problem.init_masks[problem.has_init]
                best_k = problem.select_on_device('MPPA', force_joint=force_joint)
                num_nodes = model.forest_stats.num_nodes
                num_scenarios = 1
                for m in best_k[best_k > 1].tolist():
                    num_scenarios *= m
                result[NUM_SCENARIOS] = num_scenarios
                result[NUM_UNRESOLVED_NODES] = int((best_k > 1).sum())
                result[NUM_STATES_PER_NODE] = int(best_k.sum()) / num_nodes
                result[PERC_UNRESOLVED] = result[NUM_UNRESOLVED_NODES] * 100 / num_nodes
                logger.debug('{} node{} unresolved ({:.2f}%) for {} by {}, i.e. {:.4f} state{} per node in average.'
                             .format(result[NUM_UNRESOLVED_NODES],
                                     's are' if result[NUM_UNRESOLVED_NODES] != 1 else ' is',
                                     result[PERC_UNRESOLVED], character, MPPA, result[NUM_STATES_PER_NODE],
                                     's' if result[NUM_STATES_PER_NODE] > 1 else ''))
                process_restricted_likelihood_and_states(MPPA)

            lh_feature = get_personalized_feature_name(character, LH)
            lh_sf_feature = get_personalized_feature_name(character, LH_SF)
            for i, node in enumerate(nodes):
                node.add_feature(lh_feature, lh[i])
                node.add_feature(lh_sf_feature, lh_sf[i])

        for i, node in enumerate(nodes):
            node.add_feature(A, problem.masks[i].astype(int))
        return results
    finally:
        problem.close()


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
    ``pml_marginal_counts``) unless zero-branch handling altered some nodes: their special rules (ml.py:806-812,
    840-853) stay on the host with numpy's generator, as in the reference.  Either way only statistical parity with the
    reference is meaningful.

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
        if device_sampling and not len(altered):
            # no node altered by the zero-branch handling: the scenarios are drawn on the device (same scheme, a
            # counter-based generator seeded from numpy's global one, so np.random.seed still fixes the result)
            return problem.engine.marginal_counts(n_repetitions, seed=int(np.random.randint(0, 2 ** 62, dtype=np.int64)))
        bu = problem.engine.download(hip.BUF_BU)
        P = problem.engine.pij_batch(copy_out=True)[0]
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
