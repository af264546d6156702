"""
``acr()``: the public entry point of the maximum-likelihood ancestral character reconstruction -- signature, defaults,
errors and result dictionaries of pastml/acr.py:76-279.

The reference walks the characters one by one (a thread pool over ``ml_acr`` calls, acr.py:213-231).  Here the
characters of a call are the columns of the device sweeps: they are turned into :class:`pastml_amd.batch.Task` objects,
grouped by (number of states, model family, method) and every group is reconstructed as ONE batch -- parameter
optimisation of all its characters in lock-step, one joint sweep, one marginal pass, one selection per method
(pastml_amd.batch).  Under a multi-process launch (one process per GPU, pastml_amd.sharding) every rank takes a
contiguous block of the characters and returns the results of its block; :func:`total_log_likelihood` sums the
log-likelihoods over all ranks with the library's single RCCL all-reduce.

Every prediction method of the reference is accepted: the likelihood methods (MPPA, MAP, JOINT, and the meta-methods
ML and ALL) run their sweeps on the GPU; the parsimony methods (DOWNPASS, ACCTRAN, DELTRAN, MP; integer set work, no
likelihoods) run on the host as level-wise array passes (pastml_amd.parsimony); COPY reports the annotation as it is.
Polytomy resolution (tree editing) and the HTML side of the pipeline are out of scope and raise a
clear error; ``pastml_amd.pipeline`` covers the file-to-file part.
"""
import logging
import os
import warnings

import numpy as np
import pandas as pd

from pastml_amd import value2list, STATES, METHOD, CHARACTER
from pastml_amd.annotation import preannotate_forest, ForestStats
from pastml_amd.ml import is_ml, MPPA, ml_acr, ML_METHODS, MAP, JOINT, ALL, ML, META_ML_METHODS, \
    MARGINAL_ML_METHODS, MARGINAL_PROBABILITIES, is_marginal, get_default_ml_method  # noqa: F401
from pastml_amd.models.CustomRatesModel import CustomRatesModel, CUSTOM_RATES
from pastml_amd.models.EFTModel import EFTModel, EFT
from pastml_amd.models.F81Model import F81Model, F81
from pastml_amd.models.HKYModel import HKYModel, HKY, HKY_STATES
from pastml_amd.models.JCModel import JCModel, JC
from pastml_amd.models.JTTModel import JTTModel, JTT, JTT_STATES
from pastml_amd.tree import TreeNode, get_flat_forest, AnnotationColumn

MAX_STATES = 512   # = pastml_amd.hip.MAX_STATES = PML_MAX_STATES of the library (tests/test_host_logic.py checks); F81 family
MAX_STATES_MATRIX = 256   # ... and the models with a transition matrix per branch (HKY has 4 states; JTT 20; CUSTOM_RATES any)

model2class = {F81: F81Model, JC: JCModel, CUSTOM_RATES: CustomRatesModel, HKY: HKYModel, JTT: JTTModel, EFT: EFTModel}

from pastml_amd.parsimony import is_parsimonious, parsimonious_acr, MP_METHODS, ACCTRAN, DELTRAN, DOWNPASS, MP  # noqa: E402,F401

COPY = 'COPY'

warnings.filterwarnings("ignore", append=True)


def _table_value(value):
    """
    A result value as it goes into the parameter table.  The number of scenarios is a product over all nodes and can have
    tens of thousands of digits at 10^5 tips; Python refuses to print integers of more than 4 300 digits (and the
    reference's own writer fails there): beyond that the value is written as mantissa and decimal exponent.
    """
    if isinstance(value, int) and not isinstance(value, bool) and value.bit_length() > 14000:
        import math
        shift = value.bit_length() - 64
        log10 = math.log10(value >> shift) + shift * math.log10(2.0)
        exponent = int(math.floor(log10))
        return '{:.6f}e+{}'.format(10.0 ** (log10 - exponent), exponent)
    return value


def _serialize_acr(args):
    """
    Writes one reconstruction result into ``work_dir`` in the reference's two table formats (pastml/acr.py:45-73): the
    parameter table -- statistics and model parameters, readable back through ``column2parameters`` by PastML and by
    this package -- and, for marginal methods, the marginal-probability table (one row per node).
    """
    from pastml_amd import PASTML_VERSION
    from pastml_amd.file import get_pastml_parameter_file, get_pastml_marginal_prob_file
    from pastml_amd.ml import MODEL
    result, work_dir = args
    logger = logging.getLogger('pastml')
    method, character = result[METHOD], result[CHARACTER]
    model = result.get(MODEL)
    rows = [('pastml_version', PASTML_VERSION)]
    rows += [(key, result[key]) for key in sorted(result) if key not in (STATES, MARGINAL_PROBABILITIES, METHOD, MODEL)]
    rows.append((METHOD, method))
    path = os.path.join(work_dir, get_pastml_parameter_file(method=method, model=model.name if model is not None else None,
                                                            column=character))
    with open(path, 'w+') as out:
        out.write('parameter\tvalue\n')
        out.writelines('{}\t{}\n'.format(key, _table_value(value)) for key, value in rows)
        if is_ml(method):
            model.save_parameters(out)
    logger.debug('Serialized ACR parameters and statistics for {} to {}.'.format(character, path))
    if is_marginal(method):
        path = os.path.join(work_dir, get_pastml_marginal_prob_file(method=method, model=model.name, column=character))
        result[MARGINAL_PROBABILITIES].to_csv(path, sep='\t', index_label='node')
        logger.debug('Serialized marginal probabilities for {} to {}.'.format(character, path))


def _tip_state_words(character, forest, states, flat=None):
    """Annotation words of the tips in the reference's tip order (tree by tree, left to right)."""
    from pastml_amd.batch import annotation_words
    if flat is None:
        flat = get_flat_forest(forest)
    words, _ = annotation_words(flat, character, states)
    tips = flat.tips[np.argsort(flat.post_rank[flat.tips], kind='stable')]
    return words[tips]


def calculate_observed_freqs(character, forest, states, flat=None):
    """
    Tip-state frequencies (a tip with n states counts 1/n for each) and the fraction of tips without a state
    (pastml/acr.py:282-299).  Tips with one state -- nearly all -- are counted in one pass; tips with several are added
    in the reference's tip order, which matters for the last bits of a sum of thirds.  ``flat``: the flattened forest
    if the caller already has it (flattening re-validates the cached arrays against the tree, a walk per call).
    """
    from pastml_amd.batch import popcount
    states = np.asarray(states)
    k = len(states)
    state2index = dict(zip(states, range(k)))
    words = _tip_state_words(character, forest, states, flat)
    per_tip = popcount(words).sum(axis=-1)
    bits = np.unpackbits(np.ascontiguousarray(words).view(np.uint8), axis=-1, bitorder='little')[:, :k]
    single = per_tip == 1
    several = np.flatnonzero(per_tip > 1)
    if len(several) == 0:
        observed = bits[single].sum(axis=0).astype(np.float64)
    else:
        observed = np.zeros(k, dtype=np.float64)
        for row in np.flatnonzero(per_tip > 0):
            share = 1. / per_tip[row]
            for j in np.flatnonzero(bits[row]):
                observed[j] += share
    missing = float((per_tip == 0).sum())
    total = observed.sum() + missing
    observed /= observed.sum()
    return missing / total, observed, state2index


def flatten_lists(lists):
    out = []
    for item in lists:
        out.extend(item) if isinstance(item, list) else out.append(item)
    return out


def _restrict_annotation_to(states, column, forest, flat=None):
    """HKY / JTT work on their own alphabets: states of the annotation outside of it are dropped (acr.py:155-163)."""
    if flat is None:
        flat = get_flat_forest(forest)
    col = flat.columns.get(column)
    allowed = set(states)
    if isinstance(col, AnnotationColumn) and col.absent is None and \
            not any(column in n.__dict__ for n in flat.nodes):
        keep = np.array([v in allowed for v in col.values], dtype=bool)
        lut = np.concatenate((np.where(keep, np.cumsum(keep) - 1, -1), [-1])).astype(np.int64)
        codes = col.codes.copy()
        valued = codes >= 0
        codes[valued] = lut[codes[valued]]
        multi = {}
        for i, vs in col.multi.items():
            kept = {int(lut[j]) for j in vs if lut[j] >= 0}
            if len(kept) > 1:
                multi[i] = kept
            codes[i] = min(kept) if kept else -1
        flat.set_column(column, AnnotationColumn(codes, col.values[keep], multi))
        return
    for root in forest:
        for n in root.traverse():
            if hasattr(n, column):
                n.add_feature(column, allowed & getattr(n, column))


def total_log_likelihood(results):
    """
    Sum of the log-likelihoods of all characters of a run.  Under a multi-process launch every rank passes the results
    of its own block of characters; the sum over the ranks is the library's one collective (an 8-byte RCCL all-reduce,
    ``pml_allreduce_loglik``).
    """
    from pastml_amd import sharding
    from pastml_amd.ml import LOG_LIKELIHOOD
    seen, values = set(), []
    for r in results:
        base = r[CHARACTER][:-len(r[METHOD]) - 1] if r[CHARACTER].endswith('_' + r[METHOD]) else r[CHARACTER]
        if base not in seen and LOG_LIKELIHOOD in r:   # meta-methods report one character several times; parsimony: none
            seen.add(base)
            values.append(r[LOG_LIKELIHOOD])
    comm = sharding.communicator() or sharding.LocalCommunicator()
    return comm.allreduce_loglik(values if values else [0.0])


def acr(forest, df=None, columns=None, column2states=None, prediction_method=MPPA, model=F81,
        column2parameters=None, column2rates=None,
        force_joint=True, threads=0,
        reoptimise=False, tau=0, resolve_polytomies=False, frequency_smoothing=False):
    """
    Reconstructs ancestral states for the given tree(s) and all the characters given as columns of the annotation
    dataframe (or pre-annotated on the tree with ``columns`` + ``column2states``).

    :param forest: tree or list of trees (pastml_amd.tree.TreeNode, the ete3-like container of this package)
    :param df: dataframe indexed with node names, one column per character
    :param prediction_method: MPPA (default), MAP, JOINT or ML; one value or a list (one per column)
    :param model: F81 (default), JC, EFT, HKY, JTT or CUSTOM_RATES; one value or a list
    :param column2parameters: {column: {param: value}} or {column: path to a parameter file} to preset parameters
    :param column2rates: {column: path to a rate matrix file} for CUSTOM_RATES
    :param force_joint: add the joint state to the MPPA selection even if the Brier score would not
    :param threads: kept for compatibility; characters are batched on the device, not spread over host threads
    :param reoptimise: treat given parameters as starting values
    :param tau: smoothing factor added to the branch lengths (0: zero branches are handled by state alteration)
    :return: list of ACR result dictionaries (of this rank's characters under a multi-process launch)
    """
    from pastml_amd import sharding
    from pastml_amd.batch import Task, run_tasks
    if resolve_polytomies:
        raise NotImplementedError('resolve_polytomies (tree editing, pastml/tree.py:344-492) is outside the '
                                  'accelerated likelihood path')
    if isinstance(forest, TreeNode):
        forest = [forest]
    logger = logging.getLogger('pastml')

    if columns is None:
        if df is None:
            raise ValueError('Either the tree should be preannotated with character values '
                             'and columns and column2states specified, '
                             'or an annotation dataframe provided!')
        columns = df.columns
        column2states = {column: np.array(sorted([_ for _ in df[column].unique() if not pd.isna(_) and '' != _]))
                         for column in columns}
        preannotate_forest(forest, df=df)

    forest_stats = ForestStats(forest)
    flat = get_flat_forest(forest)   # once per call: the characters below all work on these arrays
    logger.debug('\n=============ACR===============================')
    column2parameters = column2parameters or {}
    column2rates = column2rates or {}
    methods = value2list(len(columns), prediction_method, MPPA)
    model_names = value2list(len(columns), model, F81)

    # what to do per column: ('ml', Task) / ('mp', states) / ('copy', states)
    plan = []
    for character, method, model_name in zip(columns, methods, model_names):
        logger.debug('ACR settings for {}:\n\tMethod:\t{}{}.'
                     .format(character, method, '\n\tModel:\t{}'.format(model_name) if model_name and is_ml(method) else ''))
        if not (COPY == method or is_parsimonious(method) or is_ml(method)):
            raise ValueError('Method {} is unknown, should be one of ML ({}), one of MP ({}) or {}'
                             .format(method, ', '.join(ML_METHODS), ', '.join(MP_METHODS), COPY))
        states = column2states[character]
        if is_ml(method):
            if model_name not in model2class:
                raise ValueError('Model {} is unknown, should be one of {}'.format(model_name, ', '.join(model2class)))
            # A boundary difference to the reference, which has no bound on k (int64 arg-max tables, pastml/ml.py:134; its
            # pipeline only drops columns whose values are mostly unique, acr.py:774): the device path's widest lane shape
            # holds 512 states (F81 / JC / EFT; 256 for CUSTOM_RATES, whose P(t) is a k x k matrix per branch) --
            # pml_chars_alloc / pml_model_set_eigen answer PML_ERR_UNSUPPORTED beyond.  Said here, before any work is done, by
            # name ("Limits").
            most = MAX_STATES if model_name in (F81, JC, EFT) else MAX_STATES_MATRIX
            if len(states) > most:
                raise ValueError('Character {} has {} states: the MI355X likelihood path supports at most {} states per '
                                 'character under {} (PML_ERR_UNSUPPORTED); reconstruct it with a parsimonious method or merge '
                                 'rare states.'.format(character, len(states), most, model_name))
            if model_name in (HKY, JTT):
                alphabet = HKY_STATES if HKY == model_name else JTT_STATES
                if not set(states) & set(alphabet):
                    raise ValueError('The allowed states for model {} are {}, '
                                     'but your annotation file specifies {} as states in column {}.'
                                     .format(model_name, ', '.join(alphabet), ', '.join(states), character))
                _restrict_annotation_to(alphabet, character, forest, flat)
                states = alphabet
        if COPY == method:
            plan.append(('copy', character, method, states))
            continue
        if is_parsimonious(method):
            plan.append(('mp', character, method, states))
            continue
        # As in the reference (acr.py:185-187, inside its loop over the characters): with tau=None -- "smoothing" of the
        # pipeline -- the FIRST maximum-likelihood character optimises tau and resets the argument to 0, so the later
        # characters get tau = 0, fixed (unless reoptimise).  Probably unintended there; reproduced, because the results
        # of columns 2..n depend on it (tests/test_host_logic.py pins it).
        optimise_tau = tau is None or reoptimise
        if tau is None:
            tau = 0
        missing, observed, state2index = calculate_observed_freqs(character, forest, states, flat)
        logger.debug('Observed frequencies for {}:{}{}.'.format(
            character, ''.join('\n\tfrequency of {}:\t{:.6f}'.format(s, observed[state2index[s]]) for s in states),
            '\n\tfraction of missing data:\t{:.6f}'.format(missing) if missing else ''))
        instance = model2class[model_name](parameter_file=column2parameters.get(character),
                                           rate_matrix_file=column2rates.get(character), reoptimise=reoptimise,
                                           frequency_smoothing=frequency_smoothing, tau=tau, optimise_tau=optimise_tau,
                                           states=states, forest_stats=forest_stats, observed_frequencies=observed,
                                           character=character)
        plan.append(('ml', character, method, Task(character, method, instance, observed)))

    # restart seeds of the optimisers: one per maximum-likelihood character of the CALL, drawn before the characters are
    # dealt out to the ranks, so that a character restarts from the same points however many processes share the work
    n_ml = sum(1 for item in plan if item[0] == 'ml')
    seeds_all = np.random.randint(0, 2 ** 31 - 1, size=n_ml) if n_ml else np.zeros(0, dtype=np.int64)
    ml_rank = np.cumsum([item[0] == 'ml' for item in plan]) - 1
    # one process per GPU: this rank's contiguous block of the characters
    comm = sharding.communicator()
    mine = range(len(plan))
    if comm is not None and comm.world > 1:
        mine = sharding.shard_characters(len(plan), comm.rank, comm.world)
    seeds = np.array([seeds_all[ml_rank[i]] for i in mine if plan[i][0] == 'ml'], dtype=np.int64)
    plan = [plan[i] for i in mine]
    tasks = [item[3] for item in plan if item[0] == 'ml']
    ml_results = iter(run_tasks(forest, tasks, force_joint=force_joint, flat=flat, seeds=seeds)) if tasks else iter(())
    results = []
    for kind, character, method, payload in plan:
        if kind == 'ml':
            results.append(next(ml_results))
        elif kind == 'mp':
            results.append(parsimonious_acr(forest, character, method, payload, forest_stats.num_nodes,
                                            forest_stats.num_tips))
        else:
            results.append({CHARACTER: character, STATES: payload, METHOD: method})
    return flatten_lists(results)
