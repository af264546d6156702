"""
``acr()``: the public entry point of the maximum-likelihood ancestral character reconstruction
(same signature, defaults and result dictionaries as pastml/acr.py:76-279).

Only the ML path is implemented here (prediction methods MPPA, MAP, JOINT, ML): every likelihood sweep runs on the
GPU.  The parsimony / COPY methods, polytomy resolution, the pipeline around it (I/O, HTML) are out of scope of this
package and raise a clear error.
"""
import logging
import os
import warnings
from multiprocessing.pool import ThreadPool

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
from pastml_amd.tree import TreeNode

model2class = {F81: F81Model, JC: JCModel, CUSTOM_RATES: CustomRatesModel, HKY: HKYModel, JTT: JTTModel, EFT: EFTModel}

COPY = 'COPY'
MP_METHODS = {'DOWNPASS', 'ACCTRAN', 'DELTRAN', 'MP'}

warnings.filterwarnings("ignore", append=True)


def _serialize_acr(args):
    """
    Writes the parameter table (and, for marginal methods, the marginal-probability table) of one ACR result into
    ``work_dir``, in the reference's format (pastml/acr.py:45-73): the parameter file can be fed back through
    ``column2parameters`` (to PastML or to this package).
    """
    from pastml_amd import PASTML_VERSION
    from pastml_amd.file import get_pastml_parameter_file, get_pastml_marginal_prob_file
    from pastml_amd.ml import MODEL
    acr_result, work_dir = args
    out_param_file = os.path.join(work_dir, get_pastml_parameter_file(
        method=acr_result[METHOD], model=acr_result[MODEL].name if MODEL in acr_result else None,
        column=acr_result[CHARACTER]))
    with open(out_param_file, 'w+') as f:
        f.write('parameter\tvalue\n')
        f.write('pastml_version\t{}\n'.format(PASTML_VERSION))
        for name in sorted(acr_result.keys()):
            if name not in [STATES, MARGINAL_PROBABILITIES, METHOD, MODEL]:
                f.write('{}\t{}\n'.format(name, acr_result[name]))
        f.write('{}\t{}\n'.format(METHOD, acr_result[METHOD]))
        if is_ml(acr_result[METHOD]):
            acr_result[MODEL].save_parameters(f)
    logging.getLogger('pastml').debug('Serialized ACR parameters and statistics for {} to {}.'
                                      .format(acr_result[CHARACTER], out_param_file))
    if is_marginal(acr_result[METHOD]):
        out_mp_file = os.path.join(work_dir, get_pastml_marginal_prob_file(
            method=acr_result[METHOD], model=acr_result[MODEL].name, column=acr_result[CHARACTER]))
        acr_result[MARGINAL_PROBABILITIES].to_csv(out_mp_file, sep='\t', index_label='node')
        logging.getLogger('pastml').debug('Serialized marginal probabilities for {} to {}.'
                                          .format(acr_result[CHARACTER], out_mp_file))


def calculate_observed_freqs(character, forest, states):
    """
    Tip-state frequencies (a tip with several states contributes 1/n to each) and the fraction of tips without a
    state (pastml/acr.py:282-299).
    """
    n = len(states)
    missing_data = 0.
    state2index = dict(zip(states, range(n)))
    observed_frequencies = np.zeros(n, np.float64)
    for tree in forest:
        for tip in tree:
            state = getattr(tip, character, set())
            if state:
                num_node_states = len(state)
                for _ in state:
                    observed_frequencies[state2index[_]] += 1. / num_node_states
            else:
                missing_data += 1
    total_count = observed_frequencies.sum() + missing_data
    observed_frequencies /= observed_frequencies.sum()
    missing_data /= total_count
    return missing_data, observed_frequencies, state2index


def flatten_lists(lists):
    result = []
    for _ in lists:
        if isinstance(_, list):
            result.extend(_)
        else:
            result.append(_)
    return result


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
    :param threads: number of characters processed concurrently (0 = number of CPUs)
    :param reoptimise: treat given parameters as starting values
    :param tau: smoothing factor added to the branch lengths (0: zero branches are handled by state alteration)
    :return: list of ACR result dictionaries
    """
    if resolve_polytomies:
        raise NotImplementedError('resolve_polytomies (tree editing, pastml/tree.py:344-492) is outside the '
                                  'accelerated likelihood path')
    if isinstance(forest, TreeNode):
        forest = [forest]

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
    logger = logging.getLogger('pastml')
    logger.debug('\n=============ACR===============================')

    column2parameters = column2parameters if column2parameters else {}
    column2rates = column2rates if column2rates else {}

    prediction_methods = value2list(len(columns), prediction_method, MPPA)
    models = value2list(len(columns), model, F81)

    def get_states(method, model, column):
        initial_states = column2states[column]
        if not is_ml(method) or model not in {HKY, JTT}:
            return initial_states
        states = HKY_STATES if HKY == model else JTT_STATES
        if not set(initial_states) & set(states):
            raise ValueError('The allowed states for model {} are {}, '
                             'but your annotation file specifies {} as states in column {}.'
                             .format(model, ', '.join(states), ', '.join(initial_states), column))
        state_set = set(states)
        for root in forest:
            for n in root.traverse():
                if hasattr(n, column):
                    n.add_feature(column, state_set & getattr(n, column))
        return states

    character2settings = {}
    for (character, method, model_name) in zip(columns, prediction_methods, models):
        logger.debug('ACR settings for {}:\n\tMethod:\t{}{}.'
                     .format(character, method, '\n\tModel:\t{}'.format(model_name)
                             if model_name and is_ml(method) else ''))
        if COPY == method or method in MP_METHODS or ALL == method:
            raise NotImplementedError('Method {} is outside the accelerated maximum-likelihood path; '
                                      'supported: {}'.format(method, ', '.join(sorted(ML_METHODS | {ML}))))
        if not is_ml(method):
            raise ValueError('Method {} is unknown, should be one of ML ({})'.format(method, ', '.join(ML_METHODS)))
        if model_name not in model2class:
            raise ValueError('Model {} is unknown, should be one of {}'.format(model_name, ', '.join(model2class)))
        params = column2parameters[character] if character in column2parameters else None
        rate_file = column2rates[character] if character in column2rates else None
        optimise_tau = tau is None or reoptimise
        if tau is None:
            tau = 0
        states = get_states(method, model_name, character)
        missing_data, observed_frequencies, state2index = calculate_observed_freqs(character, forest, states)
        logger.debug('Observed frequencies for {}:{}{}.'
                     .format(character,
                             ''.join('\n\tfrequency of {}:\t{:.6f}'.format(state, observed_frequencies[state2index[state]])
                                     for state in states),
                             '\n\tfraction of missing data:\t{:.6f}'.format(missing_data) if missing_data else ''))
        model_instance = model2class[model_name](parameter_file=params, rate_matrix_file=rate_file,
                                                 reoptimise=reoptimise, frequency_smoothing=frequency_smoothing,
                                                 tau=tau, optimise_tau=optimise_tau, states=states,
                                                 forest_stats=forest_stats,
                                                 observed_frequencies=observed_frequencies, character=character)
        character2settings[character] = [method, model_instance, observed_frequencies]

    if threads < 1:
        threads = max(os.cpu_count(), 1)

    def _work(character):
        method, model_instance, observed_frequencies = character2settings[character]
        return ml_acr(forest=forest, character=character, prediction_method=method, model=model_instance,
                      force_joint=force_joint, observed_frequencies=observed_frequencies)

    if threads > 1 and len(character2settings) > 1:
        with ThreadPool(processes=min(threads - 1, len(character2settings))) as pool:
            acr_results = pool.map(func=_work, iterable=character2settings.keys())
    else:
        acr_results = [_work(character) for character in character2settings.keys()]

    return flatten_lists(acr_results)
