"""
Names of the tables written around the likelihood path (the part of pastml/file.py that `_serialize_acr` needs):
parameter tables and marginal-probability tables, same templates as the reference so that files are interchangeable.
"""
from pastml_amd import get_personalized_feature_name
from pastml_amd.ml import is_ml, is_marginal, is_meta_ml, get_default_ml_method

PASTML_ML_PARAMS_TAB = 'params.character_{state}.method_{method}.model_{model}.tab'
PASTML_MP_PARAMS_TAB = 'params.character_{state}.method_{method}.tab'
PASTML_MARGINAL_PROBS_TAB = 'marginal_probabilities.character_{state}.model_{model}.tab'


def col_name2cat(column):
    """Keeps letters, digits and underscores; spaces become underscores (pastml/__init__.py:56-66)."""
    return ''.join(s for s in column.replace(' ', '_') if s.isalnum() or '_' == s)


def get_column_method(column, method):
    """pastml/file.py:18-26 for the ML methods."""
    column = col_name2cat(column)
    if is_meta_ml(method):
        method = get_default_ml_method()
        return get_personalized_feature_name(column, method), method
    return column, method


def get_pastml_parameter_file(method, model, column):
    """pastml/file.py:29-43."""
    template = PASTML_ML_PARAMS_TAB if is_ml(method) else PASTML_MP_PARAMS_TAB
    column, method = get_column_method(column, method)
    return template.format(state=column, method=method, model=model)


def get_pastml_marginal_prob_file(method, model, column):
    """pastml/file.py:90-103; None for methods without marginal probabilities."""
    if not is_marginal(method):
        return None
    column, method = get_column_method(column, method)
    return PASTML_MARGINAL_PROBS_TAB.format(state=column, model=model)
