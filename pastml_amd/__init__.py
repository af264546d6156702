"""
pastml_amd: MI355X-native (gfx950) implementation of PastML's maximum-likelihood ACR hot path.

Only the path named in BASELINE.json's north_star is implemented: per-branch P(t) and the bottom-up / top-down
likelihood sweeps of ``pastml/ml.py``, behind PastML's own Python API (``acr()`` / ``ml_acr()``).

The constants and helpers below are the part of the reference's ``pastml/__init__.py`` that belongs to the
boundary (result-dict keys ``pastml/__init__.py:6-15``, feature naming ``:69-78``, ``value2list`` ``:81-94``); the reference's
logger set-up is not on the path: modules ask for ``logging.getLogger('pastml')`` themselves.
"""
PASTML_VERSION = '1.9.50'

METHOD = 'method'
STATES = 'states'
CHARACTER = 'character'

NUM_SCENARIOS = 'num_scenarios'
NUM_UNRESOLVED_NODES = 'num_unresolved_nodes'
NUM_STATES_PER_NODE = 'num_states_per_node_avg'
PERC_UNRESOLVED = 'percentage_of_unresolved_nodes'
NUM_NODES = 'num_nodes'
NUM_TIPS = 'num_tips'


def get_personalized_feature_name(character, feature):
    """Feature names are prefixed by the character name (reference: pastml/__init__.py:69-78)."""
    return '{}_{}'.format(character, feature)


def value2list(n, value, default_value):
    """
    A per-column setting of ``acr()`` as a list of n entries (same results as the reference's helper,
    pastml/__init__.py:81-94): a scalar or a one-element list is repeated, ``None`` stands for the default, a shorter list
    is completed with the default.  The caller's list is never modified.
    """
    given = [] if value is None else (list(value) if isinstance(value, list) else [value])
    if len(given) <= 1:
        return (given or [default_value]) * n
    return given + [default_value] * max(0, n - len(given))
