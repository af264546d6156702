"""
pastml_amd: MI355X-native (gfx950) implementation of PastML's maximum-likelihood ACR hot path.

Only the path named in BASELINE.json's north_star is implemented: per-branch P(t) and the bottom-up / top-down
likelihood sweeps of ``pastml/ml.py``, behind PastML's own Python API (``acr()`` / ``ml_acr()``).

The constants and helpers below are the part of the reference's ``pastml/__init__.py`` that belongs to the
boundary (result-dict keys ``pastml/__init__.py:6-15``, feature naming ``:69-78``, ``value2list`` ``:81-94``).
"""
import logging

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
    """Broadcasts a per-column setting to n columns (reference: pastml/__init__.py:81-94)."""
    if value is None:
        value = default_value
    if not isinstance(value, list):
        value = [value] * n
    elif len(value) == 1:
        value = value * n
    else:
        value += [default_value] * (n - len(value))
    return value


def _set_up_pastml_logger(verbose):
    logger = logging.getLogger('pastml')
    logger.setLevel(level=logging.DEBUG if verbose else logging.ERROR)
    logger.propagate = False
    if not logger.hasHandlers():
        ch = logging.StreamHandler()
        formatter = logging.Formatter('%(name)s:%(levelname)s:%(asctime)s %(message)s', datefmt="%H:%M:%S")
        ch.setFormatter(formatter)
        logger.addHandler(ch)
    return logger
