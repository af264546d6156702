"""General time-reversible model with a user rate matrix (reference: pastml/models/CustomRatesModel.py)."""
import logging

import numpy as np

from pastml_amd.models import ModelWithFrequencies, KIND_EIGEN
from pastml_amd.models.generator import get_diagonalisation

CUSTOM_RATES = 'CUSTOM_RATES'


def load_custom_rates(infile):
    """
    Reads a symmetric rate matrix whose first line is '# state names'; states and matrix are returned sorted by
    state name (CustomRatesModel.py:11-32).
    """
    rates = np.loadtxt(infile, dtype=np.float64, comments='#', delimiter=' ')
    if rates.ndim != 2 or rates.shape[0] != rates.shape[1]:
        raise ValueError('The input rate matrix must be squared, but yours is {}.'
                         .format('x'.join(str(_) for _ in rates.shape)))
    if not np.all(rates == rates.transpose()):
        raise ValueError('The input rate matrix must be symmetric, but yours is not.')
    np.fill_diagonal(rates, 0)
    n = len(rates)
    if np.count_nonzero(rates) != n * (n - 1):
        logging.getLogger('pastml').warning('The rate matrix contains zero rates (apart from the diagonal).')
    with open(infile, 'r') as f:
        header = f.readline()
    if not header.startswith('#'):
        raise ValueError('The rate matrix file should start with state names, '
                         'separated by whitespaces and preceded by # .')
    states = np.array(header.strip('#').strip('\n').strip().split(' '), dtype=str)
    if len(states) != n:
        raise ValueError('The number of specified state names ({}) does not correspond to the rate matrix '
                         'dimensions ({}x{}).'.format(len(states), *rates.shape))
    order = np.argsort(states)
    return states[order], rates[:, order][order, :]


class CustomRatesModel(ModelWithFrequencies):

    def __init__(self, forest_stats, sf=None, frequencies=None, rate_matrix_file=None, states=None, rate_matrix=None,
                 tau=0, optimise_tau=False, frequency_smoothing=False, parameter_file=None, reoptimise=False,
                 **kwargs):
        ModelWithFrequencies.__init__(self, states=states, forest_stats=forest_stats, sf=sf, tau=tau,
                                      frequencies=frequencies, optimise_tau=optimise_tau,
                                      frequency_smoothing=frequency_smoothing, reoptimise=reoptimise,
                                      parameter_file=parameter_file, **kwargs)
        self.name = CUSTOM_RATES
        if rate_matrix_file is None and (rate_matrix is None or states is None):
            raise ValueError('Either the rate matrix file '
                             'or the rate matrix plus the states must be specified for {} model'.format(CUSTOM_RATES))
        if rate_matrix_file is None:
            self._rate_matrix = rate_matrix
        else:
            self._states, self._rate_matrix = load_custom_rates(rate_matrix_file)
        self._diagonalise()

    def _diagonalise(self):
        # once per frequency change, as the reference does (CustomRatesModel.py:52,68)
        self.D_DIAGONAL, self.A, self.A_INV = get_diagonalisation(self._frequencies, self._rate_matrix)

    @property
    def rate_matrix(self):
        return self._rate_matrix

    @rate_matrix.setter
    def rate_matrix(self, rate_matrix):
        raise NotImplementedError('The rate matrix is preset and cannot be changed.')

    @ModelWithFrequencies.frequencies.setter
    def frequencies(self, frequencies):
        if not (self._optimise_frequencies or self._frequency_smoothing):
            raise NotImplementedError('The frequencies are preset and cannot be changed.')
        self._frequencies = frequencies
        self._diagonalise()

    def kernel_spec(self):
        return dict(kind=KIND_EIGEN, pi=np.ascontiguousarray(self.frequencies, dtype=np.float64),
                    d=np.ascontiguousarray(self.D_DIAGONAL, dtype=np.float64),
                    A=np.ascontiguousarray(self.A, dtype=np.float64),
                    Ainv=np.ascontiguousarray(self.A_INV, dtype=np.float64))
