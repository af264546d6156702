"""
Eigen-decomposed models: general time-reversible CUSTOM_RATES (pastml/models/CustomRatesModel.py, generator.py) and
JTT (pastml/models/JTTModel.py).  The decomposition happens on the host once per frequency change, as in the
reference; P(t) = A diag(exp(d t)) A^-1 for every branch is computed by libpastml_hip (FP64 matrix cores for
16 <= k <= 32).
"""
import logging
import os

import numpy as np

from pastml_amd.models import Model, ModelWithFrequencies, KIND_EIGEN


def save_matrix(states, matrix, outfile):
    """Space-separated matrix with a '# state names' header line (generator.py:4-13)."""
    np.savetxt(outfile, matrix, delimiter=' ', fmt='%.18e', header=' '.join(states))


def get_normalised_generator(frequencies, rate_matrix=None):
    """
    Q = (R o pi) with rows summing to zero, scaled so that the expected rate -sum_i pi_i q_ii is one
    (generator.py:33-51).  ``rate_matrix`` defaults to all-equal rates.
    """
    n = len(frequencies)
    if rate_matrix is None:
        rate_matrix = np.ones(shape=(n, n), dtype=np.float64) - np.eye(n)
    q = rate_matrix * frequencies
    q -= np.diag(q.sum(axis=1))
    q /= -q.diagonal().dot(frequencies)
    return q


def get_diagonalisation(frequencies, rate_matrix=None):
    """
    (d, A, A^-1) with A diag(d) A^-1 = Q, through the same numpy calls as the reference (generator.py:16-30) so that
    the three arrays are bit-identical to the reference's for identical inputs.  Should LAPACK return a complex
    pair for a (numerically) degenerate spectrum, the equivalent symmetric route
    S = Pi^1/2 Q Pi^-1/2 = U L U^T, A = Pi^-1/2 U is used instead (Q is reversible, so its spectrum is real).
    """
    q = get_normalised_generator(frequencies, rate_matrix)
    d, a = np.linalg.eig(q)
    if np.iscomplexobj(d):
        return _symmetric_route(q, frequencies)
    return d, a, np.linalg.inv(a)


# ---------------------------------------------------------------------------------------------------------------------
# Batched form: the generators of ONE finite-difference gradient diagonalised together.  The reference re-diagonalises at
# every frequency assignment (CustomRatesModel.py:62-68): k LAPACK calls per gradient, each behind a dozen small numpy
# calls.  Here the distinct frequency vectors of a batch become one (n, k, k) stack and numpy.linalg.eig / inv run once on
# it -- LAPACK per matrix inside numpy's gufunc loop, every slice bit-identical to the single call's result
# (tests/test_host_logic.py).  What it buys is the Python around LAPACK, not LAPACK: k = 20, 21 points: 2.85 -> 1.9 ms, of
# which dgeev + dgesv are 1.7.  Measured and dropped (round 6): the same LAPACK routines of numpy's bundled OpenBLAS called
# from 2 - 16 host threads through the C-ABI -- no faster than one thread, OpenBLAS serialises its small calls on its buffer
# lock (profiles/r06c_eigen_host.txt).
# ---------------------------------------------------------------------------------------------------------------------
def _batching():
    """PASTML_AMD_EIG_BATCH=0: one diagonalisation per assignment of the frequencies, as the reference does it (measurements)."""
    return os.environ.get('PASTML_AMD_EIG_BATCH', '1') != '0'


def _symmetric_route(q, frequencies):
    sq = np.sqrt(np.asarray(frequencies, dtype=np.float64))
    s = (q * sq[:, None]) / sq[None, :]
    s = (s + s.T) / 2
    d, u = np.linalg.eigh(s)
    return d, u / sq[:, None], u.T * sq[None, :]


def get_diagonalisation_batch(frequencies, rate_matrix=None):
    """
    (d, A, A^-1) of get_diagonalisation for every row of ``frequencies`` (n, k): arrays of shapes (n, k), (n, k, k),
    (n, k, k), each slice bit-identical to the single call's result.
    """
    frequencies = np.ascontiguousarray(frequencies, dtype=np.float64)
    n, k = frequencies.shape
    q = np.empty((n, k, k), dtype=np.float64)
    for i in range(n):   # (the normaliser is a BLAS dot in the reference: kept per matrix so that its bits are the same)
        q[i] = get_normalised_generator(frequencies[i], rate_matrix)
    d, a = np.linalg.eig(q)
    if np.iscomplexobj(d):   # some matrix of the stack came back complex: one by one (the real ones keep the real route)
        out = [get_diagonalisation(frequencies[i], rate_matrix) for i in range(n)]
        return (np.array([o[0] for o in out]), np.array([o[1] for o in out]), np.array([o[2] for o in out]))
    return d, a, np.linalg.inv(a)


def get_pij_matrix(t, diag, A, A_inv):
    """Host-side numpy form of A diag(exp(d t)) A^-1 (generator.py:54-65); kept for API completeness only."""
    return A.dot(np.diag(np.exp(diag * t))).dot(A_inv)


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
        # once per frequency change, as the reference does (CustomRatesModel.py:52,68) -- but not for an assignment of the
        # vector that is already diagonalised (the first stage of the search moves the scaling factor only), and not
        # inside kernel_points, which diagonalises the vectors of a whole gradient together
        if self.__dict__.get('_defer_diag'):
            return
        key = np.asarray(self._frequencies, dtype=np.float64).tobytes()
        if self.__dict__.get('_diag_key') == key and _batching():
            return
        self.D_DIAGONAL, self.A, self.A_INV = get_diagonalisation(self._frequencies, self._rate_matrix)
        self._diag_key = key

    def kernel_points(self, vectors):
        """The points of a batch (Model.kernel_points) with ONE batched diagonalisation of their distinct frequency vectors."""
        if len(vectors) < 2 or not _batching():
            return ModelWithFrequencies.kernel_points(self, vectors)
        freqs, rates = [], []
        self._defer_diag = True
        try:
            for ps in vectors:
                self.set_params_from_optimised(ps)
                freqs.append(np.ascontiguousarray(self._frequencies, dtype=np.float64))
                rates.append(self.rate_params())
        finally:
            self._defer_diag = False
        slots, index = {}, []
        for f in freqs:
            index.append(slots.setdefault(f.tobytes(), len(slots)))
        first = {}
        for j, i in enumerate(index):
            first.setdefault(i, j)
        known = self.__dict__.get('_diag_key')
        todo = [i for i in range(len(slots)) if freqs[first[i]].tobytes() != known]
        d, a, a_inv = [None] * len(slots), [None] * len(slots), [None] * len(slots)
        for i in range(len(slots)):
            if i not in todo:
                d[i], a[i], a_inv[i] = self.D_DIAGONAL, self.A, self.A_INV
        if todo:
            bd, ba, bi = get_diagonalisation_batch(np.array([freqs[first[i]] for i in todo]), self._rate_matrix)
            for q, i in enumerate(todo):
                d[i], a[i], a_inv[i] = bd[q], ba[q], bi[q]
        last = index[-1]
        self.D_DIAGONAL, self.A, self.A_INV = d[last], a[last], a_inv[last]
        self._diag_key = freqs[-1].tobytes()
        return [(dict(kind=KIND_EIGEN, pi=freqs[j], d=np.ascontiguousarray(d[i], dtype=np.float64),
                      A=np.ascontiguousarray(a[i], dtype=np.float64), Ainv=np.ascontiguousarray(a_inv[i], dtype=np.float64)),
                 rates[j]) for j, i in enumerate(index)]

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


JTT = 'JTT'

NUM_AA = 20

# amino acids in the order the published matrix is given in
_AA_PUBLISHED_ORDER = 'ARNDCQEGHILKMFPSTWYV'

# strictly-lower triangle of the symmetric exchangeability matrix, row by row (190 values)
_JTT_LOWER_TRIANGLE = (
    0.531678, 0.557967, 0.451095, 0.827445, 0.154899, 5.549530, 0.574478, 1.019843,
    0.313311, 0.105625, 0.556725, 3.021995, 0.768834, 0.521646, 0.091304, 1.066681,
    0.318483, 0.578115, 7.766557, 0.053907, 3.417706, 1.740159, 1.359652, 0.773313,
    1.272434, 0.546389, 0.231294, 1.115632, 0.219970, 3.210671, 4.025778, 1.032342,
    0.724998, 5.684080, 0.243768, 0.201696, 0.361684, 0.239195, 0.491003, 0.115968,
    0.150559, 0.078270, 0.111773, 0.053769, 0.181788, 0.310007, 0.372261, 0.137289,
    0.061486, 0.164593, 0.709004, 0.097485, 0.069492, 0.540571, 2.335139, 0.369437,
    6.529255, 2.529517, 0.282466, 0.049009, 2.966732, 1.731684, 0.269840, 0.525096,
    0.202562, 0.146481, 0.469395, 0.431045, 0.330720, 0.190001, 0.409202, 0.456901,
    0.175084, 0.130379, 0.329660, 4.831666, 3.856906, 0.624581, 0.138293, 0.065314,
    0.073481, 0.032522, 0.678335, 0.045683, 0.043829, 0.050212, 0.453428, 0.777090,
    2.500294, 0.024521, 0.436181, 1.959599, 0.710489, 0.121804, 0.127164, 0.123653,
    1.608126, 0.191994, 0.208081, 1.141961, 0.098580, 1.060504, 0.216345, 0.164215,
    0.148483, 3.887095, 1.001551, 5.057964, 0.589268, 2.155331, 0.548807, 0.312449,
    1.874296, 0.743458, 0.405119, 0.592511, 0.474478, 0.285564, 0.943971, 2.788406,
    4.582565, 0.650282, 2.351311, 0.425159, 0.469823, 0.523825, 0.331584, 0.316862,
    0.477355, 2.553806, 0.272514, 0.965641, 2.114728, 0.138904, 1.176961, 4.777647,
    0.084329, 1.257961, 0.027700, 0.057466, 1.104181, 0.172206, 0.114381, 0.544180,
    0.128193, 0.134510, 0.530324, 0.089134, 0.201334, 0.537922, 0.069965, 0.310927,
    0.080556, 0.139492, 0.235601, 0.700693, 0.453952, 2.114852, 0.254745, 0.063452,
    0.052500, 5.848400, 0.303445, 0.241094, 0.087904, 0.189870, 5.484236, 0.113850,
    0.628608, 0.201094, 0.747889, 2.924161, 0.171995, 0.164525, 0.315261, 0.621323,
    0.179771, 0.465271, 0.470140, 0.121827, 9.533943, 1.761439, 0.124066, 3.038533,
    0.593478, 0.211561, 0.408532, 1.143980, 0.239697, 0.165473,
)

# equilibrium frequencies (published order, renormalised below)
_JTT_PUBLISHED_FREQUENCIES = (
    0.076862, 0.051057, 0.042546, 0.051269, 0.020279, 0.041061, 0.061820, 0.074714, 0.022983, 0.052569,
    0.091111, 0.059498, 0.023414, 0.040530, 0.050532, 0.068225, 0.058518, 0.014336, 0.032303, 0.066374,
)


def _build_jtt():
    rates = np.zeros((NUM_AA, NUM_AA), dtype=np.float64)
    rates[np.tril_indices(NUM_AA, k=-1)] = _JTT_LOWER_TRIANGLE
    rates = np.maximum(rates, rates.T)
    freqs = np.array(_JTT_PUBLISHED_FREQUENCIES, dtype=np.float64)
    freqs = freqs / freqs.sum()
    states = np.array(list(_AA_PUBLISHED_ORDER))
    # PastML keeps states sorted by their one-letter code (JTTModel.py:58-61)
    order = np.argsort(states)
    return states[order], freqs[order], rates[:, order][order, :]


JTT_STATES, JTT_FREQUENCIES, JTT_RATE_MATRIX = _build_jtt()


class JTTModel(CustomRatesModel):
    """Fixed rates and frequencies; only sf (and tau) are free (JTTModel.py:64-95)."""

    def __init__(self, forest_stats, sf=None, tau=0, optimise_tau=False, parameter_file=None, reoptimise=False,
                 **kwargs):
        kwargs['states'] = JTT_STATES
        kwargs.pop('frequency_smoothing', None)
        CustomRatesModel.__init__(self, forest_stats=forest_stats, sf=sf, frequencies=JTT_FREQUENCIES,
                                  rate_matrix=JTT_RATE_MATRIX, parameter_file=parameter_file, reoptimise=reoptimise,
                                  frequency_smoothing=False, tau=tau, optimise_tau=optimise_tau, **kwargs)
        self._optimise_frequencies = False
        self.name = JTT

    @CustomRatesModel.states.setter
    def states(self, states):
        raise NotImplementedError('The JTT states are preset and cannot be changed.')

    def parse_parameters(self, params, reoptimise=False):
        # frequencies are part of the model: only sf / tau can be preset
        return Model.parse_parameters(self, params, reoptimise)
