"""F81-like model for an arbitrary number of states (reference: pastml/models/F81Model.py)."""
import numpy as np

from pastml_amd.models import ModelWithFrequencies, KIND_F81

F81 = 'F81'


class F81Model(ModelWithFrequencies):
    """
    P_ij(t) = pi_j (1 - exp(-mu t')) + [i == j] exp(-mu t'),  mu = 1 / (1 - sum_i pi_i^2)
    (reference: F81Model.py:18-46).  On the device P is never materialised for this family: the sweeps use
    P v = (1 - e)(pi . v) 1 + e v with e = exp(-mu t') precomputed per branch.
    """

    def __init__(self, states, forest_stats, sf=None, frequencies=None, tau=0,
                 frequency_smoothing=False, optimise_tau=False, parameter_file=None, reoptimise=False, **kwargs):
        ModelWithFrequencies.__init__(self, states=states, forest_stats=forest_stats, sf=sf, tau=tau,
                                      frequencies=frequencies, optimise_tau=optimise_tau,
                                      frequency_smoothing=frequency_smoothing, reoptimise=reoptimise,
                                      parameter_file=parameter_file, **kwargs)
        self.name = F81

    def get_mu(self):
        """mu = 1 / (1 - sum_i pi_i^2), so that the expected rate -mu * trace(Pi Q) is one (F81Model.py:18-26)."""
        return 1. / (1. - self.frequencies.dot(self.frequencies))

    def kernel_spec(self):
        pi = np.ascontiguousarray(self.frequencies, dtype=np.float64)
        with np.errstate(divide='ignore'):
            mu = np.float64(1.) / (np.float64(1.) - pi.dot(pi))
        return dict(kind=KIND_F81, pi=pi, mu=float(mu))
