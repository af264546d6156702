"""
Closed-form model families: F81 / JC / EFT (one exponential per branch, pastml/models/F81Model.py, JCModel.py,
EFTModel.py) and HKY85 (pastml/models/HKYModel.py).  The classes carry parameters, bounds and (de)serialisation; the
per-branch arithmetic runs on the device (kernel_spec() describes the model to libpastml_hip).
"""
import numpy as np

from pastml_amd.models import Model, ModelWithFrequencies, PointBlock, ScalarParameter, KIND_F81, KIND_HKY


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

    def kernel_points(self, vectors):
        # All vectors of a batch decoded at once (the arithmetic of set_params_from_optimised, row by row: scalars
        # first, then pi = (ratios, 1) / their sum): a character's gradient is k + 1 points per optimiser step, and
        # decoding them one by one through the property setters was a quarter of the host time of a sweep round.
        X = np.asarray(vectors, dtype=np.float64)
        extra_fixed = self.extra_params_fixed()
        if X.ndim != 2 or not len(X) or self.TRAILING or (self._frequency_smoothing and not extra_fixed):
            return ModelWithFrequencies.kernel_points(self, vectors)
        n = len(X)
        at = 0
        sf = np.full(n, self.sf, dtype=np.float64)
        tau = np.full(n, self.tau, dtype=np.float64)
        if self._optimise_sf:
            sf, at = X[:, at], at + 1
        if self._optimise_tau:
            tau, at = X[:, at], at + 1
        free_pi = not extra_fixed and self._optimise_frequencies
        if not free_pi:
            pi = np.repeat(np.ascontiguousarray(self.frequencies, dtype=np.float64)[None, :], n, axis=0)
        else:
            ratios = np.hstack((X[:, at: at + len(self.frequencies) - 1], np.ones((n, 1))))
            pi = ratios / ratios.sum(axis=1)[:, None]
        fs = self._forest_stats
        tau = np.asarray(tau, dtype=np.float64)
        factor = np.ones(n, dtype=np.float64)
        smoothed = tau != 0
        if smoothed.any():
            factor[smoothed] = fs.forest_length / (fs.forest_length + tau[smoothed] * (fs.num_nodes - 1))
        # the model is left at the last vector, as set_params_from_optimised(X[-1]) would leave it (the same values: the
        # rows above are its arithmetic)
        if self._optimise_sf:
            self._sf = X[-1, 0]
        if self._optimise_tau:
            self._tau = tau[-1]
            self.calc_tau_factor()
        if free_pi:
            self._frequencies = pi[-1].copy()
        return PointBlock(KIND_F81, np.ascontiguousarray(pi), np.array(sf, dtype=np.float64), tau, factor)


    def fd_block(self, ps, lower, upper, step=1e-8):
        """
        The points of one forward-difference gradient around the optimiser vector ``ps`` (scipy's 2-point scheme with the
        absolute step ``step``: 1e-8 is scipy's), decoded: (PointBlock of len(ps) + 1 points -- ps itself first --, steps to divide by), or None
        when the library's helper does not take the case (a step that leaves the bounds, smoothed frequencies, ...) and the
        caller goes through two_point_scheme + kernel_points.  One call into libpastml_hip (pml_host_f81_fd_points, host
        arithmetic only) instead of two dozen small numpy operations per optimiser round; the same numbers, bit for bit
        (tests/test_host_logic.py).  The model is left at the last point, as kernel_points leaves it.
        """
        n = len(ps)
        work = self.__dict__.get('_fd_work')
        key = (n, self._extra_params_fixed, self._optimise_sf, self._optimise_tau, self._optimise_frequencies,
               self._frequency_smoothing)
        if work is None or work[0] != key:
            # (per search stage: the layout, the buffers and the call's constant arguments)
            extra_fixed = self.extra_params_fixed()
            if self.TRAILING or (self._frequency_smoothing and not extra_fixed):
                return None
            from pastml_amd import hip
            k = len(self._frequencies)
            free_pi = not extra_fixed and self._optimise_frequencies
            pi = np.empty((n + 1, k), dtype=np.float64)
            rows = np.empty((3, n + 1), dtype=np.float64)
            steps = np.empty(max(n, 1), dtype=np.float64)
            fs = self._forest_stats
            work = self.__dict__['_fd_work'] = (key, pi, rows, steps[:n], free_pi, hip.load_library().pml_host_f81_fd_points,
                                                (1 if self._optimise_sf else 0, 1 if self._optimise_tau else 0,
                                                 1 if free_pi else 0),
                                                (float(fs.forest_length), float(fs.num_nodes), pi.ctypes.data,
                                                 rows[0].ctypes.data, rows[1].ctypes.data, rows[2].ctypes.data, steps.ctypes.data),
                                                PointBlock(KIND_F81, pi, rows[0], rows[1], rows[2]))
        _, pi, rows, steps, free_pi, call, flags, tail, block = work
        x = ps if ps.dtype == np.float64 and ps.flags.c_contiguous else np.ascontiguousarray(ps, dtype=np.float64)
        fixed = self._frequencies
        if fixed.dtype != np.float64 or not fixed.flags.c_contiguous:
            fixed = np.ascontiguousarray(fixed, dtype=np.float64)
        if call(n, pi.shape[1], x.ctypes.data, lower.ctypes.data, upper.ctypes.data, *flags, float(self._sf), float(self._tau),
                fixed.ctypes.data, *tail, float(step)) != 0:
            return None
        if flags[0]:
            self._sf = rows[0, -1]
        if flags[1]:
            self._tau = rows[1, -1]
            self.calc_tau_factor()
        if free_pi:
            self._frequencies = pi[-1].copy()
        return block, steps


JC = 'JC'


class JCModel(F81Model):

    def __init__(self, states, forest_stats, sf=None, tau=0, optimise_tau=False, parameter_file=None,
                 reoptimise=False, **kwargs):
        kwargs['frequency_smoothing'] = False
        F81Model.__init__(self, states=states, forest_stats=forest_stats, sf=sf, tau=tau, optimise_tau=optimise_tau,
                          frequencies=np.ones(len(states), dtype=np.float64) / len(states),
                          reoptimise=reoptimise, parameter_file=parameter_file, **kwargs)
        self._optimise_frequencies = False
        self.name = JC

    def parse_parameters(self, params, reoptimise=False):
        # only sf / tau can be preset: frequencies are equal by definition (JCModel.py:28-40)
        return Model.parse_parameters(self, params, reoptimise)

    def _print_parameters(self):
        return '{}\tfrequencies\tall equal to {:g}\t(fixed)\n'.format(Model._print_parameters(self),
                                                                    1 / len(self.states))


EFT = 'EFT'


class EFTModel(F81Model):

    def __init__(self, states, forest_stats, observed_frequencies, sf=None, tau=0, optimise_tau=False,
                 parameter_file=None, reoptimise=False, **kwargs):
        F81Model.__init__(self, states=states, forest_stats=forest_stats, sf=sf, tau=tau,
                          optimise_tau=optimise_tau, frequencies=observed_frequencies,
                          reoptimise=reoptimise, parameter_file=parameter_file, **kwargs)
        self._optimise_frequencies = False
        self._frequency_smoothing = False
        self.name = EFT

    def parse_parameters(self, params, reoptimise=False):
        # only sf / tau can be preset: frequencies are the observed ones (EFTModel.py:27-40)
        return Model.parse_parameters(self, params, reoptimise)

    def _print_parameters(self):
        return '{}\tfrequencies:\tobserved in the tree\t(fixed)\n'.format(Model._print_parameters(self))


HKY = 'HKY'
HKY_STATES = np.array(['A', 'C', 'G', 'T'])
A, C, G, T = 0, 1, 2, 3
KAPPA = 'kappa'

# kappa rides behind the frequency block of the optimiser vector; bounds HKYModel.py:121-130
KAPPA_PARAMETER = ScalarParameter('kappa', '_optimise_kappa', lambda model: (1e-6, 20.), KAPPA, 'Kappa', 0, True,
                                  typo=True)


class HKYModel(ModelWithFrequencies):
    """
    Four states A, C, G, T; parameters: frequencies and the transition/transversion ratio kappa.  The closed-form P(t)
    (HKYModel.py:44-82) is evaluated per branch by the HIP library.  Everything the reference spells out per method for
    kappa (HKYModel.py:84-189: vector length, packing / unpacking, bounds, file row, freezing, printing) follows from
    one entry in the model's table of trailing scalars (pastml_amd/models/__init__.py, ScalarParameter).
    """
    TRAILING = (KAPPA_PARAMETER,)

    def __init__(self, forest_stats, sf=None, frequencies=None, kappa=4, tau=0,
                 frequency_smoothing=False, optimise_tau=False, parameter_file=None, reoptimise=False, **kwargs):
        self._kappa = None           # a parameter file read by the base constructor may set both
        self._optimise_kappa = True
        kwargs['states'] = HKY_STATES
        ModelWithFrequencies.__init__(self, forest_stats=forest_stats, sf=sf, tau=tau, optimise_tau=optimise_tau,
                                      frequencies=frequencies, frequency_smoothing=frequency_smoothing,
                                      reoptimise=reoptimise, parameter_file=parameter_file, **kwargs)
        if self._kappa is None:
            self._kappa = kappa
        self.name = HKY

    @property
    def kappa(self):
        return self._kappa

    @kappa.setter
    def kappa(self, kappa):
        if not self._optimise_kappa:
            raise NotImplementedError('The kappa value is preset and cannot be changed.')
        self._kappa = kappa

    @Model.states.setter
    def states(self, states):
        raise NotImplementedError("The HKY model is only implemented for nucleotides: "
                                  "the states are A, C, G, T and cannot be reset")

    def kernel_spec(self):
        return dict(kind=KIND_HKY, pi=np.ascontiguousarray(self.frequencies, dtype=np.float64),
                    kappa=float(self.kappa))
