"""
Host-side model objects: the parameter vector / bounds / (de)serialisation side of ``pastml/models/__init__.py``
(reference lines cited per method).  The per-branch arithmetic P(t) = exp(Q t) does *not* live here: every model
describes itself to the HIP library through :meth:`Model.kernel_spec` and :meth:`Model.get_Pij_t` calls the
``pml_pij`` entry point of the C-ABI.
"""
import logging
import os

import numpy as np
import pandas as pd

from pastml_amd import NUM_NODES, NUM_TIPS

MODEL = 'model'
CHANGES_PER_AVG_BRANCH = 'state_changes_per_avg_branch'
SCALING_FACTOR = 'scaling_factor'
SMOOTHING_FACTOR = 'smoothing_factor'
FREQUENCIES = 'frequencies'

# kinds understood by the HIP library (include/pastml_hip.h)
KIND_F81 = 0
KIND_HKY = 1
KIND_EIGEN = 2


def _read_param_table(params):
    """dict or path of a two-column tab file ('parameter', 'value') -> dict (reference: models/__init__.py:206-231)."""
    if params is None:
        return None
    if isinstance(params, dict):
        table = params
    elif isinstance(params, str):
        if not os.path.exists(params):
            raise ValueError('The specified parameter file ({}) does not exist.'.format(params))
        try:
            df = pd.read_csv(params, header=0, index_col=0, sep='\t')
            table = df['value'].to_dict()
        except Exception:
            raise ValueError('The specified parameter file {} is malformed, '
                             'should be a tab-delimited file with two columns, '
                             'the first one containing parameter names, '
                             'and the second, named "value", containing parameter values.'.format(params))
    else:
        raise ValueError('Parameters must be specified either as a dict or as a path to a csv file, not as {}!'
                         .format(type(params)))
    return {str(k.encode('ASCII', 'replace').decode()): v for (k, v) in table.items()}


class ScalarParameter(object):
    """
    One scalar of a model's optimiser vector, described once: the attribute that holds it, the flag that frees it, its
    bounds, and how it is read from / written to a parameter file.  ``Model`` walks tables of these instead of spelling
    out sf, tau (and, for HKY, kappa) in every method that touches the vector -- pastml/models/__init__.py:145-258 and
    HKYModel.py:84-189 do the latter.
    """

    def __init__(self, attr, flag, bounds, key, label, lowest_ok, strictly, pins_on_read=True, typo=False):
        self.attr = attr                # property through which optimised values are assigned (setters guard frozen ones)
        self.flag = flag                # name of the boolean attribute: is the scalar a free parameter?
        self.bounds = bounds            # model -> (low, high)
        self.key = key                  # row name in parameter files / dicts
        self.label = label              # how log messages call it
        self.lowest_ok = lowest_ok      # values at or below (strictly) / below this are rejected when read
        self.strictly = strictly
        self.pins_on_read = pins_on_read  # a value read from a file fixes the scalar unless reoptimise
        self.typo = typo                # the reference's kappa message says 'paramaters'; kept for log parity

    def is_free(self, model):
        return bool(getattr(model, self.flag))

    def read(self, model, params, reoptimise):
        if self.key not in params:
            return
        logger = logging.getLogger('pastml')
        raw = params[self.key]
        try:
            value = np.float64(raw)
        except (TypeError, ValueError):
            logger.error('{} ({}) given in parameters is not float, ignoring it.'.format(self.label, raw))
            return
        if value < self.lowest_ok or (self.strictly and value == self.lowest_ok):
            logger.error('{} cannot be negative, ignoring the value given in {} ({}).'
                         .format(self.label, 'paramaters' if self.typo else 'parameters', raw))
            return
        setattr(model, '_' + self.attr, value)
        if self.pins_on_read:
            setattr(model, self.flag, reoptimise)


def _brlen_scaled(low, high):
    return lambda model: (low / model.forest_stats.avg_nonzero_brlen, high / model.forest_stats.avg_nonzero_brlen)


SF_PARAMETER = ScalarParameter('sf', '_optimise_sf', _brlen_scaled(0.001, 10.), SCALING_FACTOR, 'Scaling factor',
                               0, True)
TAU_PARAMETER = ScalarParameter('tau', '_optimise_tau', lambda model: (0, model.forest_stats.avg_nonzero_brlen),
                                SMOOTHING_FACTOR, 'Smoothing factor', 0, False, pins_on_read=False)


class PointBlock(object):
    """
    The points of one batch as arrays -- frequencies [n, k], scaling factors, smoothing factors and their tau factors [n]
    -- for the models whose kernel description is (kind, pi): the batch goes to the engine's staging arrays as slices
    instead of n dictionaries and tuples (57 000 points per acr() over the HIV1C columns).  Reads like the list of
    (kernel description, rate parameters) it replaces: len(), indexing, iteration.
    """
    __slots__ = ('kind', 'pi', 'sf', 'tau', 'tf')

    def __init__(self, kind, pi, sf, tau, tf):
        self.kind, self.pi, self.sf, self.tau, self.tf = kind, pi, sf, tau, tf

    def __len__(self):
        return len(self.sf)

    def __getitem__(self, i):
        return dict(kind=self.kind, pi=self.pi[i]), (float(self.sf[i]), float(self.tau[i]), float(self.tf[i]))

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class Model(object):
    """
    Base model: scaling factor ``sf``, smoothing factor ``tau`` (reference: pastml/models/__init__.py:17-273).
    The branch transform is t' = (t + tau) * tau_factor * sf with tau_factor = L / (L + tau (N - 1)) (``:39-42``,
    ``:269-270``).
    """

    def __init__(self, states, forest_stats, sf=None, tau=0, optimise_tau=False,
                 parameter_file=None, reoptimise=False, character=None, **kwargs):
        self._name = None
        self._states = np.sort(states)
        self._forest_stats = forest_stats
        self._optimise_tau = optimise_tau
        self._optimise_sf = True
        self._sf = None
        self._tau = None
        self._character = character
        self._extra_params_fixed = False
        self._engine = None
        self.parse_parameters(parameter_file, reoptimise)
        if self._sf is None:
            self._sf = sf if sf is not None else 1. / forest_stats.avg_nonzero_brlen
        if self._tau is None:
            self._tau = tau if tau else 0
        self.calc_tau_factor()

    # ------------------------------------------------------------------ basic properties
    def calc_tau_factor(self):
        fs = self._forest_stats
        self._tau_factor = fs.forest_length / (fs.forest_length + self._tau * (fs.num_nodes - 1)) if self._tau else 1

    @property
    def forest_stats(self):
        return self._forest_stats

    @forest_stats.setter
    def forest_stats(self, forest_stats):
        self._forest_stats = forest_stats
        self.calc_tau_factor()

    @property
    def name(self):
        return self._name

    @name.setter
    def name(self, name):
        self._name = name

    @property
    def states(self):
        return self._states

    @states.setter
    def states(self, states):
        self._states = states

    @property
    def sf(self):
        return self._sf

    @sf.setter
    def sf(self, sf):
        if not self._optimise_sf:
            raise NotImplementedError('The scaling factor is preset and cannot be changed.')
        self._sf = sf

    @property
    def tau(self):
        return self._tau

    @tau.setter
    def tau(self, tau):
        if not self._optimise_tau:
            raise NotImplementedError('Tau is preset and cannot be changed.')
        self._tau = tau
        self.calc_tau_factor()

    def transform_t(self, t):
        return (t + self.tau) * self._tau_factor * self.sf

    # ------------------------------------------------------------------ optimiser interface (models/__init__.py:145-189)
    # x = [leading scalars that are free] [subclass block] [trailing scalars that are free]
    LEADING = (SF_PARAMETER, TAU_PARAMETER)
    TRAILING = ()

    def _free(self, table):
        return [p for p in table if p.is_free(self)]

    def get_num_params(self):
        return len(self._free(self.LEADING))

    def get_optimised_parameters(self):
        return np.array([getattr(self, p.attr) for p in self._free(self.LEADING)], dtype=np.float64)

    def set_params_from_optimised(self, ps, **kwargs):
        for value, p in zip(ps, self._free(self.LEADING)):
            setattr(self, p.attr, value)

    def get_bounds(self):
        return np.array([p.bounds(self) for p in self._free(self.LEADING)], np.float64)

    def freeze(self):
        for p in self.LEADING + self.TRAILING:
            setattr(self, p.flag, False)

    def basic_params_fixed(self):
        return not self._optimise_tau and not self._optimise_sf

    def extra_params_fixed(self):
        return self._extra_params_fixed

    def fix_extra_params(self):
        self._extra_params_fixed = True

    def unfix_extra_params(self):
        self._extra_params_fixed = False

    # ------------------------------------------------------------------ parameters in / out
    def parse_parameters(self, params, reoptimise=False):
        """
        Reads the scalars of the model's tables ('scaling_factor', 'smoothing_factor', ...) from a dict or a parameter
        file; given values are fixed unless ``reoptimise`` (reference: models/__init__.py:191-258).
        """
        params = _read_param_table(params)
        if params is None:
            return {}
        for p in self.LEADING:
            p.read(self, params, reoptimise)
        return params

    def save_parameters(self, filehandle):
        """Same rows, same order as the reference's parameter file (models/__init__.py:65-77)."""
        fs = self.forest_stats
        for key, value in ((MODEL, self.name), (NUM_NODES, fs.num_nodes), (NUM_TIPS, fs.num_tips),
                           (SCALING_FACTOR, self.sf), (CHANGES_PER_AVG_BRANCH, self.sf * fs.avg_nonzero_brlen),
                           (SMOOTHING_FACTOR, self.tau)):
            filehandle.write('{}\t{}\n'.format(key, value))

    def _print_basic_parameters(self):
        return '\tscaling factor:\t{:.6f}, i.e. {:.6f} changes per avg branch\t{}\n' \
               '\tsmoothing factor:\t{:.6f}\t{}\n' \
            .format(self.sf, self.forest_stats.avg_nonzero_brlen * self.sf,
                    '(optimised)' if self._optimise_sf else '(fixed)',
                    self.tau, '(optimised)' if self._optimise_tau else '(fixed)')

    def _print_parameters(self):
        return self._print_basic_parameters()

    def __str__(self):
        return 'Model {} for character {} with parameter values:\n{}' \
            .format(self.name, self._character, self._print_parameters())

    # ------------------------------------------------------------------ device side
    def rate_params(self):
        """(sf, tau, tau_factor) of the branch transform."""
        return float(self.sf), float(self.tau), float(self._tau_factor)

    def kernel_spec(self):
        """Description of P(t) for the HIP library: dict(kind=..., arrays...)."""
        raise NotImplementedError('Please implement this method in the Model subclass')

    def kernel_points(self, vectors):
        """
        (kernel description, rate parameters) of every optimiser vector of a batch -- the points of one finite-difference
        gradient go to the device together.  The model is left at the last vector, as after evaluating them in turn.
        """
        points = []
        for ps in vectors:
            self.set_params_from_optimised(ps)
            points.append((self.kernel_spec(), self.rate_params()))
        return points

    def get_Pij_t(self, t, *args, **kwargs):
        """
        Probability matrix of substitutions i->j over time t (k x k ndarray), computed by the HIP library
        (``pml_pij``); API of pastml/models/__init__.py:136-143.
        """
        from pastml_amd import hip
        return hip.pij(self, np.array([t], dtype=np.float64))[0]


class ModelWithFrequencies(Model):
    """
    Adds equilibrium frequencies, either optimised (k-1 ratios pi_i/pi_k in the parameter vector) or smoothed
    (one pseudo-count parameter) -- reference: pastml/models/__init__.py:276-455.
    """

    def __init__(self, states, forest_stats, sf=None, frequencies=None, tau=0,
                 optimise_tau=False, frequency_smoothing=False, parameter_file=None, reoptimise=False, **kwargs):
        self._frequencies = None
        self._optimise_frequencies = not frequency_smoothing
        self._frequency_smoothing = frequency_smoothing
        Model.__init__(self, states, forest_stats=forest_stats, sf=sf, tau=tau, optimise_tau=optimise_tau,
                       reoptimise=reoptimise, parameter_file=parameter_file, **kwargs)
        if self._frequencies is None:
            self._frequencies = frequencies if frequencies is not None \
                else np.ones(len(states), dtype=np.float64) / len(states)

    @property
    def frequencies(self):
        return self._frequencies

    @frequencies.setter
    def frequencies(self, frequencies):
        if not (self._optimise_frequencies or self._frequency_smoothing):
            raise NotImplementedError('The frequencies are preset and cannot be changed.')
        self._frequencies = frequencies

    def _n_frequency_params(self):
        if self._optimise_frequencies:
            return len(self.frequencies) - 1
        return 1 if self._frequency_smoothing else 0

    def get_num_params(self):
        return Model.get_num_params(self) + self._n_frequency_params() + len(self._free(self.TRAILING))

    def extra_params_fixed(self):
        return self._extra_params_fixed or Model.get_num_params(self) == self.get_num_params()

    def basic_params_fixed(self):
        return not Model.get_num_params(self)

    def set_params_from_optimised(self, ps, **kwargs):
        Model.set_params_from_optimised(self, ps, **kwargs)
        if self.extra_params_fixed():
            return
        at = Model.get_num_params(self)
        if self._optimise_frequencies:
            freqs = np.hstack((ps[at: at + len(self.frequencies) - 1], [1.]))
            self.frequencies = freqs / freqs.sum()
        elif self._frequency_smoothing:
            # NB (reference behaviour, models/__init__.py:331-335): smoothing is applied to the *current* frequencies
            freqs = self.frequencies * self.forest_stats.num_tips + ps[at]
            self.frequencies = freqs / freqs.sum()
        at += self._n_frequency_params()
        for value, p in zip(ps[at:], self._free(self.TRAILING)):
            setattr(self, p.attr, value)

    def get_optimised_parameters(self):
        basic = Model.get_optimised_parameters(self)
        if self.extra_params_fixed():
            return basic
        if self._optimise_frequencies:
            block = self.frequencies[:-1] / self.frequencies[-1]
        else:
            block = [0] if self._frequency_smoothing else []
        return np.hstack((basic, block, [getattr(self, p.attr) for p in self._free(self.TRAILING)]))

    def get_bounds(self):
        basic = Model.get_bounds(self)
        if self.extra_params_fixed():
            return basic
        rows = [np.array([1e-6, 10e6], np.float64)] * (len(self.frequencies) - 1 if self._optimise_frequencies else 0)
        if self._frequency_smoothing:
            rows.append(np.array([0, self.forest_stats.num_nodes]))
        rows += [np.array(p.bounds(self), np.float64) for p in self._free(self.TRAILING)]
        return np.array((*basic, *rows))

    def freeze(self):
        Model.freeze(self)
        self._optimise_frequencies = False
        self._frequency_smoothing = False

    def parse_parameters(self, params, reoptimise=False):
        """
        Besides the basic parameters, reads state frequencies keyed by state name
        (reference: models/__init__.py:365-414).  As in the reference, accepted frequencies are stored as given
        (the floor-and-renormalise step of ``:398-408`` computes a local value that is never assigned).
        """
        params = self._parse_frequencies(Model.parse_parameters(self, params, reoptimise), reoptimise)
        for p in self.TRAILING:
            p.read(self, params, reoptimise)
        return params

    def _parse_frequencies(self, params, reoptimise):
        logger = logging.getLogger('pastml')
        known = set(self.states) & set(params.keys())
        if not known:
            return params
        unknown = [state for state in self.states if state not in params.keys()]
        if unknown and not reoptimise:
            logger.error('Frequencies for some of the states ({}) are missing, '
                         'ignoring the specified frequencies.'.format(', '.join(unknown)))
            return params
        raw = np.array([params[state] if state in params.keys() else 0 for state in self.states])
        try:
            freqs = raw.astype(np.float64)
        except (TypeError, ValueError):
            logger.error('Could not convert the frequencies given in parameters ({}) to float, '
                         'ignoring them.'.format(raw))
            return params
        if np.round(freqs.sum() - 1, 2) != 0 and not reoptimise:
            logger.error('Frequencies given in parameters ({}) do not sum up to one ({}),'
                         'ignoring them.'.format(freqs, freqs.sum()))
        elif np.any(freqs < 0) and not reoptimise:
            logger.error('Some of the frequencies given in parameters ({}) are negative,'
                         'ignoring them.'.format(freqs))
        else:
            try:
                min_freq = min(1 / self.forest_stats.num_tips,
                               min(float(params[state]) for state in known if float(params[state]) > 0)) / 2
            except (TypeError, ValueError, AttributeError):
                logger.error('Could not convert the frequencies given in parameters ({}) to float, '
                             'ignoring them.'.format(raw))
                return params
            if unknown:
                logger.error('Frequencies for some of the states ({}) are missing from parameters, '
                             'setting them to {}.'.format(', '.join(unknown), min_freq))
            self._frequencies = freqs
            self._optimise_frequencies = reoptimise and not self._frequency_smoothing
        return params

    def _print_parameters(self):
        text = '{}\tfrequencies\t{}\n{}\n'.format(
            Model._print_parameters(self),
            '(optimised)' if self._optimise_frequencies else '(smoothed)' if self._frequency_smoothing else '(fixed)',
            '\n'.join('\t\t{}:\t{:g}'.format(state, freq) for (state, freq) in zip(self.states, self.frequencies)))
        for p in self.TRAILING:
            text += '\t{}\t{:.6f}\t{}\n'.format(p.key, getattr(self, p.attr),
                                                 '(optimised)' if p.is_free(self) else '(fixed)')
        return text

    def save_parameters(self, filehandle):
        Model.save_parameters(self, filehandle)
        for state, frequency in zip(self.states, self.frequencies):
            filehandle.write('{}\t{}\n'.format(state, frequency))
        for p in self.TRAILING:
            filehandle.write('{}\t{:g}\n'.format(p.key, getattr(self, p.attr)))
