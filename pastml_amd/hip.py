"""
ctypes binding of libpastml_hip.so (C-ABI: include/pastml_hip.h) and the :class:`Engine` that owns one device
context (= one tree + one batch of columns sharing k and the model kind).

There is no CPU fallback: if the shared library is missing or no MI355X is visible, every entry point raises
:class:`HipUnavailableError`.
"""
import collections
import ctypes
import os
import threading

import numpy as np

from pastml_amd.models import KIND_F81, KIND_HKY, KIND_EIGEN

# PASTML_HIP_LIBRARY: another build of the library for this process (A/B scripts compare builds without touching the
# in-tree file); unset: the library built in-tree by pastml_amd/build.py
_LIB_PATH = os.environ.get('PASTML_HIP_LIBRARY') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libpastml_hip.so')

PML_OK, PML_ERR_INVALID, PML_ERR_HIP, PML_ERR_UNSUPPORTED, PML_ZERO_LIKELIHOOD = 0, 1, 2, 3, 4

BUF_BU, BUF_BU_SF, BUF_TD, BUF_TD_SF, BUF_POSTERIOR, BUF_LH_SUM, BUF_LH_SF, BUF_JOINT_TABLE, BUF_JOINT_STATE, \
    BUF_BRANCH_EXP = range(10)

MAX_STATES = 512          # the F81 family; HKY / eigen models: MAX_STATES_MATRIX
MAX_STATES_MATRIX = 256
SCHEDULE_SINGLE_LAUNCH, SCHEDULE_BLOCKS, SCHEDULE_TWO_LEVEL, SCHEDULE_LEVELS, SCHEDULE_OTHER_MODEL = 0, 1, 2, 3, 4
OPT_CHERRY_FUSION = 1
OPT_KEEP_TD = 2
OPT_EIGEN_FUSED = 3
OPT_EIGEN_JOINT_VALU = 4
OPT_IMPLICIT_TIP_POSTERIORS = 5
COMM_ID_BYTES = 128
COMM_SUM, COMM_MAX = 0, 1


class HipUnavailableError(RuntimeError):
    pass


class HipError(RuntimeError):
    def __init__(self, status, message):
        RuntimeError.__init__(self, 'libpastml_hip: {} (status {})'.format(message, status))
        self.status = status


class ZeroLikelihoodError(HipError):
    """PML_ZERO_LIKELIHOOD: carries, per column, the (parent, child) node ids (or -1)."""

    def __init__(self, message, err_parent, err_child, loglik=None):
        HipError.__init__(self, PML_ZERO_LIKELIHOOD, message)
        self.err_parent = err_parent
        self.err_child = err_child
        self.loglik = loglik   # the log-likelihoods of all columns (the failing ones are not finite)


_c_int32_p = ctypes.POINTER(ctypes.c_int32)
_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_uint64_p = ctypes.POINTER(ctypes.c_uint64)
_ctx_p = ctypes.c_void_p

# name -> argument types (all functions return int unless listed in _RESTYPES)
SIGNATURES = {
    'pml_last_error': [],
    'pml_version': [],
    'pml_build_digest': [],
    'pml_device_count': [ctypes.POINTER(ctypes.c_int)],
    'pml_ctx_create': [ctypes.c_int, ctypes.POINTER(_ctx_p)],
    'pml_ctx_destroy': [_ctx_p],
    'pml_ctx_sync': [_ctx_p],
    'pml_ctx_set_option': [_ctx_p, ctypes.c_int, ctypes.c_int],
    'pml_ctx_set_tunable': [_ctx_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int],
    'pml_ctx_memory': [_ctx_p, _c_uint64_p, _c_uint64_p],
    'pml_schedule_info': [_ctx_p, _c_int32_p, _c_int32_p, _c_int32_p],
    'pml_sweep_schedule': [_ctx_p, _c_int32_p, _c_int32_p, _c_int32_p],
    'pml_tree_order': [_ctx_p, _c_int32_p],
    'pml_tree_upload': [_ctx_p, ctypes.c_int32, ctypes.c_int32, _c_int32_p, _c_int32_p, _c_int32_p, _c_double_p,
                        ctypes.c_int32, _c_int32_p, _c_int32_p, ctypes.c_int32, _c_int32_p, _c_int32_p, _c_int32_p,
                        _c_int32_p],
    'pml_chars_alloc': [_ctx_p, ctypes.c_int32, ctypes.c_int32],
    'pml_masks_upload': [_ctx_p, ctypes.c_int32, ctypes.c_int32, _c_uint64_p],
    'pml_masks_from_tip_states': [_ctx_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _c_int32_p, _c_int32_p],
    'pml_masks_initial_upload': [_ctx_p, ctypes.c_int32, ctypes.c_int32, _c_uint64_p],
    # (the hot calls take plain addresses -- c_void_p accepts an int -- of buffers the Engine keeps: building a ctypes
    # pointer object costs ~2 us, and an optimiser round is a handful of such calls)
    'pml_model_set_f81': [_ctx_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                          ctypes.c_void_p],
    'pml_model_set_hky': [_ctx_p, ctypes.c_int32, ctypes.c_int32, _c_double_p, _c_double_p, _c_double_p, _c_double_p,
                          _c_double_p],
    'pml_model_set_eigen': [_ctx_p, ctypes.c_int32, ctypes.c_int32, _c_double_p, _c_double_p, _c_double_p,
                            _c_double_p, _c_double_p, _c_double_p, _c_double_p],
    'pml_pij': [_ctx_p, ctypes.c_int32, ctypes.c_int32, _c_double_p, _c_double_p],
    'pml_pij_batch': [_ctx_p, _c_double_p],
    'pml_bottom_up': [_ctx_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    'pml_bottom_up_submit': [_ctx_p, ctypes.c_int],
    'pml_bottom_up_submit_columns': [_ctx_p, ctypes.c_int, ctypes.c_void_p],
    'pml_loglik_total': [_ctx_p, _c_double_p],
    'pml_bottom_up_collect': [_ctx_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    'pml_top_down_marginals': [_ctx_p, _c_double_p, _c_double_p, _c_double_p],
    'pml_marginal_pass': [_ctx_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _c_double_p, _c_double_p,
                          _c_double_p],
    'pml_joint_backtrace': [_ctx_p, _c_int32_p],
    'pml_joint_pass': [_ctx_p, _c_double_p, _c_int32_p, _c_int32_p, _c_int32_p],
    'pml_select_states': [_ctx_p, ctypes.c_int, ctypes.c_int, _c_uint64_p, _c_uint64_p, _c_int32_p],
    'pml_marginal_counts': [_ctx_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64, _c_double_p],
    'pml_marginal_counts_altered': [_ctx_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint8),
                                    _c_double_p, _c_int32_p, _c_int32_p],
    'pml_download': [_ctx_p, ctypes.c_int, ctypes.c_int32, ctypes.c_void_p],
    'pml_comm_unique_id': [ctypes.POINTER(ctypes.c_ubyte)],
    'pml_comm_init': [_ctx_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_ubyte)],
    'pml_comm_destroy': [_ctx_p],
    'pml_comm_info': [_ctx_p] + [ctypes.POINTER(ctypes.c_int32)] * 4,
    'pml_device_uuid': [ctypes.c_int, ctypes.c_char_p],
    'pml_comm_allreduce': [_ctx_p, _c_double_p, _c_double_p, ctypes.c_int32, ctypes.c_int],
    'pml_allreduce_loglik': [_ctx_p, _c_double_p, ctypes.c_int32, _c_double_p],
    'pml_device_sync': [ctypes.c_int],
    'pml_host_f81_fd_points': [ctypes.c_int32, ctypes.c_int32] + [ctypes.c_void_p] * 3 + [ctypes.c_int32] * 3 +
                              [ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_double, ctypes.c_double] +
                              [ctypes.c_void_p] * 5 + [ctypes.c_double],
    'pml_download_strided': [_ctx_p, ctypes.c_int, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                             ctypes.c_void_p],
    'pml_timer_start': [_ctx_p],
    'pml_timer_stop': [_ctx_p, ctypes.POINTER(ctypes.c_float)],
    'pml_profile_enable': [_ctx_p, ctypes.c_int],
    'pml_profile_read': [_ctx_p, ctypes.c_int, _c_double_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int],
}
_RESTYPES = {'pml_last_error': ctypes.c_char_p, 'pml_build_digest': ctypes.c_char_p}

_lib = None
_lib_lock = threading.Lock()


def library_path():
    return _LIB_PATH


def load_library():
    """Loads libpastml_hip.so (without touching the GPU) and declares the prototypes."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise HipUnavailableError(
                    '{} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                    '(hipcc --offload-arch=gfx950). There is no CPU fallback for the likelihood path.'.format(_LIB_PATH))
            lib = ctypes.CDLL(_LIB_PATH)
            for name, argtypes in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.argtypes = argtypes
                fn.restype = _RESTYPES.get(name, ctypes.c_int)
            _lib = lib
    return _lib


def device_uuid(device=0):
    """The GPU's UUID as 32 hex digits (pml_device_uuid): two ranks that print the same one share a device."""
    buf = ctypes.create_string_buffer(33)
    _check(load_library().pml_device_uuid(int(device), buf))
    return buf.value.decode()


def build_digest():
    """Digest of the sources the loaded library was compiled from (pml_build_digest; pastml_amd/build.py makes it)."""
    return load_library().pml_build_digest().decode()


def _check(status):
    if status != PML_OK:
        raise HipError(status, load_library().pml_last_error().decode())


_device_count = None


def device_count():
    """Visible HIP devices (asked once per process: the runtime's answer costs milliseconds)."""
    global _device_count
    if _device_count is None:
        n = ctypes.c_int(0)
        lib = load_library()
        if lib.pml_device_count(ctypes.byref(n)) != PML_OK:
            return 0
        _device_count = n.value
    return _device_count


def default_device():
    for var in ('PASTML_HIP_DEVICE', 'LOCAL_RANK'):
        if var in os.environ:
            return int(os.environ[var])
    return 0


def _as(arr, dtype):
    return np.ascontiguousarray(arr, dtype=dtype)


def _ptr(arr, ctype):
    return arr.ctypes.data_as(ctypes.POINTER(ctype))


def pack_masks(masks, k):
    """0/1 array [..., k] -> uint64 words [..., W] (bit s of word s // 64 = state s)."""
    masks = np.asarray(masks)
    W = (k + 63) // 64
    bits = np.packbits(masks.astype(bool), axis=-1, bitorder='little')
    pad = W * 8 - bits.shape[-1]
    if pad:
        bits = np.concatenate([bits, np.zeros(bits.shape[:-1] + (pad,), dtype=np.uint8)], axis=-1)
    return np.ascontiguousarray(bits).view('<u8').reshape(masks.shape[:-1] + (W,))


def unpack_masks(words, k):
    """uint64 words [..., W] -> 0/1 int8 array [..., k]."""
    b = np.ascontiguousarray(words).view(np.uint8).reshape(words.shape[:-1] + (words.shape[-1] * 8,))
    return np.unpackbits(b, axis=-1, bitorder='little')[..., :k].astype(np.int8)


# ----------------------------------------------------------------------------------------------------------------------
# Engine pool.  acr() analyses one character after the other on the same forest (pastml/acr.py:210-231), and every
# ml_acr needs a context for the single column plus one per width of the batched optimiser: creating and destroying
# them (device allocations, tree upload, descriptor tables, captured graphs) is a fifth of the wall time of a binary
# character on a 3 600-tip tree.  Released engines of small problems are kept, keyed by the forest's arrays, the number
# of columns, k and the device, and handed out again with their per-analysis state reset.
_POOL = collections.OrderedDict()   # key -> list of idle engines; least recently used key first
_POOL_LOCK = threading.Lock()
_POOL_MAX_BYTES = 1 << 30   # an engine holding more device memory than this is destroyed on release
_POOL_MAX_ENGINES = 16


def _forest_key(flat):
    """
    Identity of the forest's arrays.  The digest is cached on the FlatForest together with the arrays' write flags
    switched off, so an in-place edit of a pooled forest raises in numpy instead of silently reusing a ctx that
    holds the old tree.
    """
    key = getattr(flat, '_pool_key', None)
    if key is None:
        import hashlib
        h = hashlib.blake2b(digest_size=16)
        for a in (flat.parent, flat.first_child, flat.n_children, flat.dist):
            h.update(np.ascontiguousarray(a).tobytes())
        key = (flat.n_nodes, h.hexdigest())
        try:
            flat._pool_key = key
            for a in (flat.parent, flat.first_child, flat.n_children, flat.dist):
                if isinstance(a, np.ndarray):
                    a.setflags(write=False)
        except (AttributeError, ValueError):
            pass
    return key


def acquire_engine(flat, n_cols, k, device=None):
    """An Engine for (flat, n_cols, k): a pooled one if there is one (masks, models and '.initial' masks are the
    caller's to set, as for a new one), else a new one.  Give it back with release_engine()."""
    dev = default_device() if device is None else device
    key = (_forest_key(flat), n_cols, k, dev)
    with _POOL_LOCK:
        free = _POOL.get(key)
        eng = None
        if free:
            eng = free.pop()
            if free:
                _POOL.move_to_end(key)
            else:
                del _POOL[key]
    if eng is not None:
        eng.flat = flat
        eng.set_initial_masks(None)
        eng.set_option(OPT_KEEP_TD, 0)
        return eng
    eng = Engine(flat, n_cols, k, device=device)
    eng._pool_key = key
    return eng


def release_engine(eng):
    """Returns an engine of acquire_engine() to the pool; a full pool evicts its least recently used engine."""
    key = getattr(eng, '_pool_key', None)
    if key is None or eng._ctx.value is None:
        eng.close()
        return
    try:
        eng.sync()
        held, _ = eng.memory()
    except HipError:
        eng.close()
        return
    if held > _POOL_MAX_BYTES:
        eng.close()
        return
    evicted = []
    with _POOL_LOCK:
        _POOL.setdefault(key, []).append(eng)
        _POOL.move_to_end(key)
        while sum(len(v) for v in _POOL.values()) > _POOL_MAX_ENGINES:
            oldest = next(iter(_POOL))
            evicted.append(_POOL[oldest].pop(0))
            if not _POOL[oldest]:
                del _POOL[oldest]
    for e in evicted:
        e.close()


def drain_engine_pool():
    with _POOL_LOCK:
        engines = [e for v in _POOL.values() for e in v]
        _POOL.clear()
    for e in engines:
        e.close()


import atexit  # noqa: E402
atexit.register(drain_engine_pool)


class BareContext(object):
    """A device context without a tree: stream, options, communicator (what Engine builds on)."""

    def __init__(self, device=None):
        lib = load_library()
        if device_count() < 1:
            raise HipUnavailableError('no HIP device visible: the likelihood path needs an MI355X (gfx950)')
        self._lib = lib
        self._ctx = _ctx_p()
        self.device = default_device() if device is None else device
        _check(lib.pml_ctx_create(self.device, ctypes.byref(self._ctx)))

    def close(self):
        if getattr(self, '_ctx', None) is not None and self._ctx.value is not None:
            self._lib.pml_ctx_destroy(self._ctx)
            self._ctx = _ctx_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        _check(self._lib.pml_ctx_sync(self._ctx))

    def set_option(self, option, value):
        _check(self._lib.pml_ctx_set_option(self._ctx, option, 1 if value else 0))

    def set_tunable(self, name, value):
        """One switch of the schedules for this context (value None: not given; see pml_ctx_set_tunable)."""
        _check(self._lib.pml_ctx_set_tunable(self._ctx, str(name).encode(), 0 if value is None else int(value),
                                             0 if value is None else 1))

    def memory(self):
        """(bytes of device memory held by this context, bytes free on its device)."""
        held, free = ctypes.c_uint64(0), ctypes.c_uint64(0)
        _check(self._lib.pml_ctx_memory(self._ctx, ctypes.byref(held), ctypes.byref(free)))
        return held.value, free.value

    def schedule_info(self):
        """(level schedule with two-level / stacked units in use, two-level nodes, stacked nodes) of the F81-family sweeps."""
        a, b, c = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        _check(self._lib.pml_schedule_info(self._ctx, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return bool(a.value), b.value, c.value

    def sweep_schedule(self):
        """(schedule of the F81 marginal sweeps -- SCHEDULE_SINGLE_LAUNCH / _BLOCKS / _TWO_LEVEL / _LEVELS / _OTHER_MODEL,
        number of subtree blocks, stored nodes absorbed by the general two-level units)."""
        a, b, c = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        _check(self._lib.pml_sweep_schedule(self._ctx, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def node_order(self):
        """new_of_old: the library's id of every node of the forest as it was uploaded (pml_tree_order); the identity unless
        pml_tree_upload renumbered a ragged forest into height order."""
        out = np.empty(self.n_nodes, dtype=np.int32)
        _check(self._lib.pml_tree_order(self._ctx, _ptr(out, ctypes.c_int32)))
        return out

    def comm_init(self, rank, world, unique_id=None):
        """Attaches a communicator (RCCL for world > 1; unique_id: the 128 bytes of rank 0's comm_unique_id())."""
        buf = None
        if unique_id is not None:
            if len(unique_id) != COMM_ID_BYTES:
                raise ValueError('unique_id must be {} bytes'.format(COMM_ID_BYTES))
            buf = (ctypes.c_ubyte * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        _check(self._lib.pml_comm_init(self._ctx, rank, world, buf))
        self._comm_attached = True
        self._total_fresh = False

    def comm_info(self):
        """dict(rank, world, backend ('local' / 'rccl'), rccl_ranks = ncclCommCount of the communicator) -- pml_comm_info."""
        v = [ctypes.c_int32(0) for _ in range(4)]
        _check(self._lib.pml_comm_info(self._ctx, *[ctypes.byref(x) for x in v]))
        return dict(rank=v[0].value, world=v[1].value, backend='rccl' if v[2].value else 'local', rccl_ranks=v[3].value)

    def comm_destroy(self):
        _check(self._lib.pml_comm_destroy(self._ctx))
        self._comm_attached = False
        self._total_fresh = False

    def loglik_total(self):
        """
        Sum over all ranks of the log-likelihoods of the last marginal_pass, reduced on the device behind the sweeps
        (pml_loglik_total); None if no marginal pass has run on this context since the last call.
        """
        if not getattr(self, '_total_fresh', False):
            return None
        total = ctypes.c_double(0)
        _check(self._lib.pml_loglik_total(self._ctx, ctypes.byref(total)))
        self._total_fresh = False
        return total.value

    def allreduce(self, values, op=COMM_SUM):
        """Sum (or max) over the ranks of a float array; returns a new array."""
        a = _as(np.atleast_1d(values), np.float64)
        out = np.empty_like(a)
        _check(self._lib.pml_comm_allreduce(self._ctx, _ptr(a, ctypes.c_double), _ptr(out, ctypes.c_double), len(a), op))
        return out

    def allreduce_loglik(self, loglik):
        """Sum over all ranks and all of their columns of the log-likelihoods (one 8-byte all-reduce)."""
        a = _as(np.atleast_1d(loglik), np.float64)
        total = ctypes.c_double(0)
        _check(self._lib.pml_allreduce_loglik(self._ctx, _ptr(a, ctypes.c_double), len(a), ctypes.byref(total)))
        return total.value


class Engine(BareContext):
    """
    One device context: a flat forest, ``n_cols`` columns with ``k`` states each, one model kind.

    >>> eng = Engine(flat, n_cols=1, k=5)
    >>> eng.set_masks(masks)                 # [n_cols, N, k] 0/1
    >>> eng.set_models([model])              # objects with kernel_spec() / rate_params(), or (spec, rates) tuples
    >>> lnl = eng.bottom_up()                # [n_cols]
    >>> post, lh_sum, lh_sf = eng.top_down_marginals()
    """

    def __init__(self, flat, n_cols, k, device=None, cherry_fusion=True, keep_td=False, tune=None):
        """
        tune: {switch name: value or None} -- the schedules' switches for THIS engine (pml_ctx_set_tunable; names as the
        PASTML_HIP_* environment variables, with or without the prefix; None = back to the built-in default whatever the
        environment says).  Set before the tree is uploaded, so the switches that shape the schedules take effect.
        """
        BareContext.__init__(self, device)
        lib = self._lib
        for name, value in (tune or {}).items():
            self.set_tunable(name, value)
        if not cherry_fusion:
            _check(lib.pml_ctx_set_option(self._ctx, OPT_CHERRY_FUSION, 0))
        if keep_td:
            _check(lib.pml_ctx_set_option(self._ctx, OPT_KEEP_TD, 1))  # kept across the tree upload below
        self.flat = flat
        self.n_nodes = flat.n_nodes
        self.n_cols = n_cols
        self.k = k
        self.kind = None
        # buffers (and their addresses) of the calls an optimiser repeats hundreds of times
        self._lnl = np.empty(n_cols, dtype=np.float64)
        self._ep = np.empty(n_cols, dtype=np.int32)
        self._ec = np.empty(n_cols, dtype=np.int32)
        self._out_addr = (self._lnl.ctypes.data, self._ep.ctypes.data, self._ec.ctypes.data)
        self._par_pi = np.empty((n_cols, k), dtype=np.float64)
        self._par = np.empty((3, n_cols), dtype=np.float64)   # sf, tau, tau factor
        self._par_addr = (self._par_pi.ctypes.data, self._par[0].ctypes.data, self._par[1].ctypes.data,
                          self._par[2].ctypes.data)
        i32 = ctypes.c_int32
        arrays = dict(parent=_as(flat.parent, np.int32), first_child=_as(flat.first_child, np.int32),
                      n_children=_as(flat.n_children, np.int32), dist=_as(flat.dist, np.float64),
                      bu_offsets=_as(flat.bu_offsets, np.int32), bu_order=_as(flat.bu_order, np.int32),
                      td_offsets=_as(flat.td_offsets, np.int32),
                      td_parent_offsets=_as(flat.td_parent_offsets, np.int32),
                      td_parents=_as(flat.td_parents, np.int32), post_rank=_as(flat.post_rank, np.int32))
        try:
            _check(lib.pml_tree_upload(
                self._ctx, flat.n_nodes, len(flat.roots), _ptr(arrays['parent'], i32), _ptr(arrays['first_child'], i32),
                _ptr(arrays['n_children'], i32), _ptr(arrays['dist'], ctypes.c_double),
                flat.n_bu_levels, _ptr(arrays['bu_offsets'], i32), _ptr(arrays['bu_order'], i32),
                flat.n_td_levels, _ptr(arrays['td_offsets'], i32), _ptr(arrays['td_parent_offsets'], i32),
                _ptr(arrays['td_parents'], i32), _ptr(arrays['post_rank'], i32)))
            _check(lib.pml_chars_alloc(self._ctx, n_cols, k))
        except Exception:
            self.close()
            raise

    # ------------------------------------------------------------------------------------------------------------------
    # ------------------------------------------------------------------------------------------------------------------
    def set_masks(self, masks, col_begin=0):
        """masks: 0/1 array [n, N, k] (or [N, k] for one column)."""
        masks = np.asarray(masks)
        if masks.ndim == 2:
            masks = masks[None]
        if masks.shape[1:] != (self.n_nodes, self.k):
            raise ValueError('masks must be [cols, {}, {}], got {}'.format(self.n_nodes, self.k, masks.shape))
        words = pack_masks(masks, self.k)
        _check(self._lib.pml_masks_upload(self._ctx, col_begin, col_begin + len(masks), _ptr(words, ctypes.c_uint64)))

    def set_mask_words(self, words, col_begin=0):
        """Packed masks: uint64 [n, N, W] (bit s of word s // 64 = state s)."""
        words = _as(words, np.uint64)
        if words.ndim == 2:
            words = words[None]
        if words.shape[1:] != (self.n_nodes, (self.k + 63) // 64):
            raise ValueError('mask words must be [cols, {}, {}], got {}'.format(self.n_nodes, (self.k + 63) // 64,
                                                                               words.shape))
        _check(self._lib.pml_masks_upload(self._ctx, col_begin, col_begin + len(words), _ptr(words, ctypes.c_uint64)))

    def set_initial_mask_words(self, words, col_begin=0):
        words = _as(words, np.uint64)
        if words.ndim == 2:
            words = words[None]
        _check(self._lib.pml_masks_initial_upload(self._ctx, col_begin, col_begin + len(words),
                                                  _ptr(words, ctypes.c_uint64)))

    def set_initial_masks(self, masks, col_begin=0):
        """Masks before zero-branch alteration (joint sweeps only); None clears them."""
        if masks is None:
            _check(self._lib.pml_masks_initial_upload(self._ctx, 0, self.n_cols, None))
            return
        masks = np.asarray(masks)
        if masks.ndim == 2:
            masks = masks[None]
        words = pack_masks(masks, self.k)
        _check(self._lib.pml_masks_initial_upload(self._ctx, col_begin, col_begin + len(masks),
                                                  _ptr(words, ctypes.c_uint64)))

    def set_tip_states(self, states, col_begin=0):
        """states: int [n, n_tips] state index of every tip of ``flat.tips`` (-1 = missing); internal nodes free."""
        states = _as(states, np.int32)
        if states.ndim == 1:
            states = states[None]
        tips = _as(self.flat.tips, np.int32)
        if states.shape[1] != len(tips):
            raise ValueError('expected {} tip states per column'.format(len(tips)))
        _check(self._lib.pml_masks_from_tip_states(self._ctx, col_begin, col_begin + len(states), len(tips),
                                                   _ptr(tips, ctypes.c_int32), _ptr(states, ctypes.c_int32)))

    # ------------------------------------------------------------------------------------------------------------------
    def stage_f81(self, col_begin, pi, sf, tau, tf):
        """F81 family: parameters of columns col_begin .. into the staging arrays (arrays [n, k], [n], [n], [n]); commit_f81
        hands a range of them to the library."""
        ce = col_begin + len(sf)
        if col_begin < 0 or ce > self.n_cols or pi.shape != (len(sf), self.k):
            raise ValueError('parameter block of {} columns x {} states does not fit at column {}'.format(len(sf), pi.shape[-1],
                                                                                                         col_begin))
        self._par_pi[col_begin:ce] = pi
        self._par[0, col_begin:ce] = sf
        self._par[1, col_begin:ce] = tau
        self._par[2, col_begin:ce] = tf

    def commit_f81(self, col_begin, col_end):
        a_pi, a_sf, a_tau, a_tf = self._par_addr
        _check(self._lib.pml_model_set_f81(self._ctx, col_begin, col_end, a_pi + col_begin * self.k * 8, a_sf + col_begin * 8,
                                           a_tau + col_begin * 8, a_tf + col_begin * 8))
        self.kind = KIND_F81

    def set_models(self, models, col_begin=0):
        """models: list (one per column) of Model objects or (spec dict, (sf, tau, tau_factor)) tuples."""
        specs, rates = [], []
        for m in models:
            if type(m) is tuple:
                spec, r = m
            else:
                spec, r = m.kernel_spec(), m.rate_params()
            specs.append(spec)
            rates.append(r)
        kind = specs[0]['kind']
        for s in specs:
            if s['kind'] != kind:
                raise ValueError('all columns of an Engine must use the same model kind')
        n = len(specs)
        if kind == KIND_F81 and 0 <= col_begin and col_begin + n <= self.n_cols:
            # straight into the engine's own staging arrays (the library copies them into its pinned mirror)
            cb, ce = col_begin, col_begin + n
            try:
                if n == 1:
                    self._par_pi[cb] = specs[0]['pi']
                else:
                    self._par_pi[cb:ce] = [s['pi'] for s in specs]
            except ValueError:
                raise ValueError('frequencies must have {} entries'.format(self.k))
            if n == 1:
                self._par[0, cb], self._par[1, cb], self._par[2, cb] = rates[0]
            else:
                self._par[:, cb:ce] = np.array(rates, dtype=np.float64).T
            a_pi, a_sf, a_tau, a_tf = self._par_addr
            _check(self._lib.pml_model_set_f81(self._ctx, cb, ce, a_pi + cb * self.k * 8, a_sf + cb * 8, a_tau + cb * 8,
                                               a_tf + cb * 8))
            self.kind = kind
            return
        dbl = ctypes.c_double
        pi = _as(np.stack([s['pi'] for s in specs]), np.float64)
        if pi.shape != (n, self.k):
            raise ValueError('frequencies must have {} entries'.format(self.k))
        sf = _as([r[0] for r in rates], np.float64)
        tau = _as([r[1] for r in rates], np.float64)
        tf = _as([r[2] for r in rates], np.float64)
        cb, ce = col_begin, col_begin + n
        if kind == KIND_F81:
            _check(self._lib.pml_model_set_f81(self._ctx, cb, ce, pi.ctypes.data, sf.ctypes.data, tau.ctypes.data,
                                               tf.ctypes.data))
        elif kind == KIND_HKY:
            kappa = _as([s['kappa'] for s in specs], np.float64)
            _check(self._lib.pml_model_set_hky(self._ctx, cb, ce, _ptr(pi, dbl), _ptr(kappa, dbl), _ptr(sf, dbl),
                                               _ptr(tau, dbl), _ptr(tf, dbl)))
        elif kind == KIND_EIGEN:
            for s in specs:
                if np.iscomplexobj(s['d']) or np.iscomplexobj(s['A']):
                    raise ValueError('complex eigen-decomposition: the rate matrix is not reversible')
            d = _as(np.stack([s['d'] for s in specs]), np.float64)
            A = _as(np.stack([s['A'] for s in specs]), np.float64)
            Ainv = _as(np.stack([s['Ainv'] for s in specs]), np.float64)
            _check(self._lib.pml_model_set_eigen(self._ctx, cb, ce, _ptr(pi, dbl), _ptr(d, dbl), _ptr(A, dbl),
                                                 _ptr(Ainv, dbl), _ptr(sf, dbl), _ptr(tau, dbl), _ptr(tf, dbl)))
        else:
            raise ValueError('unknown model kind {}'.format(kind))
        self.kind = kind

    # ------------------------------------------------------------------------------------------------------------------
    def pij(self, ts, col=0):
        ts = _as(ts, np.float64)
        out = np.empty((len(ts), self.k, self.k), dtype=np.float64)
        _check(self._lib.pml_pij(self._ctx, col, len(ts), _ptr(ts, ctypes.c_double), _ptr(out, ctypes.c_double)))
        return out

    def pij_batch(self, copy_out=False):
        out = None
        if copy_out:
            out = np.empty((self.n_cols, self.n_nodes, self.k, self.k), dtype=np.float64)
        _check(self._lib.pml_pij_batch(self._ctx, None if out is None else _ptr(out, ctypes.c_double)))
        return out

    def _sweep_results(self, status):
        # copies of the engine's own result buffers (callers keep what they get)
        if status == PML_ZERO_LIKELIHOOD:
            raise ZeroLikelihoodError(self._lib.pml_last_error().decode(), self._ep.copy(), self._ec.copy(),
                                      self._lnl.copy())
        _check(status)
        return self._lnl.copy()

    def bottom_up(self, is_marginal=True):
        return self._sweep_results(self._lib.pml_bottom_up(self._ctx, 1 if is_marginal else 0, *self._out_addr))

    def bottom_up_submit(self, is_marginal=True, active=None):
        """Puts a bottom-up sweep on the context's stream and returns at once (see bottom_up_collect).  active: uint8
        array of n_cols, 0 = the column sits this sweep out (its parameters and masks are those of the last sweep that
        computed it, and bottom_up_collect returns that sweep's value for it); None = all columns."""
        if active is None:
            _check(self._lib.pml_bottom_up_submit(self._ctx, 1 if is_marginal else 0))
            return
        active = np.ascontiguousarray(active, dtype=np.uint8)
        if active.shape != (self.n_cols,):
            raise ValueError('active must have one entry per column')
        _check(self._lib.pml_bottom_up_submit_columns(self._ctx, 1 if is_marginal else 0, active.ctypes.data))

    def bottom_up_collect(self, is_marginal=True):
        """Waits for the sweep of bottom_up_submit; returns / raises what bottom_up would have."""
        return self._sweep_results(self._lib.pml_bottom_up_collect(self._ctx, 1 if is_marginal else 0, *self._out_addr))

    def top_down_marginals(self, posterior=True, lh=True):
        CN = (self.n_cols, self.n_nodes)
        post = np.empty(CN + (self.k,), dtype=np.float64) if posterior else None
        lh_sum = np.empty(CN, dtype=np.float64) if lh else None
        lh_sf = np.empty(CN, dtype=np.float64) if lh else None
        dbl = ctypes.c_double
        _check(self._lib.pml_top_down_marginals(self._ctx, None if post is None else _ptr(post, dbl),
                                                None if lh_sum is None else _ptr(lh_sum, dbl),
                                                None if lh_sf is None else _ptr(lh_sf, dbl)))
        return post, lh_sum, lh_sf

    def marginal_pass(self, posterior=True, lh=True):
        """bottom_up(True) + top_down_marginals() with one host round trip: (lnl, posterior, lh_sum, lh_sf)."""
        CN = (self.n_cols, self.n_nodes)
        post = np.empty(CN + (self.k,), dtype=np.float64) if posterior else None
        lh_sum = np.empty(CN, dtype=np.float64) if lh else None
        lh_sf = np.empty(CN, dtype=np.float64) if lh else None
        dbl = ctypes.c_double
        status = self._lib.pml_marginal_pass(self._ctx, *self._out_addr,
                                             None if post is None else _ptr(post, dbl),
                                             None if lh_sum is None else _ptr(lh_sum, dbl),
                                             None if lh_sf is None else _ptr(lh_sf, dbl))
        self._total_fresh = getattr(self, '_comm_attached', False)   # (the collective is on the stream whatever the status)
        return self._sweep_results(status), post, lh_sum, lh_sf

    def joint_backtrace(self, copy_out=True):
        out = np.empty((self.n_cols, self.n_nodes), dtype=np.int32) if copy_out else None
        _check(self._lib.pml_joint_backtrace(self._ctx, None if out is None else _ptr(out, ctypes.c_int32)))
        return out

    def joint_pass(self, copy_out=True):
        """bottom_up(False) + joint_backtrace() with one host round trip: (lnl, joint states or None)."""
        lnl = np.empty(self.n_cols, dtype=np.float64)
        ep = np.empty(self.n_cols, dtype=np.int32)
        ec = np.empty(self.n_cols, dtype=np.int32)
        out = np.empty((self.n_cols, self.n_nodes), dtype=np.int32) if copy_out else None
        i32 = ctypes.c_int32
        status = self._lib.pml_joint_pass(self._ctx, _ptr(lnl, ctypes.c_double), _ptr(ep, i32), _ptr(ec, i32),
                                          None if out is None else _ptr(out, i32))
        if status == PML_ZERO_LIKELIHOOD:
            raise ZeroLikelihoodError(self._lib.pml_last_error().decode(), ep, ec, lnl)
        _check(status)
        return lnl, out

    def select_states(self, method, force_joint=False, lh_masks=None, packed=False):
        """
        MAP ('MAP') or MPPA ('MPPA') selection on the device from the last marginals; lh_masks: optional 0/1 array
        [n_cols, N, k] (or, with packed=True, uint64 words [n_cols, N, W]) multiplied into the marginal likelihoods
        first.  The selected masks become the columns' masks.
        Returns (masks [n_cols, N, k] int8 -- packed: the words [n_cols, N, W] --, n_states [n_cols, N]).
        """
        W = (self.k + 63) // 64
        words = np.empty((self.n_cols, self.n_nodes, W), dtype=np.uint64)
        nsel = np.empty((self.n_cols, self.n_nodes), dtype=np.int32)
        if lh_masks is None:
            lm = None
        elif packed:
            lm = _as(lh_masks, np.uint64).reshape(self.n_cols, self.n_nodes, W)
        else:
            lm = pack_masks(np.asarray(lh_masks).reshape(self.n_cols, self.n_nodes, self.k), self.k)
        _check(self._lib.pml_select_states(self._ctx, {'MAP': 0, 'MPPA': 1}[method], 1 if force_joint else 0,
                                           None if lm is None else _ptr(lm, ctypes.c_uint64),
                                           _ptr(words, ctypes.c_uint64), _ptr(nsel, ctypes.c_int32)))
        return (words if packed else unpack_masks(words, self.k)), nsel

    def marginal_counts(self, n_repetitions, seed, col=0):
        """
        k x k expected numbers of state changes per scenario from n_repetitions scenarios sampled on the device
        (pastml/ml.py:753-862) after a marginal pass of column col; seed keys the counter-based generator.
        """
        out = np.empty((self.k, self.k), dtype=np.float64)
        _check(self._lib.pml_marginal_counts(self._ctx, col, int(n_repetitions), ctypes.c_uint64(int(seed)),
                                             _ptr(out, ctypes.c_double)))
        return out

    def marginal_counts_altered(self, n_repetitions, seed, altered, col=0):
        """
        The device's part of marginal_counts on a forest with altered nodes (pml_marginal_counts_altered): altered is a 0/1 array
        [N].  Returns (sums [k, k] float -- not divided by n_repetitions --, state counts [N, k] int32, same-state draws [N, k]
        int32); pastml_amd.ml.marginal_counts adds the fractional counts of the pairs with an altered end.
        """
        alt = np.ascontiguousarray(altered, dtype=np.uint8)
        if alt.shape != (self.n_nodes,):
            raise ValueError('one flag per node expected')
        sums = np.empty((self.k, self.k), dtype=np.float64)
        counts = np.empty((self.n_nodes, self.k), dtype=np.int32)
        same = np.empty((self.n_nodes, self.k), dtype=np.int32)
        _check(self._lib.pml_marginal_counts_altered(self._ctx, col, int(n_repetitions), ctypes.c_uint64(int(seed)),
                                                     _ptr(alt, ctypes.c_uint8), _ptr(sums, ctypes.c_double),
                                                     _ptr(counts, ctypes.c_int32), _ptr(same, ctypes.c_int32)))
        return sums, counts, same

    def download(self, what, col=0):
        N, k = self.n_nodes, self.k
        if what in (BUF_BU, BUF_TD, BUF_POSTERIOR):
            out = np.empty((N, k), dtype=np.float64)
        elif what == BUF_JOINT_TABLE:
            out = np.empty((N, k), dtype=np.int32)
        elif what == BUF_JOINT_STATE:
            out = np.empty(N, dtype=np.int32)
        else:
            out = np.empty(N, dtype=np.float64)
        _check(self._lib.pml_download(self._ctx, what, col, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def download_strided(self, what, col=0, first=0, stride=1, count=None):
        """Rows first, first + stride, ... of one column's posterior / LH_SUM / LH_SF / joint-state buffer."""
        if count is None:
            count = (self.n_nodes - first + stride - 1) // stride
        if what == BUF_POSTERIOR:
            out = np.empty((count, self.k), dtype=np.float64)
        elif what == BUF_JOINT_STATE:
            out = np.empty(count, dtype=np.int32)
        else:
            out = np.empty(count, dtype=np.float64)
        _check(self._lib.pml_download_strided(self._ctx, what, col, first, stride, count,
                                              out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def profile_enable(self, on=True):
        _check(self._lib.pml_profile_enable(self._ctx, 1 if on else 0))

    def profile_read(self, which, reset=False):
        """(total milliseconds, launches) of slot which: 0 bottom-up / 1 top-down level kernels, 2 P(t)/prep kernels,
        3 / 4 the top-down / bottom-up launch of the two-level units."""
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        _check(self._lib.pml_profile_read(self._ctx, which, ctypes.byref(ms), ctypes.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def timer_start(self):
        _check(self._lib.pml_timer_start(self._ctx))

    def timer_stop(self):
        ms = ctypes.c_float(0)
        _check(self._lib.pml_timer_stop(self._ctx, ctypes.byref(ms)))
        return ms.value


# ---------------------------------------------------------------------------------------------------------------------
def comm_unique_id():
    """128 opaque bytes made by rank 0 and handed to every rank's Engine.comm_init (needs librccl)."""
    buf = (ctypes.c_ubyte * COMM_ID_BYTES)()
    _check(load_library().pml_comm_unique_id(buf))
    return bytes(buf)


def device_sync(device=None):
    _check(load_library().pml_device_sync(default_device() if device is None else device))


_PIJ_FOREST = None


def pij(model, ts):
    """P(t) for a Model object and an array of branch lengths, through pml_pij (used by Model.get_Pij_t).  The
    one-node context it runs on comes from the engine pool: per-branch callers do not pay a context each."""
    global _PIJ_FOREST
    from pastml_amd.tree import FlatForest
    if _PIJ_FOREST is None:
        _PIJ_FOREST = FlatForest([-1], [0], [1], [0.0], [0])
    eng = acquire_engine(_PIJ_FOREST, 1, len(model.states))
    try:
        eng.set_models([model])
        return eng.pij(ts)
    finally:
        release_engine(eng)
