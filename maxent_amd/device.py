"""ctypes binding of libmaxent_hip.so (include/maxent_hip.h).

This is the only place where Python crosses into native code.  There is no
CPU fallback: if the shared library is missing or no gfx950 device is
visible, every entry point raises :class:`MaxEntDeviceError`.
"""

import ctypes
import weakref
import os

import numpy as np

_LIB = None
_LIB_PATH = os.environ.get(
    'MAXENT_AMD_LIB',
    os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib',
                 'libmaxent_hip.so'))

ENTROPY_NORMAL = 0
ENTROPY_PLUSMINUS = 1
PRECISION_F64 = 0
PRECISION_F32 = 1


class MaxEntDeviceError(RuntimeError):
    pass


class MxeOpts(ctypes.Structure):
    """mirror of ``struct mxe_opts`` (include/maxent_hip.h)."""
    _fields_ = [('maxiter', ctypes.c_int32),
                ('miniter', ctypes.c_int32),
                ('tol_h', ctypes.c_double),
                ('tol_d', ctypes.c_double),
                ('tol_relq', ctypes.c_double),
                ('step_max', ctypes.c_double),
                ('mu_first', ctypes.c_double),
                ('mu_grow', ctypes.c_double),
                ('mu_max', ctypes.c_double),
                ('decouple_tol', ctypes.c_double),
                ('waves_per_chain', ctypes.c_int32),
                ('chains_per_wg', ctypes.c_int32),
                ('alpha_split', ctypes.c_int32),
                ('stop_estimate', ctypes.c_int32),
                ('precision', ctypes.c_int32),
                ('wg_per_cu', ctypes.c_int32),
                ('chi2_factor', ctypes.c_double),
                ('lds_basis', ctypes.c_int32),
                ('in_flight', ctypes.c_int32)]


_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_lp = ctypes.POINTER(ctypes.c_int64)
_vp = ctypes.c_void_p

# every symbol include/maxent_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ('mxe_version', ctypes.c_char_p, []),
    ('mxe_source_hash', ctypes.c_char_p, []),
    ('mxe_host_alloc', ctypes.c_void_p, [ctypes.c_size_t]),
    ('mxe_host_free', None, [ctypes.c_void_p]),
    ('mxe_strerror', ctypes.c_char_p, [ctypes.c_int]),
    ('mxe_device_count', ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    ('mxe_opts_default', None, [ctypes.POINTER(MxeOpts)]),
    ('mxe_ctx_create', ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, _dp, _dp, _dp,
                                      ctypes.POINTER(_vp)]),
    ('mxe_ctx_destroy', None, [_vp]),
    ('mxe_last_hip_error', ctypes.c_char_p, [_vp]),
    ('mxe_dataset_add', ctypes.c_int, [_vp, ctypes.c_int, _dp, _dp,
                                       ctypes.POINTER(ctypes.c_int)]),
    ('mxe_dataset_clear', ctypes.c_int, [_vp]),
    ('mxe_elements_set', ctypes.c_int, [_vp, ctypes.c_int, _ip, _dp, _lp, _dp,
                                        _ip]),
    ('mxe_elements_update_data', ctypes.c_int, [_vp, ctypes.c_int, _dp, _lp]),
    ('mxe_solve_chains', ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _ip,
                                        _dp, _dp, ctypes.POINTER(MxeOpts),
                                        _dp, _dp, _dp, _dp, _dp, _ip, _ip,
                                        _ip]),
    ('mxe_chains_upload', ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _ip,
                                         _dp, _dp, ctypes.POINTER(MxeOpts)]),
    ('mxe_chains_launch', ctypes.c_int, [_vp]),
    ('mxe_sync', ctypes.c_int, [_vp]),
    ('mxe_logdet', ctypes.c_int, [_vp, _dp]),
    ('mxe_chains_fetch_nact', ctypes.c_int, [_vp, _ip]),
    ('mxe_chains_finish', ctypes.c_int, [_vp, _ip]),
    ('mxe_chains_fetch', ctypes.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, _ip, _ip,
                                        _ip]),
    ('mxe_result_device_ptrs', ctypes.c_int, [_vp] + [ctypes.POINTER(_vp)] * 7),
    ('mxe_ns_padded', ctypes.c_int, [_vp]),
    ('mxe_set_result_buffer', ctypes.c_int, [_vp, ctypes.c_int]),
    ('mxe_last_kernel_ms', ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float)]),
    ('mxe_last_kernel_name', ctypes.c_char_p, [_vp]),
    ('mxe_timing_mark', ctypes.c_int, [_vp]),
    ('mxe_ms_since_mark', ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float)]),
    ('mxe_stream', ctypes.c_void_p, [_vp]),
    ('mxe_last_launch_info', ctypes.c_int, [_vp] +
     [ctypes.POINTER(ctypes.c_int)] * 3),
    ('mxe_apply_output_map', ctypes.c_int, [_vp, _dp, _dp]),
    ('mxe_launch_depth', ctypes.c_int, [_vp, _ip, _dp]),
    ('mxe_schedule_info', ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    ('mxe_eval_batch', ctypes.c_int, [_vp, ctypes.c_int, _ip, _dp, _dp, ctypes.c_int, ctypes.c_double] + [_dp] * 11),
    ('mxe_entropy', ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp, _dp]),
    ('mxe_audit', ctypes.c_int, [_vp, _dp, _dp]),
    ('mxe_select_launch', ctypes.c_int, [_vp, ctypes.c_int]),
    ('mxe_select_fetch', ctypes.c_int, [_vp, _ip, _dp]),
    ('mxe_select3_launch', ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double]),
    ('mxe_select3_fetch', ctypes.c_int, [_vp, _ip, _dp]),
    ('mxe_select3_fetch_rows', ctypes.c_int, [_vp, _ip, ctypes.c_int, ctypes.c_int, _dp]),
    ('mxe_select3_prefetch_rows', ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _dp]),
    ('mxe_chains_prefetch', ctypes.c_int, [_vp, _dp, _ip]),
    ('mxe_fetch_rows', ctypes.c_int, [_vp, ctypes.c_int, _ip, _dp]),
    ('mxe_shard_plan', ctypes.c_int, [ctypes.c_int, ctypes.c_int, _ip, _ip, _ip]),
    ('mxe_comm_unique_id', ctypes.c_int, [ctypes.c_char_p]),
    ('mxe_comm_init', ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]),
    ('mxe_comm_init_local', ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int]),
    ('mxe_comm_destroy', ctypes.c_int, [_vp]),
    ('mxe_comm_set_loopback', ctypes.c_int, [_vp, ctypes.c_int]),
    ('mxe_gather', ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _lp, _dp]),
    ('mxe_gather_local', ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int, ctypes.c_int, _lp, _dp]),
    ('mxe_comm_allreduce', ctypes.c_int, [_vp, _dp, ctypes.c_int, ctypes.c_int]),
    ('mxe_kernel_svd', ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp,
                                      ctypes.c_double, ctypes.c_int, _dp, ctypes.c_double,
                                      ctypes.c_int, _dp, _dp, _dp, _dp, _ip, _ip,
                                      ctypes.POINTER(ctypes.c_float)]),
]


def library_path():
    return _LIB_PATH


def load_library():
    """Load libmaxent_hip.so and declare every prototype. Raises if missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_LIB_PATH):
        raise MaxEntDeviceError(
            'HIP library not built: {} is missing. Run '
            '`python -c "import __graft_entry__ as g; g.build()"` or '
            '`make -C maxent_amd/csrc`. There is no CPU fallback.'.format(
                _LIB_PATH))
    lib = ctypes.CDLL(_LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)      # AttributeError if a symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    _LIB = lib
    return lib


def comm_unique_id():
    """128 bytes that rank 0 hands to the other ranks before ``DeviceContext.comm_init`` (ncclGetUniqueId)"""
    lib = load_library()
    buf = ctypes.create_string_buffer(128)
    rc = lib.mxe_comm_unique_id(buf)
    if rc != 0:
        raise MaxEntDeviceError('mxe_comm_unique_id failed: ' + lib.mxe_strerror(rc).decode())
    return buf.raw


def shard_plan(n_elem, n_ranks):
    """``mxe_shard_plan``: (rank of every element, its index inside the rank's shard, shard sizes);
    host arithmetic only, works without a GPU"""
    lib = load_library()
    rk = np.zeros(max(n_elem, 1), dtype=np.int32)
    li = np.zeros(max(n_elem, 1), dtype=np.int32)
    nl = np.zeros(n_ranks, dtype=np.int32)
    rc = lib.mxe_shard_plan(int(n_elem), int(n_ranks), _p(rk), _p(li), _p(nl))
    if rc != 0:
        raise MaxEntDeviceError('mxe_shard_plan failed: ' + lib.mxe_strerror(rc).decode())
    return rk[:n_elem], li[:n_elem], nl


def comm_init_local(contexts):
    """ranks of one process: rank = position in ``contexts`` (``mxe_comm_init_local``)"""
    lib = load_library()
    arr = (_vp * len(contexts))(*[c._h for c in contexts])
    rc = lib.mxe_comm_init_local(arr, len(contexts))
    if rc != 0:
        raise MaxEntDeviceError('mxe_comm_init_local failed: ' + lib.mxe_strerror(rc).decode())


def gather_local(contexts, root, counts, full=False, recv=None):
    lib = load_library()
    arr = (_vp * len(contexts))(*[c._h for c in contexts])
    counts = _c(counts, np.int64)
    rc = lib.mxe_gather_local(arr, len(contexts), int(root), 1 if full else 0, _p(counts), _p(recv))
    if rc != 0:
        msg = lib.mxe_strerror(rc).decode()
        if rc == -2:
            msg += ': ' + lib.mxe_last_hip_error(contexts[root]._h).decode()
        raise MaxEntDeviceError('mxe_gather_local failed: ' + msg)
    return recv


def source_hash():
    """hash of the sources the loaded library was built from (mxe_source_hash)"""
    return load_library().mxe_source_hash().decode()


def device_count():
    lib = load_library()
    n = ctypes.c_int(0)
    lib.mxe_device_count(ctypes.byref(n))
    return n.value


def default_opts(**kw):
    lib = load_library()
    o = MxeOpts()
    lib.mxe_opts_default(ctypes.byref(o))
    for k, val in kw.items():
        if not hasattr(o, k):
            raise TypeError('unknown solver option {!r}'.format(k))
        setattr(o, k, val)
    return o


def pinned_empty(shape, dtype=np.float64, min_bytes=1 << 20):
    """an uninitialised array in page-locked host memory (``mxe_host_alloc``; the block goes back to the library's pool with
    the last view of it) -- the destination of the large device-to-host copies; plain ``np.empty`` for arrays below
    ``min_bytes`` or when the runtime cannot pin that much"""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if n < min_bytes or n == 0:
        return np.empty(shape, dtype=dtype)
    if n < (1 << 20) + (1 << 21) and _small_pinned[0] + n > SMALL_PINNED_LIMIT:
        return np.empty(shape, dtype=dtype)              # (results a caller keeps hold their blocks: see SMALL_PINNED_LIMIT)
    lib = load_library()
    p = lib.mxe_host_alloc(n)
    if not p:
        return np.empty(shape, dtype=dtype)
    buf = (ctypes.c_char * n).from_address(p)
    weakref.finalize(buf, _host_free, lib, p, n)         # (every view keeps ``buf`` alive through its base)
    if n < (1 << 20) + (1 << 21):
        _small_pinned[0] += n
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


# The scalars and rows of a launch (DeviceContext.result_arrays(pinned_rows=...): ~2 MB per 16 x 16 x 100 job) live in page-locked
# memory as long as the result that holds them; a caller who keeps hundreds of results would pin gigabytes.  Beyond this many
# bytes of such blocks alive, further ones are ordinary memory (and are then copied out when the result is waited for, as before).
SMALL_PINNED_LIMIT = 256 << 20
_small_pinned = [0]


def _host_free(lib, p, n):
    if n < (1 << 20) + (1 << 21):
        _small_pinned[0] -= n
    lib.mxe_host_free(p)


def is_pinned(a):
    """whether ``a`` lies in a block of :func:`pinned_empty`"""
    b = a
    while getattr(b, 'base', None) is not None:
        b = b.base
    return isinstance(b, ctypes.Array)


def _c(a, dtype=np.float64):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    if a is None:
        return None
    if a.dtype == np.float64:
        return a.ctypes.data_as(_dp)
    if a.dtype == np.int32:
        return a.ctypes.data_as(_ip)
    if a.dtype == np.int64:
        return a.ctypes.data_as(_lp)
    raise TypeError(a.dtype)


def kernel_svd(tau, omega, delta, beta, preblur_b=(0.0,), threshold=1.e-14,
               ns_max=128, want_K=False, device=0):
    """``mxe_kernel_svd``: TauKernel (and PreblurKernel, one per entry of
    ``preblur_b`` > 0) filled and decomposed on the device.  Returns a list of
    dicts ``U, S, V`` (truncated at ``S >= threshold``), ``K`` (if wanted),
    ``qr_rank``, ``sweeps`` and the device time ``ms`` of the whole batch."""
    lib = load_library()
    if device_count() < 1:
        raise MaxEntDeviceError('no HIP device visible; the device SVD has no CPU fallback')
    tau, omega, delta = _c(tau), _c(omega), _c(delta)
    bs = _c(np.atleast_1d(np.asarray(preblur_b, dtype=float)))
    n_tau, n_w, n_b = len(tau), len(omega), len(bs)
    K = np.empty((n_b, n_tau, n_w)) if want_K else None
    U = np.empty((n_b, n_tau, ns_max))
    S = np.empty((n_b, ns_max))
    V = np.empty((n_b, n_w, ns_max))
    ns = np.zeros(n_b, dtype=np.int32)
    info = np.zeros((n_b, 3), dtype=np.int32)
    ms = ctypes.c_float(0)
    rc = lib.mxe_kernel_svd(int(device), n_tau, n_w, _p(tau), _p(omega), _p(delta), float(beta),
                            n_b, _p(bs), float(threshold), int(ns_max), _p(K), _p(U), _p(S), _p(V),
                            _p(ns), _p(info), ctypes.byref(ms))
    if rc != 0:
        raise MaxEntDeviceError('mxe_kernel_svd failed: ' + lib.mxe_strerror(rc).decode())
    out = []
    for ib in range(n_b):
        k = int(ns[ib])
        out.append(dict(U=U[ib, :, :k].copy(), S=S[ib, :k].copy(), V=V[ib, :, :k].copy(),
                        K=(K[ib] if want_K else None), qr_rank=int(info[ib, 0]),
                        sweeps=int(info[ib, 1]), ms=float(ms.value)))
    return out


def entropy(kind, H, D, device=0):
    """``mxe_entropy``: S, dS/dH and diag(d2S/dH2) of hidden images given directly; ``H``: (P, n) or (n,)."""
    lib = load_library()
    if device_count() < 1:
        raise MaxEntDeviceError('no HIP device visible; the entropy functions have no CPU fallback')
    H2 = _c(np.atleast_2d(H))
    D = _c(D)
    P, n = H2.shape
    S, dS, ddS = np.empty(P), np.empty((P, n)), np.empty((P, n))
    rc = lib.mxe_entropy(int(device), int(kind), n, P, _p(H2), _p(D), _p(S), _p(dS), _p(ddS))
    if rc != 0:
        raise MaxEntDeviceError('mxe_entropy failed: ' + lib.mxe_strerror(rc).decode())
    if np.ndim(H) == 1:
        return S[0], dS[0], ddS[0]
    return S, dS, ddS


class DeviceContext(object):
    """One solver context on one GPU: holds the truncated SVD of the kernel.

    Parameters mirror ``mxe_ctx_create``: ``U`` (n_tau x n_s), ``S`` (n_s),
    ``V`` (n_omega x n_s) as ``KernelSVD.U/.S/.V`` give them after
    ``reduce_singular_space`` (reference kernels.py:53-122).
    """

    def __init__(self, U, S, V, device=0, keep=None):
        """``keep``: stage only the first ``keep`` singular directions on the device (the caller has made sure that the others
        cannot be told from zero in its job: :func:`maxent_amd.batch_solver.directions_to_keep`).  The context still speaks
        ``n_s`` directions to its caller: start vectors are cut, returned v are filled up with zeros."""
        self._lib = load_library()
        self._h = _vp(None)
        S = _c(S)
        V = _c(V)
        self.n_s = int(S.shape[0])
        self.n_omega = int(V.shape[0])
        if V.shape[1] != self.n_s:
            raise ValueError('V must be n_omega x n_s')
        if U is not None:
            U = _c(U)
            if U.shape[1] != self.n_s:
                raise ValueError('U must be n_tau x n_s')
            self.n_tau = int(U.shape[0])
        else:
            self.n_tau = 1
        self._n_s_dev = self.n_s
        if keep is not None and int(keep) < self.n_s:
            if U is None or int(keep) < 1:
                raise ValueError('keep needs the unrotated U and at least one direction')
            self._n_s_dev = int(keep)
            U, S, V = _c(U[:, :self._n_s_dev]), _c(S[:self._n_s_dev]), _c(V[:, :self._n_s_dev])
        if device_count() < 1:
            raise MaxEntDeviceError('no HIP device visible; the solver has no '
                                    'CPU fallback')
        self._check(self._lib.mxe_ctx_create(
            int(device), self.n_tau, self.n_omega, self._n_s_dev,
            _p(U), _p(S), _p(V), ctypes.byref(self._h)), 'mxe_ctx_create')
        self.device = int(device)
        self._n_chain = 0
        self._n_alpha = 0
        self._ds_rows = []

    # -- plumbing ------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.mxe_strerror(rc).decode()
            if rc == -2 and self._h:
                msg += ': ' + self._lib.mxe_last_hip_error(self._h).decode()
            raise MaxEntDeviceError('{} failed: {}'.format(what, msg))

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self._lib.mxe_ctx_destroy(self._h)
            self._h = _vp(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data sets and elements ------------------------------------------
    def add_dataset(self, err, U_rot=None):
        """(U, err) pair -> whitened basis; returns the data-set id."""
        if U_rot is not None:
            U_rot = _c(U_rot)
            if self._n_s_dev < self.n_s:
                U_rot = _c(U_rot[:, :self._n_s_dev])
            n_rows = U_rot.shape[0]
        else:
            n_rows = self.n_tau
        err = _c(np.asarray(err, dtype=float) * np.ones(n_rows))
        i = ctypes.c_int(-1)
        self._check(self._lib.mxe_dataset_add(self._h, int(n_rows), _p(U_rot),
                                              _p(err), ctypes.byref(i)),
                    'mxe_dataset_add')
        self._ds_rows.append(int(n_rows))
        return i.value

    def clear_datasets(self):
        self._check(self._lib.mxe_dataset_clear(self._h), 'mxe_dataset_clear')
        self._ds_rows = []

    def set_elements(self, dataset_of_elem, G_list, D, entropy):
        """G_list: one data vector per element (already in its data set's
        rotated space); D: n_elem x n_omega (including delta-omega)."""
        ds = _c(dataset_of_elem, np.int32)
        n_elem = len(ds)
        if isinstance(G_list, np.ndarray) and G_list.ndim == 2:
            # one row per element, all of one length
            rows = np.asarray(self._ds_rows)[ds]
            if G_list.shape[0] != n_elem or np.any(rows != G_list.shape[1]):
                raise ValueError('G has shape {}, the data sets of its {} elements have {} rows'.format(
                    G_list.shape, n_elem, sorted(set(rows.tolist()))))
            G = _c(G_list)
            offs = np.arange(n_elem, dtype=np.int64) * G_list.shape[1]
            D = _c(D).reshape(n_elem, self.n_omega)
            ent = _c(entropy, np.int32)
            self._check(self._lib.mxe_elements_set(self._h, n_elem, _p(ds), _p(G), _p(offs), _p(D), _p(ent)),
                        'mxe_elements_set')
            self.n_elem = n_elem
            return
        offs = np.zeros(n_elem, dtype=np.int64)
        chunks = []
        pos = 0
        for e in range(n_elem):
            g = _c(G_list[e]).ravel()
            if g.shape[0] != self._ds_rows[ds[e]]:
                raise ValueError('G of element {} has length {}, its data set '
                                 'has {} rows'.format(e, g.shape[0],
                                                      self._ds_rows[ds[e]]))
            offs[e] = pos
            pos += g.shape[0]
            chunks.append(g)
        G = _c(np.concatenate(chunks))
        D = _c(D).reshape(n_elem, self.n_omega)
        ent = _c(entropy, np.int32)
        self._check(self._lib.mxe_elements_set(self._h, n_elem, _p(ds), _p(G),
                                               _p(offs), _p(D), _p(ent)),
                    'mxe_elements_set')
        self.n_elem = n_elem

    def update_data(self, G):
        """``mxe_elements_update_data``: new data vectors (one row per element, all of one length) for the elements that
        are set; data sets, default models, entropies and the staged chains stay"""
        G = _c(G)
        if G.ndim != 2 or G.shape[0] != self.n_elem:
            raise ValueError('G has shape {}, {} elements are set'.format(G.shape, self.n_elem))
        offs = np.arange(G.shape[0], dtype=np.int64) * G.shape[1]
        self._check(self._lib.mxe_elements_update_data(self._h, G.shape[0], _p(G), _p(offs)), 'mxe_elements_update_data')

    # -- the hot path ----------------------------------------------------
    def upload_chains(self, elem_of_chain, alpha_scaled, v0, opts=None):
        el = _c(elem_of_chain, np.int32)
        al = _c(alpha_scaled)
        if al.ndim == 1:
            al = np.ascontiguousarray(np.broadcast_to(al, (len(el), al.shape[0])))
        n_chain, n_alpha = al.shape
        v0 = _c(v0).reshape(n_chain, self.n_s)
        if self._n_s_dev < self.n_s:
            v0 = _c(v0[:, :self._n_s_dev])
        if opts is None:
            opts = default_opts()
        self._check(self._lib.mxe_chains_upload(self._h, n_chain, n_alpha,
                                                _p(el), _p(al), _p(v0),
                                                ctypes.byref(opts)),
                    'mxe_chains_upload')
        self._n_chain, self._n_alpha = n_chain, n_alpha

    def launch(self):
        self._check(self._lib.mxe_chains_launch(self._h), 'mxe_chains_launch')

    def sync(self):
        self._check(self._lib.mxe_sync(self._h), 'mxe_sync')

    def finish(self):
        """``mxe_chains_finish``: the alphas the lock-step layout gave up on, solved again in the one-chain
        layout (blocking); returns how many that was"""
        n = ctypes.c_int32(0)
        self._check(self._lib.mxe_chains_finish(self._h, ctypes.byref(n)), 'mxe_chains_finish')
        return int(n.value)

    def result_arrays(self, pinned_rows=None):
        """the per-alpha arrays :meth:`fetch` fills, uninitialised: chi2 / S / Q and n_iter / converged / n_evals as the
        rows of ONE block each -- the library then brings each block in one copy.  ``pinned_rows`` = n: both blocks and
        ``_rows`` [n][n_chain][n_omega] (the destination of :meth:`select3_prefetch_rows`) in one page-locked allocation,
        for :meth:`prefetch`"""
        nc, na = self._n_chain, self._n_alpha
        if pinned_rows is None:
            d = np.empty((3, nc, na))
            i = np.empty((3, nc, na), dtype=np.int32)
            return dict(chi2=d[0], S=d[1], Q=d[2], n_iter=i[0], converged=i[1], n_evals=i[2])
        nd, ni, nr = 3 * nc * na * 8, 3 * nc * na * 4, int(pinned_rows) * nc * self.n_omega * 8
        ni += (-ni) % 8
        buf = pinned_empty((nd + ni + nr,), np.uint8, min_bytes=0)
        d = buf[:nd].view(np.float64).reshape(3, nc, na)
        i = buf[nd:nd + 3 * nc * na * 4].view(np.int32).reshape(3, nc, na)
        rows = buf[nd + ni:].view(np.float64).reshape(int(pinned_rows), nc, self.n_omega)
        return dict(chi2=d[0], S=d[1], Q=d[2], n_iter=i[0], converged=i[1], n_evals=i[2], _d=d, _i=i, _rows=rows,
                    _pinned=is_pinned(buf))

    def prefetch(self, out):
        """``mxe_chains_prefetch``: the scalars of the launch copied into ``out`` (of ``result_arrays(pinned_rows=...)``) behind
        the kernel; :meth:`fetch` with the same ``out`` then waits and copies nothing"""
        self._check(self._lib.mxe_chains_prefetch(self._h, _p(out['_d']), _p(out['_i'])), 'mxe_chains_prefetch')
        self._held = getattr(self, '_held', [])
        self._held.append(out)          # (the destinations of copies in flight stay allocated until a call has waited for the stream)

    def select3_prefetch_rows(self, first, count, rows):
        """``mxe_select3_prefetch_rows``: behind :meth:`select3_launch`; :meth:`select3_fetch_rows` with the same arguments
        then waits and converts the indices"""
        self._check(self._lib.mxe_select3_prefetch_rows(self._h, int(first), int(count), _p(rows) if count > 0 else None),
                    'mxe_select3_prefetch_rows')
        self._held = getattr(self, '_held', [])
        self._held.append(rows)

    def fetch(self, want_v=True, want_H=True, out=None):
        """``out``: the arrays of :meth:`result_arrays`, made by the caller before the launch"""
        nc, na = self._n_chain, self._n_alpha
        if out is None:
            out = self.result_arrays()
        elif out['chi2'].shape != (nc, na):
            raise ValueError('result arrays of another launch')
        v = np.empty((nc, na, self._n_s_dev)) if want_v else None
        H = pinned_empty((nc, na, self.n_omega)) if want_H else None
        self._check(self._lib.mxe_chains_fetch(
            self._h, _p(v), _p(H), _p(out['chi2']), _p(out['S']), _p(out['Q']),
            _p(out['n_iter']), _p(out['converged']), _p(out['n_evals'])),
            'mxe_chains_fetch')
        self._held = []                 # (the stream has been waited for)
        if v is not None and self._n_s_dev < self.n_s:
            full = np.zeros((nc, na, self.n_s))
            full[..., :self._n_s_dev] = v
            v = full
        out['v'] = v
        out['H'] = H
        return out

    def logdet(self):
        """log det(I + M W / alpha) per problem of the last launch, [n_chain][n_alpha]."""
        out = np.empty((self._n_chain, self._n_alpha))
        self._check(self._lib.mxe_logdet(self._h, _p(out)), 'mxe_logdet')
        return out

    def fetch_n_act(self):
        """diagnostic: size of the coupled block per problem, [n_chain][n_alpha]."""
        out = np.empty((self._n_chain, self._n_alpha), dtype=np.int32)
        self._check(self._lib.mxe_chains_fetch_nact(self._h, _p(out)), 'mxe_chains_fetch_nact')
        return out

    def solve_chains(self, elem_of_chain, alpha_scaled, v0, opts=None,
                     want_v=True, want_H=True):
        """Blocking solve: upload, one launch, fetch (``mxe_solve_chains``)."""
        self.upload_chains(elem_of_chain, alpha_scaled, v0, opts)
        self.launch()
        self.finish()
        return self.fetch(want_v, want_H)

    def eval_batch(self, elem_of_problem, alpha_scaled, x, input_is_H=False, chi2_factor=1.0,
                   want=('Q', 'chi2', 'S', 'H', 'g', 'W')):
        """``mxe_eval_batch``: cost function and derivative ingredients at caller-supplied points.
        ``x``: (P, n_s) vectors v, or (P, n_omega) hidden images with ``input_is_H``.  ``want``: any of
        Q, chi2, S, H, u, w, q, h, g, W, W2.  Returns a dict of arrays."""
        if self._n_s_dev < self.n_s:
            raise MaxEntDeviceError('eval_batch on a context that keeps %d of %d singular directions' % (self._n_s_dev, self.n_s))
        el = _c(np.atleast_1d(elem_of_problem), np.int32)
        P = len(el)
        al = _c(np.broadcast_to(np.asarray(alpha_scaled, dtype=float), (P,)))
        x = _c(x).reshape(P, self.n_omega if input_is_H else self.n_s)
        shapes = dict(Q=(P,), chi2=(P,), S=(P,), H=(P, self.n_omega), u=(P, self.n_omega),
                      w=(P, self.n_omega), q=(P, self.n_omega), h=(P, self.n_s), g=(P, self.n_s),
                      W=(P, self.n_s, self.n_s), W2=(P, self.n_s, self.n_s))
        out = {}
        for k in want:
            if k not in shapes:
                raise TypeError('unknown output {!r}'.format(k))
            out[k] = np.empty(shapes[k])
        args = [_p(out.get(k)) for k in ('Q', 'chi2', 'S', 'H', 'u', 'w', 'q', 'h', 'g', 'W', 'W2')]
        self._check(self._lib.mxe_eval_batch(self._h, P, _p(el), _p(al), _p(x), int(bool(input_is_H)),
                                             float(chi2_factor), *args), 'mxe_eval_batch')
        return out

    def audit(self):
        """``mxe_audit``: exact Newton correction size and relative gradient of every problem of the
        last launch, each [n_chain][n_alpha]."""
        corr = np.empty((self._n_chain, self._n_alpha))
        gmax = np.empty((self._n_chain, self._n_alpha))
        self._check(self._lib.mxe_audit(self._h, _p(corr), _p(gmax)), 'mxe_audit')
        return dict(corr=corr, gmax=gmax)

    # -- the analyzer's alpha on the device, selected rows --------------------
    def select_launch(self, linefit_deg=0):
        self._check(self._lib.mxe_select_launch(self._h, int(linefit_deg)), 'mxe_select_launch')

    def select_fetch(self, want_H=True):
        idx = np.empty(self._n_chain, dtype=np.int32)
        Hs = np.empty((self._n_chain, self.n_omega)) if want_H else None
        self._check(self._lib.mxe_select_fetch(self._h, _p(idx), _p(Hs)), 'mxe_select_fetch')
        return idx, Hs

    def select3_launch(self, linefit_deg=0, gamma=0.2):
        """line fit, chi2 curvature and entropy analyzers of every scan of the last launch, on the ctx stream"""
        self._check(self._lib.mxe_select3_launch(self._h, int(linefit_deg), float(gamma)), 'mxe_select3_launch')

    def select3_fetch(self, want_H=True):
        """(indices [3][n_chain], rows [3][n_chain][n_omega]) of the three analyzers, one copy"""
        idx = np.empty((3, self._n_chain), dtype=np.int32)
        Hs = np.empty((3, self._n_chain, self.n_omega)) if want_H else None
        self._check(self._lib.mxe_select3_fetch(self._h, _p(idx), _p(Hs)), 'mxe_select3_fetch')
        return idx, Hs

    def select3_arrays(self, count=1):
        """uninitialised destinations of :meth:`select3_fetch_rows`: indices [3][n_chain], rows [count][n_chain][n_omega]
        (page-locked when large: one DMA)"""
        return np.empty((3, self._n_chain), dtype=np.int32), pinned_empty((count, self._n_chain, self.n_omega))

    def select3_fetch_rows(self, first=0, count=1, idx=None, rows=None, want_index=True):
        """``mxe_select3_fetch_rows``: the indices of all three analyzers (``want_index``) and the rows of analyzers
        ``first`` .. ``first + count - 1`` -- the others stay on the device until the next selection"""
        if want_index and idx is None:
            idx = np.empty((3, self._n_chain), dtype=np.int32)
        if rows is None and count > 0:
            rows = pinned_empty((count, self._n_chain, self.n_omega))
        if rows is not None and rows.shape != (count, self._n_chain, self.n_omega):
            raise ValueError('rows of another launch')
        self._check(self._lib.mxe_select3_fetch_rows(self._h, _p(idx) if want_index else None, int(first), int(count),
                                                     _p(rows) if count > 0 else None), 'mxe_select3_fetch_rows')
        return idx, rows

    def fetch_rows(self, problem_index):
        pi = _c(np.atleast_1d(problem_index), np.int32)
        out = np.empty((len(pi), self.n_omega))
        self._check(self._lib.mxe_fetch_rows(self._h, len(pi), _p(pi), _p(out)), 'mxe_fetch_rows')
        return out

    def compact_count(self):
        """doubles of the compact result pack of the staged chains (mxe_gather)"""
        P = self._n_chain * self._n_alpha
        return 3 * P + self._n_chain * (self.n_omega + 1)

    def full_count(self):
        return self._n_chain * self._n_alpha * self.n_omega + self.compact_count()

    # -- ranks in separate processes -----------------------------------------
    def comm_init(self, n_ranks, rank, unique_id):
        self._check(self._lib.mxe_comm_init(self._h, int(n_ranks), int(rank), bytes(unique_id)), 'mxe_comm_init')

    def comm_set_loopback(self, on=True):
        """one-GPU test plumbing: the root's own pack through ncclSend / ncclRecv to itself, ncclAllReduce with one rank"""
        self._check(self._lib.mxe_comm_set_loopback(self._h, 1 if on else 0), 'mxe_comm_set_loopback')

    def comm_destroy(self):
        self._check(self._lib.mxe_comm_destroy(self._h), 'mxe_comm_destroy')

    def gather(self, root, counts, full=False, recv=None):
        counts = _c(counts, np.int64)
        self._check(self._lib.mxe_gather(self._h, int(root), 1 if full else 0, _p(counts), _p(recv)), 'mxe_gather')
        return recv

    def allreduce(self, values, op='sum'):
        x = _c(np.atleast_1d(values))
        self._check(self._lib.mxe_comm_allreduce(self._h, _p(x), len(x), 0 if op == 'sum' else 1), 'mxe_comm_allreduce')
        return x

    def apply_output_map(self, B):
        B = _c(B).reshape(self.n_omega, self.n_omega)
        A = np.empty((self._n_chain, self._n_alpha, self.n_omega))
        self._check(self._lib.mxe_apply_output_map(self._h, _p(B), _p(A)),
                    'mxe_apply_output_map')
        return A

    def last_kernel_ms(self):
        ms = ctypes.c_float(0)
        self._check(self._lib.mxe_last_kernel_ms(self._h, ctypes.byref(ms)),
                    'mxe_last_kernel_ms')
        return float(ms.value)

    def timing_mark(self):
        self._check(self._lib.mxe_timing_mark(self._h), 'mxe_timing_mark')

    def ms_since_mark(self):
        ms = ctypes.c_float(0)
        self._check(self._lib.mxe_ms_since_mark(self._h, ctypes.byref(ms)), 'mxe_ms_since_mark')
        return float(ms.value)

    def stream_handle(self):
        return int(self._lib.mxe_stream(self._h) or 0)

    def last_launch_info(self):
        a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        self._check(self._lib.mxe_last_launch_info(self._h, ctypes.byref(a),
                                                   ctypes.byref(b),
                                                   ctypes.byref(c)),
                    'mxe_last_launch_info')
        return dict(waves_per_chain=a.value, n_workgroups=b.value,
                    lds_bytes=c.value,
                    kernel=self._lib.mxe_last_kernel_name(self._h).decode())

    def schedule_info(self):
        """``mxe_schedule_info``: dict(n_solo, placement_rule) of the staged chains (placement_rule 0: not needed, 1: probed and
        holds, 2: does not hold -- no solo workgroups)"""
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        self._check(self._lib.mxe_schedule_info(self._h, ctypes.byref(a), ctypes.byref(b)), 'mxe_schedule_info')
        return dict(n_solo=a.value, placement_rule=b.value)

    def launch_depth(self):
        """``mxe_launch_depth``: rounds of the deepest workgroup and the mean over the workgroups of the last lock-step launch,
        per pass: dict(max_rounds=[a, b], mean_rounds=[a, b])"""
        mx, mean = np.zeros(2, dtype=np.int32), np.zeros(2)
        self._check(self._lib.mxe_launch_depth(self._h, _p(mx), _p(mean)), 'mxe_launch_depth')
        return dict(max_rounds=[int(mx[0]), int(mx[1])], mean_rounds=[float(mean[0]), float(mean[1])])

    def set_result_buffer(self, which):
        self._check(self._lib.mxe_set_result_buffer(self._h, int(which)),
                    'mxe_set_result_buffer')

    def result_device_ptrs(self):
        ptrs = [_vp(None) for _ in range(7)]
        self._check(self._lib.mxe_result_device_ptrs(
            self._h, *[ctypes.byref(x) for x in ptrs]),
            'mxe_result_device_ptrs')
        names = ['H', 'chi2', 'S', 'Q', 'v', 'n_iter', 'converged']
        return dict(zip(names, [x.value for x in ptrs]))
