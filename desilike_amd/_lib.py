"""ctypes binding of the C ABI declared in ``include/desilike_amd.h`` (thin: plain pointers and sizes).

The shared library holds the hand-written HIP kernels for gfx950.  There is NO CPU fallback:
if the library is missing, or no GPU is visible, evaluation raises.
"""
import ctypes
import os

import numpy as np

_LIB_PATH = os.environ.get('DL_LIB_PATH') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libdesilike_amd.so')   # DL_LIB_PATH: another build of the same library (A/B runs of a kernel change on one box)
_lib = None

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int32_p = ctypes.POINTER(ctypes.c_int32)

# every symbol include/desilike_amd.h declares: (restype, argtypes)
SYMBOLS = {
    'dl_config_new': (ctypes.c_void_p, []),
    'dl_config_set_f64': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, _c_double_p, ctypes.c_int64]),
    'dl_config_set_i32': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, _c_int32_p, ctypes.c_int64]),
    'dl_config_free': (None, [ctypes.c_void_p]),
    'dl_options_refresh': (None, []),
    'dl_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p]),
    'dl_destroy': (None, [ctypes.c_void_p]),
    'dl_last_error': (ctypes.c_char_p, [ctypes.c_void_p]),
    'dl_info': (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_char_p]),
    'dl_eval_batch': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'dl_eval_batch_derived': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'dl_eval_logposterior': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'dl_eval_logposterior_grad': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'dl_eval_fisher': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'dl_eval_theory': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'dl_eval_batch_host': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_int64, _c_double_p, _c_double_p, _c_double_p, _c_int32_p, _c_double_p]),
    'dl_eval_tns_tables': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    'dl_eval_theory_host': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_int64, ctypes.c_int32, _c_double_p, _c_double_p]),
    'dl_eval_logposterior_host': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_int64, _c_double_p, _c_int32_p]),
    'dl_profile_enable': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'dl_profile_read': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_int32]),
    'dl_fftlog_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _c_double_p, _c_double_p, _c_double_p]),
    'dl_fftlog_apply': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    'dl_fftlog_destroy': (None, [ctypes.c_void_p]),
    'dl_comm_unique_id': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p]),
    'dl_comm_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p]),
    'dl_comm_destroy': (None, [ctypes.c_void_p]),
    'dl_comm_info': (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_char_p]),
    'dl_comm_allgather_f64': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    'dl_comm_broadcast_f64': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    'dl_ensemble_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int32, ctypes.c_double, ctypes.c_uint64, ctypes.c_double, ctypes.c_void_p]),
    'dl_ensemble_destroy': (None, [ctypes.c_void_p]),
    'dl_ensemble_set_state': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, _c_double_p, ctypes.c_void_p]),
    'dl_ensemble_run': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'dl_ensemble_get_state': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, _c_double_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p]),
    'dl_ensemble_set_counter': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p]),
    'dl_ensemble_info': (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_char_p]),
    'dl_mh_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                    ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_double, ctypes.c_uint64, ctypes.c_double, ctypes.c_int64]),
    'dl_mh_destroy': (None, [ctypes.c_void_p]),
    'dl_mh_set_covariance': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_void_p]),
    'dl_mh_set_state': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, _c_double_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), ctypes.c_int64, ctypes.c_void_p]),
    'dl_mh_run': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'dl_mh_run_host': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, _c_double_p, _c_double_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p]),
    'dl_mh_get_state': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, _c_double_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p]),
    'dl_mh_info': (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_char_p]),
    'dl_mlp_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int32, _c_int32_p, ctypes.c_int32, _c_double_p]),
    'dl_mlp_destroy': (None, [ctypes.c_void_p]),
    'dl_mlp_info': (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_char_p]),
    'dl_mlp_train': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                    ctypes.c_double, _c_double_p, ctypes.c_void_p]),
    'dl_mlp_loss_and_grad': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, _c_double_p, _c_double_p, ctypes.c_void_p]),
    'dl_mlp_forward': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    'dl_mlp_get_weights': (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_void_p]),
    'dl_cov_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int32, ctypes.c_int32, _c_int32_p, _c_int32_p, _c_int32_p, _c_double_p, ctypes.c_int64, _c_int32_p,
                                     _c_double_p, ctypes.c_int64, _c_int32_p, _c_double_p, ctypes.c_int32, _c_double_p, ctypes.c_int64, _c_int32_p]),
    'dl_cov_apply': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    'dl_cov_destroy': (None, [ctypes.c_void_p]),
}


class LibraryError(RuntimeError):
    """Raised when the HIP library is missing or a C-ABI call fails."""


def lib_path():
    return _LIB_PATH


def load():
    """Load ``libdesilike_amd.so`` (built by ``__graft_entry__.build()`` / ``make -C desilike_amd/csrc``)."""
    global _lib
    if _lib is None:
        if not os.path.isfile(_LIB_PATH):
            raise LibraryError('HIP library {} not found: build it with `python -c "import __graft_entry__ as g; g.build()"`; there is no CPU fallback'.format(_LIB_PATH))
        # kernel arguments in device memory (latency-bound launches wait for them first; the runtime's default on ROCm 7, stated for older defaults; no effect once
        # HIP is initialised)
        os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
        try:
            # PyTorch-ROCm wheels bundle their own libamdhip64: load it FIRST so that this library binds to the same HIP runtime
            # (two HIP runtimes in one process do not see each other's devices: "No HIP GPUs are available")
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(_LIB_PATH)
        for name, (restype, argtypes) in SYMBOLS.items():
            func = getattr(lib, name)
            func.restype, func.argtypes = restype, argtypes
        _lib = lib
    return _lib


def refresh_options():
    """Re-read the library's diagnostic switches (environment variables ``DL_*``, include/desilike_amd.h) -- they are read once per process; the tests flip them in place."""
    load().dl_options_refresh()


def rccl_library_path():
    """RCCL to bind (``dl_comm_*`` load it with dlopen): ``DL_RCCL_PATH`` if set, else the copy PyTorch-ROCm bundles (the one already mapped into this process when
    torch is imported: one RCCL and one HIP runtime per process), else ``None`` (the library's own search: already-loaded copy, default path, /opt/rocm/lib)."""
    path = os.environ.get('DL_RCCL_PATH', None)
    if path:
        return path
    try:
        import torch
        path = os.path.join(os.path.dirname(os.path.abspath(torch.__file__)), 'lib', 'librccl.so')
        if os.path.isfile(path): return path
    except ImportError:
        pass
    return None


def _f64_ptr(array):
    return None if array is None else array.ctypes.data_as(_c_double_p)


def _i32_ptr(array):
    return None if array is None else array.ctypes.data_as(_c_int32_p)


def fill_config(spec, set_f64, set_i32):
    """Flatten a likelihood ``spec`` (nested dict, see ``desilike_amd.compile``) into config keys (include/desilike_amd.h).

    ``set_f64(key, float64 array)`` and ``set_i32(key, int32 array)`` receive contiguous 1-D arrays.
    """
    def put(key, value):
        value = np.asarray(value)
        if value.dtype.kind in 'iub':
            set_i32(key, np.ascontiguousarray(value.ravel(), dtype=np.int32))
        else:
            set_f64(key, np.ascontiguousarray(value.ravel(), dtype=np.float64))

    for key, value in spec.items():
        if key.startswith('_'):    # host-side entries (Context reads '_expand')
            continue
        if key == 'observables':
            for iobs, obs in enumerate(value):
                for okey, ovalue in obs.items():
                    if okey == 'inputs':
                        for name, (col, const) in ovalue.items():
                            if name in ('ct', 'sn', 'pass', 'x', 'vp', 'ml', 'band'):
                                put('obs{:d}.in.{}'.format(iobs, name), np.array([[c, v] for c, v in zip(np.ravel(col), np.ravel(const))], dtype='f8'))
                            else:
                                put('obs{:d}.in.{}'.format(iobs, name), np.array([col, const], dtype='f8'))
                    elif okey.startswith('emu') and isinstance(ovalue, dict):
                        for name, array in ovalue.items():
                            put('obs{:d}.{}.{}'.format(iobs, okey, name), array)
                    elif okey == 'marg':
                        for name, index in ovalue.items():
                            put('obs{:d}.marg.{}'.format(iobs, name), np.asarray(index, dtype='i4'))
                    elif ovalue is not None:
                        put('obs{:d}.{}'.format(iobs, okey), ovalue)
            put('n_obs', np.array([len(value)], dtype='i4'))
        elif key == 'marg':
            for name, array in value.items():
                put('marg.{}'.format(name), array)
        elif value is not None:
            put(key, value)


class Context(object):
    """Owner of one ``dl_ctx`` (device constants + workspaces) on one GPU."""

    def __init__(self, spec, device=0):
        lib = load()
        cfg = lib.dl_config_new()
        keep = []

        def set_f64(key, array):
            keep.append(array)
            if lib.dl_config_set_f64(cfg, key.encode(), _f64_ptr(array), array.size): raise LibraryError(lib.dl_last_error(None).decode())

        def set_i32(key, array):
            keep.append(array)
            if lib.dl_config_set_i32(cfg, key.encode(), _i32_ptr(array), array.size): raise LibraryError(lib.dl_last_error(None).decode())

        try:
            fill_config(spec, set_f64, set_i32)
            handle = ctypes.c_void_p()
            rc = lib.dl_create(ctypes.byref(handle), int(device), cfg)
        finally:
            lib.dl_config_free(cfg)
        if rc != 0:
            raise LibraryError(lib.dl_last_error(None).decode())
        self._lib, self._handle, self.device = lib, handle, int(device)
        self.n_params, self.n_data, self.n_obs, self.n_solved = (self.info(name) for name in ['n_params', 'n_data', 'n_obs', 'n_solved'])
        # parameters derived from others by an expression (parameter.py:758-807) are extra theta columns of the device context, computed here from the caller's
        # columns: ``expand(theta [B, n_params]) -> [B, n_device_params]`` (numpy array or torch tensor, whatever it is given), set by the likelihood
        self.n_device_params, self.expand = self.n_params, None
        if spec.get('_expand', None) is not None:
            self.set_expand(*spec['_expand'])

    def set_expand(self, expand, n_params):
        """Callers pass ``theta [B, n_params]``; the device context receives ``expand(theta) [B, n_device_params]``."""
        self.expand, self.n_params = expand, int(n_params)

    def _host_theta(self, theta):
        theta = np.ascontiguousarray(np.atleast_2d(theta), dtype='f8')
        if theta.shape[1] != self.n_params:
            raise ValueError('theta must have shape (B, {:d}), found {}'.format(self.n_params, theta.shape))
        if self.expand is not None:
            theta = np.ascontiguousarray(self.expand(theta), dtype='f8')
            assert theta.shape[1] == self.n_device_params
        return theta

    def _device_theta(self, theta, stream):
        import torch
        assert theta.is_contiguous() and theta.dtype == torch.float64 and theta.shape[1] == self.n_params
        if self.expand is not None:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=theta.device)):   # the extra columns are computed on the stream the evaluation runs on
                theta = self.expand(theta).contiguous()
            assert theta.shape[1] == self.n_device_params
            self._expanded = theta    # alive until the next call on this context (calls on one context are serialised: include/desilike_amd.h)
        return theta

    def info(self, key):
        return int(self._lib.dl_info(self._handle, key.encode()))

    def _check(self, rc):
        if rc != 0:
            raise LibraryError(self._lib.dl_last_error(self._handle).decode())

    def close(self):
        if getattr(self, '_handle', None):
            self._lib.dl_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-pointer entry points (numpy in / numpy out) ----
    def eval_batch_host(self, theta, return_flattheory=False, return_solved=False):
        """numpy in / numpy out: (loglike, logprior, status[, flattheory][, solved])."""
        theta = self._host_theta(theta)
        B = theta.shape[0]
        loglike, logprior, status = np.empty(B, dtype='f8'), np.empty(B, dtype='f8'), np.empty(B, dtype='i4')
        flat = np.empty((B, self.n_data), dtype='f8') if return_flattheory else None
        solved = np.empty((B, self.n_solved), dtype='f8') if return_solved else None
        self._check(self._lib.dl_eval_batch_host(self._handle, _f64_ptr(theta), B, _f64_ptr(loglike), _f64_ptr(logprior), _f64_ptr(flat), _i32_ptr(status),
                                                 _f64_ptr(solved) if self.n_solved else None))
        toret = (loglike, logprior, status)
        if return_flattheory: toret += (flat,)
        if return_solved: toret += (solved,)
        return toret

    def eval_logposterior_host(self, theta):
        """numpy in / numpy out: (logposterior [B], status [B]) with the samplers' -inf conventions applied on the device (samplers/base.py:144-200)."""
        theta = self._host_theta(theta)
        B = theta.shape[0]
        logposterior, status = np.empty(B, dtype='f8'), np.empty(B, dtype='i4')
        self._check(self._lib.dl_eval_logposterior_host(self._handle, _f64_ptr(theta), B, _f64_ptr(logposterior), _i32_ptr(status)))
        return logposterior, status

    def eval_batch_derived_host(self, theta):
        """numpy in / numpy out: (loglike, logprior, status, solved [B, n_solved], hessian [B, n_solved, n_solved]); stages through device tensors."""
        import torch
        theta = np.ascontiguousarray(np.atleast_2d(theta), dtype='f8')    # (columns of derived-by-expression parameters are added by eval_batch_derived)
        B, ns = theta.shape[0], self.n_solved
        device = torch.device('cuda', self.device)
        th = torch.as_tensor(theta, dtype=torch.float64, device=device).contiguous()
        loglike, logprior = torch.empty(B, dtype=torch.float64, device=device), torch.empty(B, dtype=torch.float64, device=device)
        status = torch.empty(B, dtype=torch.int32, device=device)
        solved, hessian = torch.empty((B, ns), dtype=torch.float64, device=device), torch.empty((B, ns, ns), dtype=torch.float64, device=device)
        if B:
            self.eval_batch_derived(th, loglike=loglike, logprior=logprior, status=status, solved=solved, hessian=hessian,
                                    stream=torch.cuda.current_stream(device).cuda_stream)
            torch.cuda.synchronize(device)
        return tuple(t.cpu().numpy() for t in (loglike, logprior, status, solved, hessian))

    def eval_theory_host(self, theta, iobs=0, return_tables=False):
        theta = self._host_theta(theta)
        B = theta.shape[0]
        n_ell, n_kin = self.info('n_ell_obs{:d}'.format(iobs)), self.info('n_kin_obs{:d}'.format(iobs))
        power = np.empty((B, n_ell, n_kin), dtype='f8')
        tables = np.empty((B, 3, n_ell, n_kin), dtype='f8') if return_tables else None
        self._check(self._lib.dl_eval_theory_host(self._handle, _f64_ptr(theta), B, int(iobs), _f64_ptr(power), _f64_ptr(tables)))
        return (power, tables) if return_tables else power

    # ---- device-pointer entry points (torch tensors as the array container) ----
    def eval_batch(self, theta, loglike=None, logprior=None, flattheory=None, status=None, solved=None, stream=None):
        """All arguments are CUDA(ROCm) torch tensors on this context's device; asynchronous on ``stream``."""
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(theta.device).cuda_stream
        B = theta.shape[0]
        theta = self._device_theta(theta, stream)

        def ptr(tensor, dtype, shape):
            if tensor is None: return None
            assert tensor.is_contiguous() and tensor.dtype == dtype and tuple(tensor.shape) == shape, (tensor.shape, shape)
            return ctypes.c_void_p(tensor.data_ptr())

        self._check(self._lib.dl_eval_batch(self._handle, ctypes.c_void_p(theta.data_ptr()), B, ptr(loglike, torch.float64, (B,)), ptr(logprior, torch.float64, (B,)),
                                            ptr(flattheory, torch.float64, (B, self.n_data)), ptr(status, torch.int32, (B,)),
                                            ptr(solved, torch.float64, (B, self.n_solved)), ctypes.c_void_p(stream)))

    def eval_batch_derived(self, theta, loglike=None, logprior=None, status=None, solved=None, hessian=None, stream=None):
        """``eval_batch`` plus ``hessian [B, n_solved, n_solved]``: likelihood Hessian w.r.t. the analytically solved parameters (torch tensors, asynchronous)."""
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(theta.device).cuda_stream
        B = theta.shape[0]
        theta = self._device_theta(theta, stream)

        def ptr(tensor, dtype, shape):
            if tensor is None: return None
            assert tensor.is_contiguous() and tensor.dtype == dtype and tuple(tensor.shape) == shape, (tensor.shape, shape)
            return ctypes.c_void_p(tensor.data_ptr())

        self._check(self._lib.dl_eval_batch_derived(self._handle, ctypes.c_void_p(theta.data_ptr()), B, ptr(loglike, torch.float64, (B,)), ptr(logprior, torch.float64, (B,)),
                                                    ptr(status, torch.int32, (B,)), ptr(solved, torch.float64, (B, self.n_solved)),
                                                    ptr(hessian, torch.float64, (B, self.n_solved, self.n_solved)), ctypes.c_void_p(stream)))

    def eval_logposterior(self, theta, logposterior, status=None, stream=None):
        """``logposterior[B]`` = loglikelihood + logprior, -inf where the sampler would reject the point (samplers/base.py:185-191); torch tensors, asynchronous."""
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(theta.device).cuda_stream
        B = theta.shape[0]
        theta = self._device_theta(theta, stream)
        assert logposterior.is_contiguous() and logposterior.dtype == torch.float64 and tuple(logposterior.shape) == (B,)
        assert status is None or (status.is_contiguous() and status.dtype == torch.int32 and tuple(status.shape) == (B,))
        self._check(self._lib.dl_eval_logposterior(self._handle, ctypes.c_void_p(theta.data_ptr()), B, ctypes.c_void_p(logposterior.data_ptr()),
                                                   None if status is None else ctypes.c_void_p(status.data_ptr()), ctypes.c_void_p(stream)))

    def eval_theory(self, theta, power, iobs=0, tables=None, stream=None):
        """Theory multipoles of observable ``iobs`` into the device tensor ``power [B, n_ell, n_kin]`` (``dl_eval_theory``; torch tensors, asynchronous)."""
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(theta.device).cuda_stream
        B = theta.shape[0]
        n_ell, n_kin = self.info('n_ell_obs{:d}'.format(iobs)), self.info('n_kin_obs{:d}'.format(iobs))
        theta = self._device_theta(theta, stream)
        assert power.is_contiguous() and power.dtype == torch.float64 and tuple(power.shape) == (B, n_ell, n_kin)
        assert tables is None or (tables.is_contiguous() and tables.dtype == torch.float64 and tuple(tables.shape) == (B, 3, n_ell, n_kin))
        self._check(self._lib.dl_eval_theory(self._handle, ctypes.c_void_p(theta.data_ptr()), B, int(iobs), ctypes.c_void_p(power.data_ptr()),
                                             None if tables is None else ctypes.c_void_p(tables.data_ptr()), ctypes.c_void_p(stream)))
        return power

    def eval_tns_tables(self, theta, n_k11, iobs=0, stream=None):
        """The 29 one-loop tables ``[B, 29, n_k11]`` of a TNS observable before AP / damping / projection (``dl_eval_tns_tables``); ``theta``: array or device tensor."""
        import torch
        if not torch.is_tensor(theta):
            theta = torch.as_tensor(np.ascontiguousarray(theta, dtype='f8'), device='cuda:{:d}'.format(self.device))
        if stream is None:
            stream = torch.cuda.current_stream(theta.device).cuda_stream
        B = theta.shape[0]
        theta = self._device_theta(theta, stream)
        tables = torch.empty((B, 29, int(n_k11)), dtype=torch.float64, device=theta.device)
        self._check(self._lib.dl_eval_tns_tables(self._handle, ctypes.c_void_p(theta.data_ptr()), B, int(iobs), ctypes.c_void_p(tables.data_ptr()), ctypes.c_void_p(stream)))
        return tables

    def eval_logposterior_grad(self, theta, logposterior=None, grad=None, status=None, stream=None):
        """Log-posterior ``[B]`` and its analytic gradient ``[B, P]`` (``dl_eval_logposterior_grad``; float64 device tensors, allocated if ``None``; asynchronous).
        Returns ``(logposterior, grad)``, or ``None`` when the context is outside the analytic gradient's scope (the caller differentiates numerically)."""
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(theta.device).cuda_stream
        if self.expand is not None:
            return None      # parameters derived by an expression: chain rule through the expression is not implemented
        B = theta.shape[0]
        assert theta.is_contiguous() and theta.dtype == torch.float64 and theta.shape[1] == self.n_params
        if logposterior is None: logposterior = torch.empty(B, dtype=torch.float64, device=theta.device)
        if grad is None: grad = torch.empty((B, self.n_params), dtype=torch.float64, device=theta.device)
        rc = self._lib.dl_eval_logposterior_grad(self._handle, ctypes.c_void_p(theta.data_ptr()), B, ctypes.c_void_p(logposterior.data_ptr()), ctypes.c_void_p(grad.data_ptr()),
                                                 None if status is None else ctypes.c_void_p(status.data_ptr()), ctypes.c_void_p(stream))
        if rc == 2: return None
        self._check(rc)
        return logposterior, grad

    def eval_fisher(self, centers, steps, hessian=None, gradient=None, offset=None, stream=None):
        """Fisher algebra on the device (``dl_eval_fisher``): ``centers [B, P]``, ``steps [B, P, 2]`` (lower, upper) -> ``hessian [B, P, P]``, ``gradient [B, P]``,
        ``offset [B]`` (float64 device tensors, allocated if ``None``); asynchronous on ``stream``."""
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(centers.device).cuda_stream
        B, P = centers.shape
        if self.expand is not None:
            raise NotImplementedError('Fisher algebra with parameters derived by an expression: differentiate w.r.t. the device columns and apply the chain rule on the host')
        assert P == self.n_params and centers.is_contiguous() and centers.dtype == torch.float64
        assert steps.is_contiguous() and steps.dtype == torch.float64 and tuple(steps.shape) == (B, P, 2)
        if hessian is None: hessian = torch.empty((B, P, P), dtype=torch.float64, device=centers.device)
        if gradient is None: gradient = torch.empty((B, P), dtype=torch.float64, device=centers.device)
        if offset is None: offset = torch.empty(B, dtype=torch.float64, device=centers.device)
        for tensor, shape in [(hessian, (B, P, P)), (gradient, (B, P)), (offset, (B,))]:
            assert tensor.is_contiguous() and tensor.dtype == torch.float64 and tuple(tensor.shape) == shape
        self._check(self._lib.dl_eval_fisher(self._handle, ctypes.c_void_p(centers.data_ptr()), ctypes.c_void_p(steps.data_ptr()), B, ctypes.c_void_p(hessian.data_ptr()),
                                             ctypes.c_void_p(gradient.data_ptr()), ctypes.c_void_p(offset.data_ptr()), ctypes.c_void_p(stream)))
        return hessian, gradient, offset

    def profile_enable(self, every=1, only=None):
        """Attach HIP events to the kernels' dispatch packets on one ``eval_batch`` call out of ``every`` (0 / False: off); ``only``: 'theory' /
        'window_gemm' / 'finalize' -- a single kernel per sampled call (short runs: every kernel launched with events costs the step ~3 us)."""
        flags = 0
        if only is not None and every:
            flags = (1 << 16) | (['theory', 'window_gemm', 'finalize'].index(only) << 17)
        self._check(self._lib.dl_profile_enable(self._handle, int(every) | flags))

    def profile_read(self):
        ms = np.zeros(9, dtype='f8')
        self._check(self._lib.dl_profile_read(self._handle, _f64_ptr(ms), 9))
        return dict(theory=ms[0], window_gemm=ms[1], finalize=ms[2], total=ms[3], event_overhead=ms[4], samples=int(ms[5]),
                    samples_per_kernel=dict(theory=int(ms[6]), window_gemm=int(ms[7]), finalize=int(ms[8])))


class FFTLogPlan(object):
    """Owner of one ``dl_fftlog`` plan (include/desilike_amd.h): batched FFTLog Hankel transform of ``fun [B, n_ell, n]`` on one GPU."""

    def __init__(self, n, npad, pre, u, post, device=0):
        lib = load()
        pre = np.ascontiguousarray(pre, dtype='f8')
        u = np.ascontiguousarray(u, dtype='f8')          # [n_ell, npad // 2 + 1, 2]
        post = np.ascontiguousarray(post, dtype='f8')    # [n_ell, n]
        self.n, self.npad, self.n_ell, self.device = int(n), int(npad), int(post.shape[0]), int(device)
        if pre.shape != (self.n,) or u.shape != (self.n_ell, self.npad // 2 + 1, 2) or post.shape != (self.n_ell, self.n):
            raise ValueError('inconsistent FFTLog plan arrays: pre {}, u {}, post {}'.format(pre.shape, u.shape, post.shape))
        handle = ctypes.c_void_p()
        if lib.dl_fftlog_create(ctypes.byref(handle), self.device, self.n, self.npad, self.n_ell, _f64_ptr(pre), _f64_ptr(u), _f64_ptr(post)) != 0:
            raise LibraryError(lib.dl_last_error(None).decode())
        self._lib, self._handle = lib, handle

    def apply(self, fun, out=None, stream=None):
        """``fun``: contiguous float64 CUDA(ROCm) tensor [B, n_ell, n] on this plan's device; returns ``out`` (allocated if None); asynchronous on ``stream``."""
        import torch
        assert fun.is_cuda and fun.is_contiguous() and fun.dtype == torch.float64 and tuple(fun.shape[1:]) == (self.n_ell, self.n), fun.shape
        if out is None: out = torch.empty_like(fun)
        assert out.is_contiguous() and out.dtype == torch.float64 and out.shape == fun.shape and out.device == fun.device
        if stream is None: stream = torch.cuda.current_stream(fun.device).cuda_stream
        if self._lib.dl_fftlog_apply(self._handle, ctypes.c_void_p(fun.data_ptr()), fun.shape[0], ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream)) != 0:
            raise LibraryError(self._lib.dl_last_error(None).decode())
        return out

    def close(self):
        if getattr(self, '_handle', None):
            self._lib.dl_fftlog_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceEnsemble(object):
    """Owner of one ``dl_ensemble`` (include/desilike_amd.h): affine-invariant ensemble sampler resident on the GPU of ``ctx``; ``group``: an
    :class:`desilike_amd.parallel.RcclGroup` to shard every half-step's proposals over the ranks (or ``None``)."""

    def __init__(self, ctx, nwalkers, a=2., seed=0, offset=0., group=None):
        lib = load()
        if ctx.expand is not None:
            raise NotImplementedError('the device-resident ensemble proposes in the columns of the device context: parameters derived by an expression need the host-driven sampler')
        handle = ctypes.c_void_p()
        comm = getattr(group, 'handle', None)
        if lib.dl_ensemble_create(ctypes.byref(handle), ctx._handle, int(nwalkers), float(a), ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), float(offset), comm) != 0:
            raise LibraryError(lib.dl_last_error(None).decode())
        self._lib, self._handle, self._ctx, self._group = lib, handle, ctx, group   # (the context and the group must outlive the ensemble)
        self.nwalkers, self.n_params, self.device = int(nwalkers), ctx.n_params, ctx.device

    def _check(self, rc):
        if rc != 0:
            raise LibraryError(self._lib.dl_last_error(None).decode())

    def info(self, key):
        return int(self._lib.dl_ensemble_info(self._handle, key.encode()))

    def _stream(self, stream):
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream(torch.device('cuda', self.device)).cuda_stream if stream is None else stream)

    def set_state(self, coords, logposterior=None, stream=None):
        coords = np.ascontiguousarray(coords, dtype='f8')
        if coords.shape != (self.nwalkers, self.n_params):
            raise ValueError('coords must have shape ({:d}, {:d}), found {}'.format(self.nwalkers, self.n_params, coords.shape))
        if logposterior is not None:
            logposterior = np.ascontiguousarray(logposterior, dtype='f8')
            if logposterior.shape != (self.nwalkers,): raise ValueError('logposterior must have shape ({:d},)'.format(self.nwalkers))
        self._check(self._lib.dl_ensemble_set_state(self._handle, _f64_ptr(coords), _f64_ptr(logposterior), self._stream(stream)))

    def set_counter(self, iteration, naccepted=None, stream=None):
        """Resume: iteration counter of the counter-based generator (and accepted counts per walker)."""
        nacc = None
        if naccepted is not None:
            naccepted = np.ascontiguousarray(naccepted, dtype='i8')
            if naccepted.shape != (self.nwalkers,): raise ValueError('naccepted must have shape ({:d},)'.format(self.nwalkers))
            nacc = naccepted.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
        self._check(self._lib.dl_ensemble_set_counter(self._handle, int(iteration), nacc, self._stream(stream)))

    def run(self, niterations, thin_by=1, chain=None, chain_logp=None, stream=None):
        """Enqueue ``niterations`` ensemble updates (asynchronous); ``chain [niterations // thin_by, nwalkers, P]`` / ``chain_logp [niterations // thin_by, nwalkers]``:
        float64 device tensors receiving the ensemble after every ``thin_by``-th update (optional)."""
        import torch
        nrec = int(niterations) // int(thin_by)

        def ptr(tensor, shape):
            if tensor is None: return None
            assert tensor.is_cuda and tensor.is_contiguous() and tensor.dtype == torch.float64 and tuple(tensor.shape) == shape, (tensor.shape, shape)
            return ctypes.c_void_p(tensor.data_ptr())

        self._check(self._lib.dl_ensemble_run(self._handle, int(niterations), int(thin_by), ptr(chain, (nrec, self.nwalkers, self.n_params)),
                                              ptr(chain_logp, (nrec, self.nwalkers)), self._stream(stream)))

    def get_state(self, stream=None):
        """(coords [nwalkers, P], logposterior [nwalkers], naccepted [nwalkers]) as numpy arrays; synchronises the stream."""
        coords, logp = np.empty((self.nwalkers, self.n_params), dtype='f8'), np.empty(self.nwalkers, dtype='f8')
        nacc = np.empty(self.nwalkers, dtype='i8')
        self._check(self._lib.dl_ensemble_get_state(self._handle, _f64_ptr(coords), _f64_ptr(logp), nacc.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), self._stream(stream)))
        return coords, logp, nacc

    def close(self):
        if getattr(self, '_handle', None):
            self._lib.dl_ensemble_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CovariancePlan(object):
    """Owner of one ``dl_cov`` plan (include/desilike_amd.h): Gaussian covariance of multipole observables from theory spectra resident on the GPU.  ``arrays``: dict with the
    plan's host arrays (built by ``desilike_amd.observables.galaxy_clustering.covariance``)."""

    def __init__(self, n, n_ell, n_k, ell0, shotnoise, cell_i, cell_d, pt_i, pt_d, gtab, sym, device=0):
        lib = load()
        n_ell, n_k, ell0 = (np.ascontiguousarray(a, dtype=np.int32) for a in (n_ell, n_k, ell0))
        shotnoise = np.ascontiguousarray(shotnoise, dtype='f8')
        cell_i, pt_i, sym = (np.ascontiguousarray(a, dtype=np.int32) for a in (np.reshape(cell_i, (-1, 8)), np.reshape(pt_i, (-1, 2)), np.reshape(sym, (-1, 2))))
        cell_d, pt_d, gtab = (np.ascontiguousarray(a, dtype='f8') for a in (np.reshape(cell_d, (-1, 4)), np.reshape(pt_d, (-1, 6)), np.reshape(gtab, (-1, 5, 5))))
        handle = ctypes.c_void_p()
        if lib.dl_cov_create(ctypes.byref(handle), int(device), int(n), len(n_ell), _i32_ptr(n_ell), _i32_ptr(n_k), _i32_ptr(ell0), _f64_ptr(shotnoise), len(cell_i), _i32_ptr(cell_i),
                             _f64_ptr(cell_d), len(pt_i), _i32_ptr(pt_i), _f64_ptr(pt_d), len(gtab), _f64_ptr(gtab), len(sym), _i32_ptr(sym)) != 0:
            raise LibraryError(lib.dl_last_error(None).decode())
        self._lib, self._handle, self.device, self.n, self.shapes = lib, handle, int(device), int(n), [(int(a), int(b)) for a, b in zip(n_ell, n_k)]

    def apply(self, powers, out=None, stream=None):
        """``powers``: one float64 device tensor ``[B, n_ell, n_k]`` per theory; returns the covariance matrices ``[B, n, n]`` (device tensor; asynchronous)."""
        import torch
        B = powers[0].shape[0]
        for power, shape in zip(powers, self.shapes):
            assert power.is_cuda and power.is_contiguous() and power.dtype == torch.float64 and tuple(power.shape) == (B,) + shape, (power.shape, shape)
        if out is None: out = torch.empty((B, self.n, self.n), dtype=torch.float64, device=powers[0].device)
        assert out.is_contiguous() and out.dtype == torch.float64 and tuple(out.shape) == (B, self.n, self.n)
        if stream is None: stream = torch.cuda.current_stream(powers[0].device).cuda_stream
        ptrs = (ctypes.c_void_p * len(powers))(*[power.data_ptr() for power in powers])
        if self._lib.dl_cov_apply(self._handle, ptrs, B, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream)) != 0:
            raise LibraryError(self._lib.dl_last_error(None).decode())
        return out

    def close(self):
        if getattr(self, '_handle', None):
            self._lib.dl_cov_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MLPTrainer(object):
    """Owner of one ``dl_mlp`` (include/desilike_amd.h): fp64 Adam training of a dense network on one GPU.  ``layers``: list of (kernel [n_in, n_out], bias [n_out])
    initial values; ``activation``: 'silu' / 'relu' / 'tanh'."""

    def __init__(self, layers, activation='silu', device=0):
        lib = load()
        self.shapes = [(np.shape(kernel), np.shape(bias)) for kernel, bias in layers]
        widths = np.array([self.shapes[0][0][0]] + [shape[0][1] for shape in self.shapes], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate([np.concatenate([np.ravel(kernel), np.ravel(bias)]) for kernel, bias in layers]), dtype='f8')
        handle = ctypes.c_void_p()
        if lib.dl_mlp_create(ctypes.byref(handle), int(device), len(layers), _i32_ptr(widths), {'silu': 0, 'relu': 1, 'tanh': 2}[activation], _f64_ptr(flat)) != 0:
            raise LibraryError(lib.dl_last_error(None).decode())
        self._lib, self._handle, self.device = lib, handle, int(device)
        self.n_weights = int(lib.dl_mlp_info(handle, b'n_weights'))

    def _check(self, rc):
        if rc != 0: raise LibraryError(self._lib.dl_last_error(None).decode())

    def _stream(self, stream):
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream(torch.device('cuda', self.device)).cuda_stream if stream is None else stream)

    def train(self, x, y, batch, nsteps, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, return_loss=True, stream=None):
        """``nsteps`` Adam steps on consecutive chunks of ``batch`` rows of the device tensors ``x [S, n_in]``, ``y [S, n_out]``; returns the batch losses [nsteps]."""
        import torch
        assert x.is_cuda and y.is_cuda and x.is_contiguous() and y.is_contiguous() and x.dtype == y.dtype == torch.float64 and x.shape[0] == y.shape[0]
        loss = np.empty(int(nsteps), dtype='f8') if return_loss else None
        self._check(self._lib.dl_mlp_train(self._handle, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), x.shape[0], int(batch), int(nsteps), float(lr), float(beta1),
                                           float(beta2), float(eps), _f64_ptr(loss), self._stream(stream)))
        return loss

    def loss_and_grad(self, x, y, stream=None):
        loss, grad = np.empty(1, dtype='f8'), np.empty(self.n_weights, dtype='f8')
        self._check(self._lib.dl_mlp_loss_and_grad(self._handle, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), x.shape[0], _f64_ptr(loss), _f64_ptr(grad), self._stream(stream)))
        return float(loss[0]), grad

    def forward(self, x, stream=None):
        import torch
        out = torch.empty((x.shape[0], int(self._lib.dl_mlp_info(self._handle, b'n_out'))), dtype=torch.float64, device=x.device)
        self._check(self._lib.dl_mlp_forward(self._handle, ctypes.c_void_p(x.data_ptr()), x.shape[0], ctypes.c_void_p(out.data_ptr()), self._stream(stream)))
        return out

    def layers(self, stream=None):
        """Current parameters as a list of (kernel [n_in, n_out], bias [n_out])."""
        flat = np.empty(self.n_weights, dtype='f8')
        self._check(self._lib.dl_mlp_get_weights(self._handle, _f64_ptr(flat), self._stream(stream)))
        out, off = [], 0
        for kshape, bshape in self.shapes:
            nk, nb = int(np.prod(kshape)), int(np.prod(bshape))
            out.append((flat[off:off + nk].reshape(kshape).copy(), flat[off + nk:off + nk + nb].copy()))
            off += nk + nb
        return out

    def close(self):
        if getattr(self, '_handle', None):
            self._lib.dl_mlp_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceMH(object):
    """Owner of one ``dl_mh`` (include/desilike_amd.h): ``nchains`` blocked Metropolis-Hastings chains x ``vectorize`` speculative proposals per try, resident on the
    GPU of ``ctx``.  ``order``: sorted (block) position -> column of the context; ``blocks`` / ``oversample``: block sizes (slowest first) and oversampling factors."""

    def __init__(self, ctx, nchains, vectorize=1, blocks=None, oversample=None, order=None, chain_ids=None, proposal_scale=2.4, seed=0, offset=0., max_tries=1000):
        lib = load()
        if ctx.expand is not None:
            raise NotImplementedError('the device-resident sampler proposes in the columns of the device context: parameters derived by an expression need the host-driven sampler')
        P = ctx.n_params
        blocks = np.ascontiguousarray([P] if blocks is None else blocks, dtype='i4')
        oversample = np.ascontiguousarray(np.ones(len(blocks)) if oversample is None else oversample, dtype='i4')
        order = np.ascontiguousarray(np.arange(P) if order is None else order, dtype='i4')
        chain_ids = np.ascontiguousarray(np.arange(nchains) if chain_ids is None else chain_ids, dtype='i4')
        if len(oversample) != len(blocks) or len(order) != P or len(chain_ids) != nchains:
            raise ValueError('blocks / oversample / order / chain_ids have inconsistent sizes')
        i32 = ctypes.POINTER(ctypes.c_int32)
        handle = ctypes.c_void_p()
        if lib.dl_mh_create(ctypes.byref(handle), ctx._handle, int(nchains), int(vectorize), chain_ids.ctypes.data_as(i32), order.ctypes.data_as(i32), blocks.ctypes.data_as(i32),
                            oversample.ctypes.data_as(i32), len(blocks), float(proposal_scale), ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), float(offset), int(max_tries)) != 0:
            raise LibraryError(lib.dl_last_error(None).decode())
        self._lib, self._handle, self._ctx = lib, handle, ctx     # (the context must outlive the sampler)
        self.nchains, self.vectorize, self.n_params, self.device = int(nchains), int(vectorize), P, ctx.device

    def _check(self, rc):
        if rc != 0:
            raise LibraryError(self._lib.dl_last_error(None).decode())

    def info(self, key):
        return int(self._lib.dl_mh_info(self._handle, key.encode()))

    def _stream(self, stream):
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream(torch.device('cuda', self.device)).cuda_stream if stream is None else stream)

    def set_covariance(self, cholesky, stream=None):
        """Lower-triangular Cholesky factor [P, P] of the proposal covariance, parameters in sorted (block) order."""
        cholesky = np.ascontiguousarray(cholesky, dtype='f8')
        if cholesky.shape != (self.n_params,) * 2: raise ValueError('cholesky must have shape ({0:d}, {0:d})'.format(self.n_params))
        self._check(self._lib.dl_mh_set_covariance(self._handle, _f64_ptr(cholesky), self._stream(stream)))

    def set_state(self, coords, logposterior=None, weight=None, naccepted=None, tries=0, stream=None):
        coords = np.ascontiguousarray(coords, dtype='f8')
        if coords.shape != (self.nchains, self.n_params):
            raise ValueError('coords must have shape ({:d}, {:d}), found {}'.format(self.nchains, self.n_params, coords.shape))
        i64 = ctypes.POINTER(ctypes.c_int64)

        def vector(values, dtype):
            if values is None: return None
            values = np.ascontiguousarray(values, dtype=dtype)
            if values.shape != (self.nchains,): raise ValueError('expected one value per chain')
            return values

        logposterior, weight, naccepted = vector(logposterior, 'f8'), vector(weight, 'i8'), vector(naccepted, 'i8')
        self._check(self._lib.dl_mh_set_state(self._handle, _f64_ptr(coords), _f64_ptr(logposterior), None if weight is None else weight.ctypes.data_as(i64),
                                              None if naccepted is None else naccepted.ctypes.data_as(i64), int(tries), self._stream(stream)))

    def run(self, ntries, thin_by=1, stream=None):
        """Enqueue ``ntries`` tries of every chain (asynchronous); returns the device tensors (coords [nchains, ntries, P], logposterior [nchains, ntries],
        weight [nchains, ntries], count [nchains]) that hold the ``count`` states recorded by this call once the stream has run."""
        import torch
        device = torch.device('cuda', self.device)
        ntries = int(ntries)
        coords = torch.empty((self.nchains, max(ntries, 1), self.n_params), dtype=torch.float64, device=device)
        logp = torch.empty((self.nchains, max(ntries, 1)), dtype=torch.float64, device=device)
        weight = torch.empty((self.nchains, max(ntries, 1)), dtype=torch.int64, device=device)
        count = torch.zeros(self.nchains, dtype=torch.int32, device=device)
        self._check(self._lib.dl_mh_run(self._handle, ntries, int(thin_by), ctypes.c_void_p(coords.data_ptr()), ctypes.c_void_p(logp.data_ptr()), ctypes.c_void_p(weight.data_ptr()),
                                        ctypes.c_void_p(count.data_ptr()), self._stream(stream)))
        return coords, logp, weight, count

    def get_state(self, stream=None):
        """(coords [nchains, P], logposterior, weight, naccepted, consecutive failed tries) as numpy arrays; synchronises the stream."""
        coords, logp = np.empty((self.nchains, self.n_params), dtype='f8'), np.empty(self.nchains, dtype='f8')
        weight, nacc, fails = np.empty(self.nchains, dtype='i8'), np.empty(self.nchains, dtype='i8'), np.empty(self.nchains, dtype='i4')
        i64 = ctypes.POINTER(ctypes.c_int64)
        self._check(self._lib.dl_mh_get_state(self._handle, _f64_ptr(coords), _f64_ptr(logp), weight.ctypes.data_as(i64), nacc.ctypes.data_as(i64),
                                              fails.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), self._stream(stream)))
        return coords, logp, weight, nacc, fails

    def close(self):
        if getattr(self, '_handle', None):
            self._lib.dl_mh_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
