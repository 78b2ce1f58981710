"""CPU (-m "not gpu"): the C-ABI shared library loads without a GPU and exports every entry point include/desilike_amd.h declares;
the ctypes host binds exactly that set; the host-only part of the ABI (configuration store, error reporting) works; nothing computes without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, 'include', 'desilike_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dl_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported_and_bound():
    from desilike_amd import _lib
    names = declared_functions()
    assert len(names) >= 14 and 'dl_eval_batch' in names and 'dl_eval_logposterior' in names and 'dl_eval_batch_derived' in names
    lib = ctypes.CDLL(_lib.lib_path())
    for name in names:
        assert hasattr(lib, name), 'libdesilike_amd.so does not export {}'.format(name)
    assert sorted(_lib.SYMBOLS) == names, 'ctypes table and header disagree: {}'.format(set(_lib.SYMBOLS) ^ set(names))


def test_config_store_and_errors_without_gpu():
    import torch
    from desilike_amd import _lib
    lib = _lib.load()
    cfg = lib.dl_config_new()
    assert cfg
    a = np.arange(3, dtype='f8')
    assert lib.dl_config_set_f64(cfg, b'priors', a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), a.size) == 0
    i = np.array([1], dtype='i4')
    assert lib.dl_config_set_i32(cfg, b'n_params', i.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), 1) == 0
    if not torch.cuda.is_available():
        handle = ctypes.c_void_p()
        rc = lib.dl_create(ctypes.byref(handle), 0, cfg)   # incomplete configuration and no GPU: must fail, never compute
        assert rc != 0 and not handle.value
        assert len(lib.dl_last_error(None)) > 0
    lib.dl_config_free(cfg)
    # null-context calls are errors, not crashes
    assert lib.dl_eval_batch(None, None, 1, None, None, None, None, None, None) != 0
    assert lib.dl_eval_logposterior(None, None, 1, None, None, None) != 0
    assert lib.dl_info(None, b'n_params') == -1
    assert lib.dl_fftlog_apply(None, None, 1, None, None) != 0
    if not torch.cuda.is_available():   # no GPU: the FFTLog plan cannot be created either (no host fallback behind the ABI)
        from desilike_amd.fftlog import PowerToCorrelation
        with pytest.raises(_lib.LibraryError):
            PowerToCorrelation(np.logspace(-3., 1., 64), ell=(0,), engine='hip', device=0)(np.ones((1, 64)))
