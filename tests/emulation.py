"""Build + ctypes-load the CPU emulation of the theory kernel phases (tests/csrc/emulate.cpp): test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_emu = None
SANITIZE_FLAGS = ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer', '-g']


def build_emulation(sanitize=False):
    """Compile tests/csrc/emulate.cpp (the device phase functions of desilike_amd/csrc/dl_fullshape.h + the host-side constant folding of dl_host.hpp, run on the CPU);
    ``sanitize``: the AddressSanitizer + UndefinedBehaviorSanitizer build (CPU only: GPU sanitizers are not available on this pool), to be loaded in a process started
    with libasan preloaded (tests/test_emulation.py::test_emulation_under_sanitizers)."""
    build = os.path.join(HERE, 'csrc', '_build')
    os.makedirs(build, exist_ok=True)
    so = os.path.join(build, 'libdl_emulate_asan.so' if sanitize else 'libdl_emulate.so')
    src = os.path.join(HERE, 'csrc', 'emulate.cpp')
    deps = [src] + [os.path.join(HERE, '..', 'desilike_amd', 'csrc', name) for name in ['dl_fullshape.h', 'dl_fullshape_grad.h', 'dl_host.hpp', 'dl_tns.h']]
    if not os.path.isfile(so) or any(os.path.getmtime(dep) > os.path.getmtime(so) for dep in deps):
        subprocess.check_call(['g++', '-O1' if sanitize else '-O2', '-std=c++17', '-fPIC', '-shared'] + (SANITIZE_FLAGS if sanitize else []) + ['-o', so, src])
    return so


def load_emulation():
    global _emu
    if _emu is None:
        so = build_emulation(sanitize=os.environ.get('DL_EMULATION_SANITIZE', '0') == '1')
        lib = ctypes.CDLL(so)
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
        lib.emu_config_new.restype = ctypes.c_void_p
        lib.emu_config_free.argtypes = [ctypes.c_void_p]
        lib.emu_config_set_f64.argtypes = [ctypes.c_void_p, ctypes.c_char_p, dp, ctypes.c_int64]
        lib.emu_config_set_i32.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ip, ctypes.c_int64]
        lib.emu_last_error.restype = ctypes.c_char_p
        lib.emu_eval_theory.argtypes = [ctypes.c_void_p, dp, ctypes.c_int64, ctypes.c_int, dp, dp]
        lib.emu_eval_batch.argtypes = [ctypes.c_void_p, dp, ctypes.c_int64, dp, dp]
        lib.emu_eval_grad.argtypes = [ctypes.c_void_p, dp, ctypes.c_int64, dp, dp]
        lib.emu_tns_tables.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int, dp, dp, ctypes.c_int, dp, dp]
        lib.emu_tns_combine.argtypes = [ctypes.c_double] * 5 + [dp]
        _emu = lib
    return _emu


class Emulation(object):

    def __init__(self, spec):
        from desilike_amd._lib import fill_config
        self.lib = load_emulation()
        self.cfg = self.lib.emu_config_new()
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
        fill_config(spec, lambda key, a: self.lib.emu_config_set_f64(self.cfg, key.encode(), a.ctypes.data_as(dp), a.size),
                    lambda key, a: self.lib.emu_config_set_i32(self.cfg, key.encode(), a.ctypes.data_as(ip), a.size))
        self.spec = spec

    def eval_theory(self, theta, iobs=0):
        theta = np.ascontiguousarray(theta, dtype='f8')
        obs = self.spec['observables'][iobs]
        n_ell, n_kin = len(obs['ells_in']), len(obs['kin'])
        power, tables = np.empty((len(theta), n_ell, n_kin)), np.empty((len(theta), 3, n_ell, n_kin))
        dp = ctypes.POINTER(ctypes.c_double)
        if self.lib.emu_eval_theory(self.cfg, theta.ctypes.data_as(dp), len(theta), iobs, power.ctypes.data_as(dp), tables.ctypes.data_as(dp)):
            raise RuntimeError(self.lib.emu_last_error().decode())
        return power, tables

    def eval_batch(self, theta):
        theta = np.ascontiguousarray(theta, dtype='f8')
        n = sum(len(obs['flatdata']) for obs in self.spec['observables'])
        loglike, flat = np.empty(len(theta)), np.empty((len(theta), n))
        dp = ctypes.POINTER(ctypes.c_double)
        if self.lib.emu_eval_batch(self.cfg, theta.ctypes.data_as(dp), len(theta), loglike.ctypes.data_as(dp), flat.ctypes.data_as(dp)):
            raise RuntimeError(self.lib.emu_last_error().decode())
        return loglike, flat

    def eval_grad(self, theta):
        """(loglike [B], gradient [B, P]) by the device's analytic gradient phases (csrc/dl_fullshape_grad.h); None if the configuration is out of their scope."""
        theta = np.ascontiguousarray(theta, dtype='f8')
        loglike, grad = np.empty(len(theta)), np.empty(theta.shape)
        dp = ctypes.POINTER(ctypes.c_double)
        rc = self.lib.emu_eval_grad(self.cfg, theta.ctypes.data_as(dp), len(theta), loglike.ctypes.data_as(dp), grad.ctypes.data_as(dp))
        if rc == 2: return None
        if rc: raise RuntimeError(self.lib.emu_last_error().decode())
        return loglike, grad

    def __del__(self):
        self.lib.emu_config_free(self.cfg)


def tns_tables(k11, q, mus, wmus, pk):
    """The 29 TNS loop tables [29, n_k11] of one template, by the device's own functions (csrc/dl_tns.h) run sequentially on the CPU."""
    lib = load_emulation()
    k11, q, mus, wmus, pk = (np.ascontiguousarray(a, dtype='f8') for a in (k11, q, mus, wmus, pk))
    out = np.empty((29, len(k11)), dtype='f8')
    dp = ctypes.POINTER(ctypes.c_double)
    outside = lib.emu_tns_tables(k11.ctypes.data_as(dp), len(k11), q.ctypes.data_as(dp), len(q), mus.ctypes.data_as(dp), wmus.ctypes.data_as(dp), len(mus), pk.ctypes.data_as(dp), out.ctypes.data_as(dp))
    return out, outside


def tns_combine(f, b1, b2, bs, b3):
    lib = load_emulation()
    out = np.empty((6, 32), dtype='f8')
    lib.emu_tns_combine(f, b1, b2, bs, b3, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return out
