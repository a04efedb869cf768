"""ctypes wrapper of the CPU oracle (oracle/simplyp_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package."""

import ctypes as C
import os
import subprocess

import numpy as np

from simplyp_amd import abi, marshal

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libsimplyp_oracle.so')
# `make -C oracle asan-test` points this at the AddressSanitizer / UBSan build of the same source (never rebuilt from here)
LIB_OVERRIDE = os.environ.get('SIMPLYP_ORACLE_LIB')
_lib = None
HAS_F32_MIRROR = True      # integrator 3 (fp32 stages) has a same-arithmetic mirror: cashkarp_aug_f32_day


def build(force=False):
    if LIB_OVERRIDE:
        return LIB_OVERRIDE
    inc = os.path.join(os.path.dirname(HERE), 'include')
    srcs = [os.path.join(HERE, 'simplyp_oracle.c'), os.path.join(inc, 'simplyp.h'), os.path.join(inc, 'simplyp_controller.h')]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(LIB_PATH) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(['make', '-C', HERE, '-B', 'libsimplyp_oracle.so'], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not LIB_OVERRIDE and not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_OVERRIDE or LIB_PATH)
        L.simplyp_oracle_fx.restype = C.c_double
        L.simplyp_oracle_fx.argtypes = [C.c_double] * 3
        L.simplyp_oracle_run.restype = C.c_int
        _lib = L
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def fx(x, th, reld=0.01):
    return lib().simplyp_oracle_fx(float(x), float(th), float(reld))


def soilp(in10):
    a = np.ascontiguousarray(in10, dtype=np.float64)
    out = np.empty(2)
    lib().simplyp_oracle_soilp(_p(a, C.c_double), _p(out, C.c_double))
    return out


def ode_f(y12, p43):
    y = np.ascontiguousarray(y12, dtype=np.float64)
    p = np.ascontiguousarray(p43, dtype=np.float64)
    dy = np.empty(12)
    lib().simplyp_oracle_ode_f(_p(y, C.c_double), _p(p, C.c_double), _p(dy, C.c_double))
    return dy


def run(forcing, doy, member_params, reach_params, up_ptr, up_idx, opts, forcing_of_member=None,
        out_reaches=None, n_threads=1):
    """Same contract as simplyp_amd.engine.Engine.run, on host arrays.
    Returns (out[n_cols, D, n_out_reaches, E], status[E], stats dict)."""
    forcing = np.ascontiguousarray(forcing, dtype=np.float64)
    doy = np.ascontiguousarray(doy, dtype=np.int32)
    mp = np.ascontiguousarray(member_params, dtype=np.float64)
    rp = np.ascontiguousarray(reach_params, dtype=np.float64)
    up_ptr = np.ascontiguousarray(up_ptr, dtype=np.int32)
    up_idx = np.ascontiguousarray(up_idx, dtype=np.int32)
    n_sets, _, D = forcing.shape
    _, S, E = rp.shape
    dims = abi.Dims(E, S, D, n_sets)
    ncols = bin(opts.out_mask).count('1')
    if opts.out_mask & marshal.MASK_D_SNOW and not opts.snow:
        raise ValueError("column D_snow exists only with opts.snow = 1")
    if out_reaches is not None:
        out_reaches = np.ascontiguousarray(out_reaches, dtype=np.int32)
        n_or = len(out_reaches)
    else:
        n_or = S
    fom = None if forcing_of_member is None else np.ascontiguousarray(forcing_of_member, dtype=np.int32)
    out = np.full((ncols, D, n_or, E), np.nan)
    status = np.zeros(E, dtype=np.int32)
    stats = abi.Stats()
    rc = lib().simplyp_oracle_run(C.byref(dims), C.byref(opts), _p(forcing, C.c_double), _p(doy, C.c_int32),
                                  _p(fom, C.c_int32), _p(mp, C.c_double), _p(rp, C.c_double),
                                  _p(up_ptr, C.c_int32), _p(up_idx, C.c_int32), _p(out_reaches, C.c_int32),
                                  C.c_int32(n_or), _p(out, C.c_double), _p(status, C.c_int32), C.byref(stats),
                                  C.c_int(n_threads))
    if rc != 0:
        raise RuntimeError('simplyp_oracle_run failed: %d' % rc)
    return out, status, stats.as_dict()
