"""ctypes loader for the CPU oracle (oracle/nb_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libnb_oracle.so")

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")

EPS2 = 1e-4  # nbody3d.js:234


def build(force=False):
    src = os.path.join(_HERE, "nb_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        u32, f32, f64, i32 = C.c_uint32, C.c_float, C.c_double, C.c_int
        L.nbo_accel_f32.argtypes = [_f32p, u32, f32, f32, u32, u32, _f32p]
        L.nbo_accel_f32.restype = None
        L.nbo_run_f32.argtypes = [_f32p, _f32p, _f32p, u32, f32, f32, f32, u32]
        L.nbo_run_f32.restype = i32
        L.nbo_integrate_range_f32.argtypes = [_f32p, _f32p, _f32p, _f32p, u32, u32, f32]
        L.nbo_integrate_range_f32.restype = None
        L.nbo_accel_f64.argtypes = [_f64p, u32, f64, f64, u32, u32, _f64p]
        L.nbo_accel_f64.restype = None
        L.nbo_run_f64.argtypes = [_f64p, _f64p, _f64p, u32, f64, f64, f64, u32]
        L.nbo_run_f64.restype = i32
        L.nbo_energy_f64.argtypes = [_f64p, _f64p, u32, f64, f64, _f64p]
        L.nbo_energy_f64.restype = None
        L.nbo_accel_f32_mt.argtypes = [_f32p, u32, f32, f32, u32, u32, _f32p, i32]
        L.nbo_accel_f32_mt.restype = i32
        L.nbo_max_threads.restype = i32
        _lib = L
    return _lib


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def accel_f32(bodies, G, eps2=EPS2, i0=0, i1=None):
    b = _c32(bodies).reshape(-1, 4)
    n = b.shape[0]
    i1 = n if i1 is None else i1
    out = np.zeros((i1 - i0, 4), np.float32)
    lib().nbo_accel_f32(b, n, G, eps2, i0, i1, out)
    return out


def accel_f64(bodies, G, eps2=EPS2, i0=0, i1=None):
    b = _c64(bodies).reshape(-1, 4)
    n = b.shape[0]
    i1 = n if i1 is None else i1
    out = np.zeros((i1 - i0, 4), np.float64)
    lib().nbo_accel_f64(b, n, G, eps2, i0, i1, out)
    return out


def run_f32(bodies, vel, accel, dt, G, nsteps, eps2=EPS2):
    """Returns NEW (bodies, vel, accel) after nsteps reference-semantics steps."""
    b = _c32(bodies).reshape(-1, 4).copy()
    v = _c32(vel).reshape(-1, 4).copy()
    a = np.zeros_like(b) if accel is None else _c32(accel).reshape(-1, 4).copy()
    rc = lib().nbo_run_f32(b, v, a, b.shape[0], dt, G, eps2, nsteps)
    assert rc == 0
    return b, v, a


def run_f64(bodies, vel, accel, dt, G, nsteps, eps2=EPS2):
    b = _c64(bodies).reshape(-1, 4).copy()
    v = _c64(vel).reshape(-1, 4).copy()
    a = np.zeros_like(b) if accel is None else _c64(accel).reshape(-1, 4).copy()
    rc = lib().nbo_run_f64(b, v, a, b.shape[0], dt, G, eps2, nsteps)
    assert rc == 0
    return b, v, a


def integrate_range_f32(bodies, vel, accel, a_new, i0, i1, dt):
    """In place on the given float32 arrays (shard tests)."""
    lib().nbo_integrate_range_f32(bodies, vel, accel, _c32(a_new), i0, i1, dt)


def energy(bodies, vel, G, eps2=EPS2):
    """fp64 (kinetic, potential, momentum[3]).  Pair KE(vel after call n) with
    PE(positions BEFORE call n): SURVEY.md §8(c) energy note."""
    b = _c64(bodies).reshape(-1, 4)
    v = _c64(vel).reshape(-1, 4)
    out = np.zeros(5, np.float64)
    lib().nbo_energy_f64(b, v, b.shape[0], G, eps2, out)
    return out[0], out[1], out[2:5].copy()


def accel_f32_mt(bodies, G, eps2=EPS2, i0=0, i1=None, nthreads=0):
    """The timed CPU baseline kernel.  Returns (acc, threads_used)."""
    b = _c32(bodies).reshape(-1, 4)
    n = b.shape[0]
    i1 = n if i1 is None else i1
    out = np.zeros((i1 - i0, 4), np.float32)
    used = lib().nbo_accel_f32_mt(b, n, G, eps2, i0, i1, out, nthreads)
    assert used > 0
    return out, used
