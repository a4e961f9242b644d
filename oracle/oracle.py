"""ctypes loader for oracle/liboracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The functions restate the reference's ver7 hot path on the CPU (see the header of
oracle/nbody_oracle.c for the file:line anchors).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_c_int = ctypes.c_int


def build(force=False):
    """Compile the C restatement with the pinned flags (gcc -O2 -fopenmp, no FMA contraction)."""
    src = os.path.join(_HERE, "nbody_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.orc_mt19937_first.argtypes = [ctypes.c_uint32]
        L.orc_mt19937_first.restype = ctypes.c_uint32
        L.orc_init_pos.argtypes = [_c_int, _f32p, _f32p, _f32p]
        L.orc_init_vel.argtypes = [_c_int, _f32p, _f32p, _f32p]
        L.orc_init_mass.argtypes = [_c_int, _f32p]
        L.orc_accel_f32.argtypes = [_c_int, _c_int, _c_int] + [_f32p] * 7
        L.orc_integrate_f32.argtypes = [_c_int, _c_int, ctypes.c_float] + [_f32p] * 10
        L.orc_integrate_f32.restype = ctypes.c_float
        L.orc_kenergy_from_sum_f32.argtypes = [ctypes.c_float]
        L.orc_kenergy_from_sum_f32.restype = ctypes.c_float
        L.orc_run_f32.argtypes = [_c_int, _c_int, ctypes.c_float] + [_f32p] * 10 + [ctypes.c_void_p]
        L.orc_accel_f64.argtypes = [_c_int, _c_int, _c_int] + [_f64p] * 7
        L.orc_integrate_f64.argtypes = [_c_int, _c_int, ctypes.c_double] + [_f64p] * 10
        L.orc_integrate_f64.restype = ctypes.c_double
        L.orc_run_f64.argtypes = [_c_int, _c_int, ctypes.c_double] + [_f64p] * 10 + [ctypes.c_void_p]
        L.orc_default_dt_f32.restype = ctypes.c_float
        _lib = L
    return _lib


DT_F32 = np.float32(0.1)  # (float)0.1, ver7/GSimulation.cpp:30,99


class State:
    """The reference's ParticleSoA (ver7/Particle.hpp:43-58) as ten numpy arrays."""

    FIELDS = ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "acc_x", "acc_y", "acc_z", "mass")

    def __init__(self, n, dtype=np.float32):
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        for f in self.FIELDS:
            setattr(self, f, np.zeros(self.n, dtype=self.dtype))

    def arrays(self):
        return [getattr(self, f) for f in self.FIELDS]

    def copy(self):
        s = State(self.n, self.dtype)
        for f in self.FIELDS:
            getattr(s, f)[:] = getattr(self, f)
        return s

    def astype(self, dtype):
        s = State(self.n, dtype)
        for f in self.FIELDS:
            getattr(s, f)[:] = getattr(self, f).astype(dtype)
        return s


def init_state(n):
    """Seed-42 initial conditions, ver7/GSimulation.cpp:45-94 (always fp32-drawn)."""
    s = State(n, np.float32)
    L = lib()
    if n > 0:
        L.orc_init_pos(n, s.pos_x, s.pos_y, s.pos_z)
        L.orc_init_vel(n, s.vel_x, s.vel_y, s.vel_z)
        L.orc_init_mass(n, s.mass)
    return s


def accel(s, i0=0, i1=None):
    """Accumulate accelerations of bodies [i0,i1) into s.acc_* (ver7:141-177)."""
    i1 = s.n if i1 is None else i1
    L = lib()
    fn = L.orc_accel_f32 if s.dtype == np.float32 else L.orc_accel_f64
    fn(s.n, i0, i1, s.pos_x, s.pos_y, s.pos_z, s.mass, s.acc_x, s.acc_y, s.acc_z)


def integrate(s, dt=None, i0=0, i1=None):
    """Euler update + energy sum of bodies [i0,i1) (ver7:178-198); returns sum m*v^2."""
    i1 = s.n if i1 is None else i1
    L = lib()
    if s.dtype == np.float32:
        dt = DT_F32 if dt is None else np.float32(dt)
        return L.orc_integrate_f32(i0, i1, dt, s.pos_x, s.pos_y, s.pos_z, s.vel_x, s.vel_y, s.vel_z,
                                   s.acc_x, s.acc_y, s.acc_z, s.mass)
    dt = float(DT_F32) if dt is None else float(dt)
    return L.orc_integrate_f64(i0, i1, dt, s.pos_x, s.pos_y, s.pos_z, s.vel_x, s.vel_y, s.vel_z,
                               s.acc_x, s.acc_y, s.acc_z, s.mass)


def run(s, nsteps, dt=None):
    """nsteps reference time steps in place; returns the per-step kinetic-energy trace."""
    L = lib()
    if s.dtype == np.float32:
        ke = np.zeros(max(nsteps, 1), dtype=np.float32)
        dt = DT_F32 if dt is None else np.float32(dt)
        L.orc_run_f32(s.n, nsteps, dt, *s.arrays(), ke.ctypes.data)
    else:
        ke = np.zeros(max(nsteps, 1), dtype=np.float64)
        dt = float(DT_F32) if dt is None else float(dt)
        L.orc_run_f64(s.n, nsteps, dt, *s.arrays(), ke.ctypes.data)
    return ke[:nsteps]
