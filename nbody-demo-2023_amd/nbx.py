"""ctypes binding of libnbx.so (include/nbx.h) -- used by tests/, bench.py and __graft_entry__.

Thin by design: every method is one C-ABI call.  There is NO fallback: if libnbx.so is missing
the import of the library fails loudly, and without a HIP device nbx_create() returns
NBX_ERR_DEVICE which is raised as NbxError.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NBX_LIB") or os.path.join(_HERE, "libnbx.so")

NBX_OK, NBX_ERR_ARG, NBX_ERR_DEVICE, NBX_ERR_STATE, NBX_ERR_ALLOC = 0, -1, -2, -3, -4
ORDER_AUTO, ORDER_REFERENCE, ORDER_TREE = 0, 1, 2
LOOP_AUTO, LOOP_CXX, LOOP_ASM, LOOP_ASM_TS, LOOP_ASM_PF = 0, 1, 2, 3, 4
KERNEL_AUTO, KERNEL_LDS, KERNEL_SGPR, KERNEL_SGPRW, KERNEL_EXACT, KERNEL_EXACT_FMA, KERNEL_JLANE = 0, 1, 2, 3, 4, 5, 6

# every symbol include/nbx.h declares (tests check the library exports each of them)
SYMBOLS = (
    "nbx_last_error", "nbx_abi_version", "nbx_create", "nbx_destroy", "nbx_upload", "nbx_step",
    "nbx_step_trace", "nbx_step_local", "nbx_exchange_buffer", "nbx_commit", "nbx_kenergy_partial",
    "nbx_accel", "nbx_sync", "nbx_download", "nbx_ic_pos", "nbx_ic_vel", "nbx_ic_mass", "nbx_profile",
    "nbx_stats", "nbx_group_create", "nbx_group_destroy", "nbx_group_upload", "nbx_group_step", "nbx_group_download",
    "nbx_group_info", "nbx_partition", "nbx_comm_unique_id", "nbx_group_create_rank", "nbx_collective_timeout",
    "nbx_partition_weighted", "nbx_group_create_weighted", "nbx_group_shares", "nbx_tune_weights", "nbx_group_retune",
)


class NbxError(RuntimeError):
    def __init__(self, code, where, text):
        super().__init__("%s failed (%d): %s" % (where, code, text))
        self.code = code


class Opts(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_int32), ("device", ctypes.c_int32), ("stream", ctypes.c_void_p),
        ("i_begin", ctypes.c_int32), ("i_count", ctypes.c_int32), ("n_alloc", ctypes.c_int32),
        ("bodies_per_lane", ctypes.c_int32), ("j_split", ctypes.c_int32), ("kernel_variant", ctypes.c_int32),
        ("fused_epilogue", ctypes.c_int32), ("use_graph", ctypes.c_int32), ("external_stream", ctypes.c_int32),
        ("summation_order", ctypes.c_int32), ("inner_loop", ctypes.c_int32), ("reserved", ctypes.c_int32 * 2),
    ]


class Stats(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_int32), ("n_alloc", ctypes.c_int32), ("i_begin", ctypes.c_int32), ("i_count", ctypes.c_int32),
        ("precision", ctypes.c_int32), ("bodies_per_lane", ctypes.c_int32), ("j_split", ctypes.c_int32),
        ("j_tile", ctypes.c_int32), ("kernel_variant", ctypes.c_int32), ("fused_epilogue", ctypes.c_int32), ("summation_order", ctypes.c_int32),
        ("force_grid_x", ctypes.c_int32), ("force_grid_y", ctypes.c_int32), ("force_block", ctypes.c_int32),
        ("cu_count", ctypes.c_int32), ("clock_mhz", ctypes.c_int32), ("steps_done", ctypes.c_int64),
        ("force_launches_timed", ctypes.c_int64), ("force_ms_total", ctypes.c_double),
        ("pairs_per_launch", ctypes.c_double), ("device_name", ctypes.c_char * 64),
        ("graph_replays", ctypes.c_int64), ("use_graph", ctypes.c_int32), ("inner_loop", ctypes.c_int32),
    ]

    def asdict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["device_name"] = self.device_name.decode(errors="replace")
        return d


_lib = None


def load():
    """Load libnbx.so (raises OSError with a build hint if it is not there)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("%s not found: build it with `make lib` (or __graft_entry__.build()); "
                      "there is no fallback path" % LIB_PATH)
    if "torch" not in sys.modules and not os.environ.get("NBX_NO_TORCH_PRELOAD"):
        # The PyTorch-ROCm wheel bundles its own libamdhip64 / libhsa-runtime64.  Two copies of the
        # ROCm runtime cannot both initialise in one process (the second one sees no GPU), so when
        # torch is installed let it load first: libnbx.so then binds to the copy torch brought
        # (same sonames).  Stand-alone consumers (nbody.x) use /opt/rocm's copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, dbl = ctypes.c_void_p, ctypes.c_int32, ctypes.c_double
    L.nbx_last_error.restype = ctypes.c_char_p
    L.nbx_abi_version.restype = i32
    L.nbx_create.argtypes = [ctypes.POINTER(vp), i32, i32, ctypes.POINTER(Opts)]
    L.nbx_destroy.argtypes = [vp]
    L.nbx_destroy.restype = None
    L.nbx_upload.argtypes = [vp] + [vp] * 7
    L.nbx_step.argtypes = [vp, dbl, i32, ctypes.POINTER(dbl)]
    L.nbx_step_trace.argtypes = [vp, dbl, i32, vp]
    L.nbx_step_local.argtypes = [vp, dbl]
    L.nbx_exchange_buffer.argtypes = [vp, ctypes.POINTER(vp)] + [ctypes.POINTER(ctypes.c_size_t)] * 3
    L.nbx_commit.argtypes = [vp]
    L.nbx_kenergy_partial.argtypes = [vp, ctypes.POINTER(dbl)]
    L.nbx_accel.argtypes = [vp, vp, vp, vp]
    L.nbx_sync.argtypes = [vp]
    L.nbx_download.argtypes = [vp] + [vp] * 6
    L.nbx_ic_pos.argtypes = [i32, i32, vp, vp, vp]
    L.nbx_ic_vel.argtypes = [i32, i32, vp, vp, vp]
    L.nbx_ic_mass.argtypes = [i32, i32, vp]
    L.nbx_profile.argtypes = [vp, i32]
    L.nbx_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    L.nbx_group_create.argtypes = [ctypes.POINTER(vp), i32, i32, i32, ctypes.POINTER(i32), ctypes.POINTER(Opts)]
    L.nbx_group_destroy.argtypes = [vp]
    L.nbx_group_destroy.restype = None
    L.nbx_group_upload.argtypes = [vp] + [vp] * 7
    L.nbx_group_step.argtypes = [vp, dbl, i32, ctypes.POINTER(dbl)]
    L.nbx_group_download.argtypes = [vp] + [vp] * 6
    L.nbx_group_info.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), i32, ctypes.POINTER(Stats)]
    L.nbx_partition.argtypes = [i32, i32, i32] + [ctypes.POINTER(i32)] * 5
    L.nbx_comm_unique_id.argtypes = [vp]
    L.nbx_group_create_rank.argtypes = [ctypes.POINTER(vp), i32, i32, i32, i32, vp, i32, ctypes.POINTER(Opts)]
    L.nbx_collective_timeout.argtypes = [dbl]
    pd = ctypes.POINTER(dbl)
    L.nbx_partition_weighted.argtypes = [i32, i32, pd, i32] + [ctypes.POINTER(i32)] * 4
    L.nbx_group_create_weighted.argtypes = [ctypes.POINTER(vp), i32, i32, i32, ctypes.POINTER(i32), pd, ctypes.POINTER(Opts)]
    L.nbx_group_shares.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), pd]
    L.nbx_tune_weights.argtypes = [i32, ctypes.POINTER(i32), pd, pd]
    L.nbx_group_retune.argtypes = [vp, pd, ctypes.POINTER(i32)]
    _lib = L
    return L


def _check(rc, where):
    if rc != NBX_OK:
        raise NbxError(rc, where, load().nbx_last_error().decode(errors="replace"))


def _dtype(precision):
    if precision == 32:
        return np.float32
    if precision == 64:
        return np.float64
    raise NbxError(NBX_ERR_ARG, "precision", "must be 32 or 64")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


DT = float(np.float32(0.1))  # (float)0.1 widened: the reference's _tstep (ver7/GSimulation.cpp:30)

FIELDS = ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "mass")


def initial_conditions(n, precision=32):
    """Seed-42 particles of ver7/GSimulation.cpp:45-94 as a dict of seven arrays."""
    L = load()
    dt = _dtype(precision)
    s = {f: np.zeros(max(n, 0), dtype=dt) for f in FIELDS}
    _check(L.nbx_ic_pos(n, precision, _ptr(s["pos_x"]), _ptr(s["pos_y"]), _ptr(s["pos_z"])), "nbx_ic_pos")
    _check(L.nbx_ic_vel(n, precision, _ptr(s["vel_x"]), _ptr(s["vel_y"]), _ptr(s["vel_z"])), "nbx_ic_vel")
    _check(L.nbx_ic_mass(n, precision, _ptr(s["mass"])), "nbx_ic_mass")
    return s


class Context:
    """One nbx_ctx.  Keyword options are the nbx_opts fields."""

    def __init__(self, n, precision=32, **opts):
        self._L = load()
        self._h = ctypes.c_void_p()
        self.n = int(n)
        self.precision = int(precision)
        o = Opts()
        o.struct_size = ctypes.sizeof(Opts)
        o.device = -1
        for k, v in opts.items():
            if not hasattr(o, k):
                raise TypeError("unknown nbx_opts field %r" % k)
            setattr(o, k, v)
        _check(self._L.nbx_create(ctypes.byref(self._h), self.n, self.precision, ctypes.byref(o)), "nbx_create")
        self.dtype = _dtype(self.precision)

    def close(self):
        if self._h:
            self._L.nbx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _arr(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.shape != (self.n,):
            raise NbxError(NBX_ERR_ARG, "array", "expected shape (%d,), got %r" % (self.n, a.shape))
        return a

    def upload(self, state):
        arrs = [self._arr(state[f]) for f in FIELDS]
        _check(self._L.nbx_upload(self._h, *[_ptr(a) for a in arrs]), "nbx_upload")

    def step(self, nsteps, dt=DT, kenergy=True):
        ke = ctypes.c_double(0.0)
        _check(self._L.nbx_step(self._h, dt, nsteps, ctypes.byref(ke) if kenergy else None), "nbx_step")
        return ke.value if kenergy else None

    def step_trace(self, nsteps, dt=DT):
        ke = np.zeros(max(nsteps, 1), dtype=np.float64)
        _check(self._L.nbx_step_trace(self._h, dt, nsteps, _ptr(ke)), "nbx_step_trace")
        return ke[:nsteps]

    def step_local(self, dt=DT):
        _check(self._L.nbx_step_local(self._h, dt), "nbx_step_local")

    def exchange_buffer(self):
        p = ctypes.c_void_p()
        tot, off, own = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        _check(self._L.nbx_exchange_buffer(self._h, ctypes.byref(p), ctypes.byref(tot), ctypes.byref(off),
                                           ctypes.byref(own)), "nbx_exchange_buffer")
        return p.value, tot.value, off.value, own.value

    def commit(self):
        _check(self._L.nbx_commit(self._h), "nbx_commit")

    def kenergy_partial(self):
        s = ctypes.c_double(0.0)
        _check(self._L.nbx_kenergy_partial(self._h, ctypes.byref(s)), "nbx_kenergy_partial")
        return s.value

    def accel(self):
        a = [np.zeros(self.n, dtype=self.dtype) for _ in range(3)]
        _check(self._L.nbx_accel(self._h, *[_ptr(x) for x in a]), "nbx_accel")
        return a

    def sync(self):
        _check(self._L.nbx_sync(self._h), "nbx_sync")

    def download(self):
        out = {f: np.zeros(self.n, dtype=self.dtype) for f in FIELDS[:6]}
        _check(self._L.nbx_download(self._h, *[_ptr(out[f]) for f in FIELDS[:6]]), "nbx_download")
        return out

    def profile(self, enable=True):
        _check(self._L.nbx_profile(self._h, 1 if enable else 0), "nbx_profile")

    def stats(self):
        s = Stats()
        _check(self._L.nbx_stats(self._h, ctypes.byref(s)), "nbx_stats")
        return s.asdict()


class Group:
    """One nbx_group: n_ranks contexts driven by this process (multi-GPU; logical ranks when devices repeat)."""

    def __init__(self, n, precision=32, n_ranks=1, devices=None, rank=None, unique_id=None, device=-1, weights=None, weighted=False, **opts):
        """Single process: n_ranks contexts on `devices`.  One process per GPU: pass rank= and unique_id= (the 128 bytes
        of unique_id() made on rank 0 and shipped to every rank); n_ranks is then the world size and every call on the
        group is collective (nbx_group_create_rank).  weights= (or weighted=True for equal weights): unequal shares in whole
        256-record tiles (nbx_group_create_weighted), which retune() can move."""
        self._L = load()
        self._h = ctypes.c_void_p()
        self.n, self.precision, self.dtype = int(n), int(precision), _dtype(precision)
        o = Opts()
        o.struct_size = ctypes.sizeof(Opts)
        for k, v in opts.items():
            setattr(o, k, v)
        if rank is not None:
            buf = ctypes.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
            _check(self._L.nbx_group_create_rank(ctypes.byref(self._h), self.n, self.precision, n_ranks, rank, buf, device, ctypes.byref(o)),
                   "nbx_group_create_rank")
            return
        dev = None if devices is None else (ctypes.c_int32 * len(devices))(*devices)
        if weights is not None or weighted:
            w = None if weights is None else (ctypes.c_double * len(weights))(*weights)
            _check(self._L.nbx_group_create_weighted(ctypes.byref(self._h), self.n, self.precision, n_ranks, dev, w, ctypes.byref(o)),
                   "nbx_group_create_weighted")
            return
        _check(self._L.nbx_group_create(ctypes.byref(self._h), self.n, self.precision, n_ranks, dev, ctypes.byref(o)),
               "nbx_group_create")

    def close(self):
        if self._h:
            self._L.nbx_group_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def upload(self, state):
        arrs = [np.ascontiguousarray(state[f], dtype=self.dtype) for f in FIELDS]
        _check(self._L.nbx_group_upload(self._h, *[_ptr(a) for a in arrs]), "nbx_group_upload")

    def step(self, nsteps, dt=DT, kenergy=True):
        ke = ctypes.c_double(0.0)
        _check(self._L.nbx_group_step(self._h, dt, nsteps, ctypes.byref(ke) if kenergy else None), "nbx_group_step")
        return ke.value if kenergy else None

    def download(self):
        out = {f: np.zeros(self.n, dtype=self.dtype) for f in FIELDS[:6]}
        _check(self._L.nbx_group_download(self._h, *[_ptr(out[f]) for f in FIELDS[:6]]), "nbx_group_download")
        return out

    def shares(self, timings=True):
        """nbx_group_shares: (i_begin list, i_count list, mean force-kernel ms per rank since the last retune)."""
        P = self.info(0)[0]
        b, c, ms = (ctypes.c_int32 * P)(), (ctypes.c_int32 * P)(), (ctypes.c_double * P)()
        _check(self._L.nbx_group_shares(self._h, b, c, ms if timings else None), "nbx_group_shares")
        return list(b), list(c), list(ms)

    def retune(self, force_ms=None):
        """nbx_group_retune: new shares from the measured (or the given) per-rank force-kernel times; True if they moved."""
        ch = ctypes.c_int32(0)
        ms = None if force_ms is None else (ctypes.c_double * len(force_ms))(*force_ms)
        _check(self._L.nbx_group_retune(self._h, ms, ctypes.byref(ch)), "nbx_group_retune")
        return bool(ch.value)

    def info(self, rank=0):
        P, rccl, st = ctypes.c_int32(), ctypes.c_int32(), Stats()
        _check(self._L.nbx_group_info(self._h, ctypes.byref(P), ctypes.byref(rccl), rank, ctypes.byref(st)), "nbx_group_info")
        return P.value, bool(rccl.value), st.asdict()


UNIQUE_ID_BYTES = 128


def unique_id():
    """nbx_comm_unique_id: the rendezvous token rank 0 creates for a one-process-per-GPU group."""
    buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
    _check(load().nbx_comm_unique_id(buf), "nbx_comm_unique_id")
    return buf.raw


EXIT_COLLECTIVE_TIMEOUT = 75


def collective_timeout(seconds):
    """nbx_collective_timeout: the watchdog's bound on blocking group collectives (<= 0: off)."""
    _check(load().nbx_collective_timeout(float(seconds)), "nbx_collective_timeout")


def partition(n, n_ranks, rank):
    """nbx_partition: (ranks_used, block, i_begin, i_count, n_alloc) of the library's block partition."""
    out = [ctypes.c_int32() for _ in range(5)]
    _check(load().nbx_partition(n, n_ranks, rank, *[ctypes.byref(o) for o in out]), "nbx_partition")
    return tuple(o.value for o in out)


def partition_weighted(n, n_ranks, weights, rank):
    """nbx_partition_weighted: (ranks_used, i_begin, i_count, n_alloc); weights None = equal."""
    out = [ctypes.c_int32() for _ in range(4)]
    w = None if weights is None else (ctypes.c_double * len(weights))(*weights)
    _check(load().nbx_partition_weighted(n, n_ranks, w, rank, *[ctypes.byref(o) for o in out]), "nbx_partition_weighted")
    return tuple(o.value for o in out)


def tune_weights(i_count, force_ms):
    """nbx_tune_weights: the tuner's arithmetic (each rank's measured bodies per millisecond, normalised)."""
    P = len(i_count)
    out = (ctypes.c_double * P)()
    _check(load().nbx_tune_weights(P, (ctypes.c_int32 * P)(*i_count), (ctypes.c_double * P)(*force_ms), out), "nbx_tune_weights")
    return list(out)


def read_snapshot(path):
    """Read an NBXSNAP1 file written by nbody.x (NBODY_SNAPSHOT=...): returns (state dict, steps_done)."""
    import struct
    with open(path, "rb") as f:
        magic, n, prec, steps = struct.unpack("<8siiq", f.read(24))
        if magic != b"NBXSNAP1":
            raise ValueError("%s is not an NBXSNAP1 snapshot" % path)
        dt = _dtype(prec)
        state = {k: np.frombuffer(f.read(n * dt().itemsize), dtype=dt).copy() for k in FIELDS}
    return state, steps
