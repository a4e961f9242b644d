"""Property-based checks (hypothesis) of the host-side pieces: partition arithmetic, the restated
generator's structure, oracle invariants.  CPU only."""
import numpy as np
from hypothesis import given, settings, strategies as st

import sharded


@settings(max_examples=200, deadline=None)
@given(n=st.integers(1, 5_000_000), world=st.integers(1, 16))
def test_partition_covers_every_body_exactly_once(n, world):
    parts = [sharded.block_partition(n, world, r) for r in range(world)]
    block, n_alloc = parts[0][0], parts[0][3]
    assert block % 256 == 0 and n_alloc == world * block >= n and block < n / world + 256
    owned = np.zeros(n, dtype=np.int32)
    for _, ib, ic, na in parts:
        assert na == n_alloc and 0 <= ib <= n and (ib % block == 0 or ib == n)
        owned[ib:ib + ic] += 1
    assert (owned == 1).all()


@settings(max_examples=25, deadline=None)
@given(n=st.integers(1, 3000), m=st.integers(1, 3000))
def test_initial_conditions_are_prefixes_of_one_stream(nbx, n, m):
    """pos/vel of a smaller run are a prefix of a larger run's (one generator stream per array family,
    ver7/GSimulation.cpp:48,62); mass scales with (float)n (:85,92); vel = (2u - 1) * 1e-3 of the same u."""
    a, b = nbx.initial_conditions(n), nbx.initial_conditions(m)
    k = min(n, m)
    for f in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z"):
        assert np.array_equal(a[f][:k], b[f][:k])
    u = a["pos_x"]
    assert np.array_equal(a["vel_x"], (u * np.float32(2) + np.float32(-1)) * np.float32(1e-3))
    flat = np.stack([a["pos_x"], a["pos_y"], a["pos_z"]], axis=1).reshape(-1)[:n]   # mass reuses the stream: x0,y0,z0,x1,...
    assert np.array_equal(a["mass"], np.float32(n) * flat)
    assert (a["pos_x"] >= 0).all() and (a["pos_x"] < 1).all()


@settings(max_examples=15, deadline=None)
@given(n=st.integers(2, 400), seed=st.integers(0, 2**31 - 1))
def test_oracle_forces_obey_newtons_third_law_and_permutation(oracle, n, seed):
    rng = np.random.default_rng(seed)
    s = oracle.State(n)
    for f in ("pos_x", "pos_y", "pos_z"):
        getattr(s, f)[:] = rng.random(n, dtype=np.float32)
    s.mass[:] = rng.random(n, dtype=np.float32) * n
    t = s.copy()
    oracle.accel(t)
    f = t.mass.astype(np.float64) * t.acc_x.astype(np.float64)
    assert abs(f.sum()) <= 1e-5 * np.abs(f).sum() + 1e-30
    perm = rng.permutation(n)
    p = oracle.State(n)
    for fld in ("pos_x", "pos_y", "pos_z", "mass"):
        getattr(p, fld)[:] = getattr(s, fld)[perm]
    oracle.accel(p)
    scale = np.abs(t.acc_x).max() + 1e-30
    assert np.abs(p.acc_x - t.acc_x[perm]).max() <= 2e-5 * scale
