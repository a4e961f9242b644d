"""Unequal shares (include/nbx.h: nbx_partition_weighted, nbx_tune_weights) -- host arithmetic, no GPU.

The reference's co-execution split gives device 0 `n * cpu_ratio` bodies and the other device the rest, and in tuning mode steps
the ratio once per print window (ver5_all/programming_models/opencl/Compute.cpp:154-162,241-255,317-321).  The GPU-native reading:
every device is a GPU, shares are whole 256-record tiles in proportion to a weight, and the tuner sets each weight to the rate the
device measured (bodies per millisecond of force kernel).  Here: the partition's invariants, and -- with synthetic per-rank speeds
standing in for eight GPUs that hold clocks 10 % apart -- the skew one retune removes.
"""
import numpy as np
import pytest


def shares(nbx, n, P, w):
    out = [nbx.partition_weighted(n, P, w, r) for r in range(P)]
    used, n_alloc = out[0][0], out[0][3]
    assert all(o[0] == used and o[3] == n_alloc for o in out)
    return used, [o[1] for o in out], [o[2] for o in out], n_alloc


@pytest.mark.parametrize("n,P,w", [(1048576, 8, None), (1048576, 8, [1, 1, 1, 1, 1, 1, 1, 0.9]), (262144, 4, [1, 2, 1, 3]), (5001, 3, [1, 2, 1]),
                                   (2000, 8, [5, 1, 1, 1, 1, 1, 1, 1]), (300, 8, None), (256, 2, [1, 1]), (257, 2, [1, 1000]), (1, 4, None),
                                   (4099, 5, [0.3, 0.1, 0.2, 0.25, 0.15])])
def test_every_body_has_one_owner_and_blocks_are_whole_tiles(nbx, n, P, w):
    used, b, c, n_alloc = shares(nbx, n, P, w)
    tiles = -(-n // 256)
    assert used == min(P, tiles) and n_alloc == 256 * tiles
    assert b[0] == 0 and all(c[r] > 0 for r in range(used)) and all(c[r] == 0 for r in range(used, P))
    for r in range(used):
        assert b[r] % 256 == 0
        assert b[r] + c[r] == (b[r + 1] if r + 1 < used else n)               # contiguous, nothing twice, nothing left out
        assert c[r] % 256 == 0 or r == used - 1                                 # only the last block may be ragged
    ww = np.array(w[:used] if w else [1.0] * used, dtype=float)
    ideal = tiles * ww / ww.sum()
    got = np.array([-(-c[r] // 256) for r in range(used)])
    assert (np.abs(got - ideal) < 1.0 + 1e-9).all() or (got[np.abs(got - ideal) >= 1.0] == 1).all()   # within one tile of the ideal share (or at the one-tile floor)


def test_equal_weights_of_a_divisible_problem_are_the_equal_blocks(nbx):
    for P in (2, 4, 8):
        _, b, c, n_alloc = shares(nbx, 1048576, P, None)
        assert c == [1048576 // P] * P and n_alloc == 1048576
        assert [nbx.partition(1048576, P, r)[2:4] for r in range(P)] == list(zip(b, c))   # = nbx_partition where that one is exact


def test_bad_weights_are_refused(nbx):
    for w in ([1, 0, 1], [1, -1, 1], [1, float("nan"), 1], [1, float("inf"), 1]):
        with pytest.raises(nbx.NbxError) as e:
            nbx.partition_weighted(5000, 3, w, 0)
        assert e.value.code == nbx.NBX_ERR_ARG
    with pytest.raises(nbx.NbxError):
        nbx.tune_weights([100, 0], [1.0, 1.0])
    with pytest.raises(nbx.NbxError):
        nbx.tune_weights([100, 100], [1.0, 0.0])


def test_tuner_weights_are_the_measured_rates(nbx):
    w = nbx.tune_weights([1000, 1000, 2000], [1.0, 2.0, 1.0])
    assert abs(sum(w) - 1.0) < 1e-15 and np.allclose(w, np.array([1000, 500, 2000]) / 3500.0)


def test_one_retune_removes_the_skew_of_devices_that_hold_different_clocks(nbx):
    """Eight synthetic devices whose speed differs like the boxes of the pool did under this kernel (0.553 ... 0.609 of the roofline,
    one binary): with equal blocks every step lasts as long as the slowest device needs; after ONE pass of measure -> nbx_tune_weights
    -> nbx_partition_weighted all devices finish within one 256-record tile of each other."""
    n, P = 1048576, 8
    speed = np.array([0.609, 0.600, 0.553, 0.592, 0.585, 0.604, 0.570, 0.597])     # bodies per unit time, per device
    _, b, c, _ = shares(nbx, n, P, None)
    t0 = np.array(c) / speed
    skew0 = t0.max() / t0.min() - 1.0
    w = nbx.tune_weights(c, list(t0))
    _, b1, c1, _ = shares(nbx, n, P, w)
    t1 = np.array(c1) / speed
    skew1 = t1.max() / t1.min() - 1.0
    assert skew0 > 0.10 and skew1 < 2.5 * 256 / (n / P)          # 10.1 % -> 0.3 %: two tiles of 512 between the extremes
    assert t1.max() < 0.96 * t0.max()                             # the step (= the slowest device) is 4.5 % shorter
    assert sum(c1) == n and c1[2] < c[2] < c1[0]                  # the slow device owns fewer bodies, the fast one more
    # a second pass changes at most one tile per rank: the fixed point
    w2 = nbx.tune_weights(c1, list(t1))
    _, _, c2, _ = shares(nbx, n, P, w2)
    assert max(abs(x - y) for x, y in zip(c1, c2)) <= 256


def test_random_weights_never_lose_a_body(nbx):
    rng = np.random.default_rng(7)
    for _ in range(200):
        n = int(rng.integers(1, 300000))
        P = int(rng.integers(1, 17))
        w = list(rng.uniform(0.05, 5.0, P))
        used, b, c, n_alloc = shares(nbx, n, P, w)
        assert sum(c) == n and n_alloc % 256 == 0 and n_alloc >= n and all(x > 0 for x in c[:used])
