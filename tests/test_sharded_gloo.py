"""Multi-rank host logic on CPU: world_size 2 and 3 over gloo (the N>1 path of bench.py and of
sharded.ShardedSimulation), plus the partition arithmetic."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, rel_err


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_block_partition_properties():
    import sharded
    for n in (1, 5, 255, 256, 257, 2000, 4099, 16384, 262144, 1048576, 1000003):
        for world in (1, 2, 3, 4, 8):
            parts = [sharded.block_partition(n, world, r) for r in range(world)]
            block, n_alloc = parts[0][0], parts[0][3]
            assert block % 256 == 0 and n_alloc == world * block and n_alloc >= n
            assert all(p[0] == block and p[3] == n_alloc for p in parts)
            covered = 0
            for r, (_, ib, ic, _) in enumerate(parts):
                assert ib == min(r * block, n) and 0 <= ic <= block
                covered += ic
            assert covered == n  # every body owned exactly once (the reference's slices are not: cpu/Compute.cpp:50-51)
    with pytest.raises(ValueError):
        sharded.block_partition(0, 2, 0)
    with pytest.raises(ValueError):
        sharded.block_partition(10, 2, 2)


@pytest.mark.parametrize("n,world,staged", [(1000, 2, False), (1500, 3, False), (1000, 2, True)])
def test_gloo_ranks_reproduce_single_process_trace(oracle, tmp_path, n, world, staged):
    steps = 15
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), OMP_NUM_THREADS="2")
    if staged:  # sharded.py's out-of-place fallback (second buffer + copy) must give the same positions
        env["NBX_TEST_STAGED_GATHER"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"],
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(n), str(steps), out]
    subprocess.run(cmd, env=env, check=True, timeout=300, capture_output=True)
    res = [json.load(open("%s.%d" % (out, r))) for r in range(world)]
    pos = [np.load("%s.%d.npy" % (out, r)) for r in range(world)]

    s = oracle.init_state(n)
    ke_ref = oracle.run(s, steps)
    ref_pos = np.stack([s.pos_x, s.pos_y, s.pos_z], axis=1)
    for r in range(world):
        assert res[r]["world"] == world
        # every rank ends with ALL positions, bit-identical to the single-process run
        assert np.array_equal(pos[r], ref_pos), r
        # energy: per-rank float partial sums added in double instead of one float reduction
        assert rel_err(res[r]["ke"], ke_ref).max() < 2e-6
        assert res[r]["ke"] == res[0]["ke"]
        assert res[r]["bytes_gathered"] == steps * (world - 1) * res[r]["block"] * 16
    assert sum(x["i_count"] for x in res) == n
    # the gather-to-rank-0 breakdown of bench.py's N > 1 line: every rank reported, in rank order
    summ = json.load(open(out + ".summary"))
    assert summ["n_reports"] == world
    b = summ["summary"]
    assert b["world_seen"] == world and [r["rank"] for r in b["per_rank"]] == list(range(world))
    assert [r["bodies_owned"] for r in b["per_rank"]] == [x["i_count"] for x in res]
    assert b["force_kernel_ms"] == {"min": 1.0, "mean": (world + 1) / 2.0, "max": float(world)} and b["skew_ms"] == world - 1.0
    assert b["slowest_rank"] == world - 1
    assert b["bytes_gathered_per_step"] == (world - 1) * res[0]["block"] * 16 and b["bytes_sent_per_step"] == res[0]["block"] * 16
    ag = b["allgather_ms_per_step"]
    assert 0.0 < ag["min"] <= ag["mean"] <= ag["max"] <= b["allgather_ms_worst_step"]
    assert all(r["allgather_ms"] > 0.0 for r in b["per_rank"])
    # ... and its parity leg: restart from seed 42, 10 steps, energies all-reduced, against the reference's trace
    par = json.load(open(out + ".parity"))
    assert par["ke_restart"] == res[0]["ke"][:10]  # the restart reproduces the first run, bit for bit
    if n == 1000:  # a fixture from the reference's own binary exists for this size
        p = par["parity"]
        assert p["fixture"] == "ver7_f32_n1000_s100.json" and p["steps"] == 10 and p["ranks"] == world
        assert p["pass"] and p["max_rel_kenergy_err"] < 2e-6 and len(p["rel_kenergy_err_per_step"]) == 10
    else:
        assert par["parity"] is None


@pytest.mark.parametrize("n,world,weights,counts", [(1000, 2, "1,3", [256, 744]), (1500, 3, "2,1,1", [768, 512, 220])])
def test_gloo_ranks_with_unequal_shares_reproduce_the_single_process_run(oracle, tmp_path, n, world, weights, counts):
    """ShardedSimulation(weights=...): shares in whole 256-record tiles from libnbx's nbx_partition_weighted, the per-step exchange as one
    in-place broadcast per owner -- every rank still ends every step with ALL positions, bit-identical to the single-process run."""
    steps = 12
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), OMP_NUM_THREADS="2", NBX_TEST_WEIGHTS=weights)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"],
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(n), str(steps), out]
    subprocess.run(cmd, env=env, check=True, timeout=300, capture_output=True)
    res = [json.load(open("%s.%d" % (out, r))) for r in range(world)]
    pos = [np.load("%s.%d.npy" % (out, r)) for r in range(world)]
    s = oracle.init_state(n)
    ke_ref = oracle.run(s, steps)
    ref_pos = np.stack([s.pos_x, s.pos_y, s.pos_z], axis=1)
    assert [x["i_count"] for x in res] == counts and [x["i_begin"] for x in res] == [sum(counts[:r]) for r in range(world)]
    for r in range(world):
        assert np.array_equal(pos[r], ref_pos), r
        assert rel_err(res[r]["ke"], ke_ref).max() < 2e-6 and res[r]["ke"] == res[0]["ke"]
        assert res[r]["bytes_gathered"] == steps * (n - counts[r]) * 16     # what the other owners broadcast
        assert res[r]["n_alloc"] == -(-n // 256) * 256


def test_rank_report_summary_arithmetic():
    """summarise_rank_reports is plain arithmetic: check it without any process group (and that a missing rank is an error)."""
    import sharded
    reps = [{"rank": 1, "device": 1, "host": "h", "bodies_owned": 100, "force_ms_mean": 32.0, "allgather_ms": [0.5, 0.7], "elapsed_s": 1.0},
            {"rank": 0, "device": 0, "host": "h", "bodies_owned": 128, "force_ms_mean": 30.0, "allgather_ms": [2.5, 2.9], "elapsed_s": 1.1}]
    b = sharded.summarise_rank_reports(reps, 4096)
    assert b["force_kernel_ms"] == {"min": 30.0, "mean": 31.0, "max": 32.0} and b["skew_ms"] == 2.0 and b["slowest_rank"] == 1
    assert abs(b["allgather_ms_per_step"]["min"] - 0.6) < 1e-12 and abs(b["allgather_ms_per_step"]["max"] - 2.7) < 1e-12
    assert b["allgather_ms_worst_step"] == 2.9 and b["bytes_gathered_per_step"] == 4096
    assert [r["device"] for r in b["per_rank"]] == [0, 1]
    with pytest.raises(ValueError):
        sharded.summarise_rank_reports([reps[0], dict(reps[0])], 4096)


def test_world_too_large_for_n_is_refused_on_every_rank(tmp_path):
    """n = 300 on 3 ranks gives blocks of 256 -> rank 2 would own nothing.  All ranks must raise the same error
    before any engine or collective exists; the run ends promptly instead of hanging in an all-gather."""
    out = str(tmp_path / "res")
    port = str(_free_port())
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "tests", "_dist_worker.py"), "300", "2", out]
    subprocess.run(cmd, env=env, check=True, timeout=120, capture_output=True)
    res = [json.load(open("%s.%d" % (out, r))) for r in range(3)]
    assert all("refused" in r for r in res) and len({r["refused"] for r in res}) == 1
    assert "at most 2 ranks" in res[0]["refused"]


def test_check_world_matches_the_native_partition(nbx):
    """sharded.check_world / block_partition against libnbx's nbx_partition (the arithmetic nbx_group_create and
    nbx_group_create_rank use): same block, same slices, and the same verdict on which world sizes leave a rank empty."""
    import sharded
    for n in (1, 5, 255, 256, 257, 300, 1500, 2000, 4099, 16384, 262144, 1048576, 1000003):
        for world in (1, 2, 3, 4, 5, 7, 8, 16, 64):
            used = nbx.partition(n, world, 0)[0]
            try:
                sharded.check_world(n, world)
                assert used == world, (n, world, used)
            except ValueError:
                assert used < world, (n, world, used)
                continue
            for r in range(world):
                u, block, ib, ic, n_alloc = nbx.partition(n, world, r)
                assert (block, ib, ic, n_alloc) == sharded.block_partition(n, world, r), (n, world, r)
            # when ranks are dropped the native partition re-balances over the ranks that remain
        for world in (3, 8):
            used, block, _, _, n_alloc = nbx.partition(n, world, 0)
            assert n_alloc == used * block and (used - 1) * block < n <= used * block
            assert sum(nbx.partition(n, world, r)[3] for r in range(world)) == n
