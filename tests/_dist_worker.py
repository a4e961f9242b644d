"""Worker for test_sharded_gloo.py: one rank of a world_size-N gloo job on the CPU.

Exercises the product's multi-rank HOST logic (nbody-demo-2023_amd/sharded.py: block partition,
in-place all-gather of the position blocks, scalar all-reduce of the energy partials) with the
compute engine replaced by the oracle -- test infrastructure standing in for the GPU, so the
collective pattern can be rehearsed where no GPU exists.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-demo-2023_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import oracle as O  # noqa: E402
import sharded  # noqa: E402

G = np.float32(6.67259e-11)


class OracleEngine:
    """Same interface as sharded.NbxEngine, arithmetic by oracle/nbody_oracle.c."""

    def __init__(self, n, precision, i_begin, i_count, n_alloc, **opts):
        assert precision == 32
        self.n, self.i0, self.i1, self.n_alloc = n, i_begin, i_begin + i_count, n_alloc
        self.rec = [np.zeros((n_alloc, 4), dtype=np.float32) for _ in range(2)]
        self.cur = 0
        self.s = None
        self.energy = 0.0

    def upload(self, state):
        s = O.State(self.n)
        for f in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "mass"):
            getattr(s, f)[:] = state[f]
        self.s = s
        for r in self.rec:
            r[: self.n, 0], r[: self.n, 1], r[: self.n, 2], r[: self.n, 3] = s.pos_x, s.pos_y, s.pos_z, G * s.mass

    def step_local(self, dt):
        s = self.s
        cur = self.rec[self.cur]
        s.pos_x[:], s.pos_y[:], s.pos_z[:] = cur[: self.n, 0], cur[: self.n, 1], cur[: self.n, 2]
        O.accel(s, self.i0, self.i1)
        self.energy = float(O.integrate(s, dt, self.i0, self.i1))
        nxt = self.rec[self.cur ^ 1]
        sl = slice(self.i0, self.i1)
        nxt[sl, 0], nxt[sl, 1], nxt[sl, 2] = s.pos_x[sl], s.pos_y[sl], s.pos_z[sl]

    def exchange_tensor(self):
        return torch.from_numpy(self.rec[self.cur ^ 1].reshape(-1).view(np.uint8))

    def commit(self):
        self.cur ^= 1

    def kenergy_partial(self):
        return self.energy

    def positions(self):
        r = self.rec[self.cur]
        return r[: self.n, :3].copy()

    def sync(self):
        pass

    def close(self):
        pass


def main():
    n, steps, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    weights = [float(x) for x in os.environ["NBX_TEST_WEIGHTS"].split(",")] if os.environ.get("NBX_TEST_WEIGHTS") else None
    try:
        sim = sharded.ShardedSimulation(n, 32, dist=dist, engine_factory=OracleEngine, weights=weights)
    except ValueError as e:  # a world too large for n: every rank must get here (none may be left in a collective)
        with open("%s.%d" % (out, rank), "w") as f:
            json.dump({"rank": rank, "world": world, "refused": str(e)}, f)
        dist.barrier()
        dist.destroy_process_group()
        return
    if os.environ.get("NBX_TEST_STAGED_GATHER"):  # the out-of-place fallback of the all-gather
        sim._staged = True
    ic = O.init_state(n)
    sim.upload({f: getattr(ic, f) for f in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "mass")})
    ke = []
    sim.profile_exchange(True)  # bench.py's N > 1 breakdown: every all-gather timed on every rank ...
    for _ in range(steps):
        sim.step(1, dt=float(O.DT_F32))
        ke.append(sim.kenergy())
    pos = sim.engine.positions()
    xms = sim.exchange_ms()
    # ... and gathered to rank 0 (all ranks receive the list; rank 0 writes the summary the bench line carries)
    reports = sharded.gather_rank_reports(dist, {"rank": rank, "device": None, "host": "cpu", "bodies_owned": sim.i_count,
                                                 "force_ms_mean": 1.0 + rank, "allgather_ms": xms, "elapsed_s": 0.5 + rank})
    if rank == 0:
        with open(out + ".summary", "w") as f:
            json.dump({"summary": sharded.summarise_rank_reports(reports, sim.block * sim.rec), "n_reports": len(reports)}, f)
    res = {"rank": rank, "world": world, "ke": ke, "i_begin": sim.i_begin, "i_count": sim.i_count,
           "block": sim.block, "n_alloc": sim.n_alloc, "bytes_gathered": sim.bytes_gathered,
           "pos_crc": int(np.frombuffer(pos.tobytes(), dtype=np.uint32).sum() & 0xFFFFFFFF)}
    # bench.py's N > 1 parity leg (collective: every rank calls it): the same sharded simulation restarted from the seed-42
    # state, 10 steps with the energy all-reduced after each, against the reference's own trace where a fixture exists
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    parity, ke_restart = bench.parity_probe_sharded(sim, {f: getattr(ic, f) for f in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "mass")},
                                                    n, 32, rank)
    if rank == 0:
        with open(out + ".parity", "w") as f:
            json.dump({"parity": parity, "ke_restart": ke_restart}, f)
    with open("%s.%d" % (out, rank), "w") as f:
        json.dump(res, f)
    np.save("%s.%d.npy" % (out, rank), pos)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
