"""Worker for the GPU multi-rank rehearsals: the PRODUCT engine (libnbx on cuda:0) under torch.distributed.
   argv: n steps out backend   -- backend gloo: several ranks share the one GPU, exchange staged through the host;
                                  backend nccl: a 1-rank RCCL group with the collectives forced on."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-demo-2023_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import nbx  # noqa: E402
import sharded  # noqa: E402


def main():
    n, steps, out, backend = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    weights = [float(x) for x in os.environ["NBX_TEST_WEIGHTS"].split(",")] if os.environ.get("NBX_TEST_WEIGHTS") else None
    shape = dict(summation_order=nbx.ORDER_REFERENCE) if weights else dict(j_split=4, bodies_per_lane=2)  # unequal shares: reference order is share-independent bit for bit
    sim = sharded.ShardedSimulation(n, 32, dist=dist, force_collective=True, weights=weights, **shape)
    sim.upload(nbx.initial_conditions(n))
    ke = []
    for _ in range(steps):
        sim.step(1)
        ke.append(sim.kenergy())
    d = sim.engine.download()
    res = {"rank": rank, "world": world, "ke": ke, "n_alloc": sim.n_alloc, "i_begin": sim.i_begin, "i_count": sim.i_count,
           "pos_x": d["pos_x"].tolist()[:64] + d["pos_x"].tolist()[-64:], "bytes_gathered": sim.bytes_gathered}
    with open("%s.%d" % (out, rank), "w") as f:
        json.dump(res, f)
    sim.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
