#!/usr/bin/env python3
"""Does NBX_ORDER_REFERENCE_SPLIT (j-splits whose accumulators start from the previous step's prefix sums) keep the
reference's kinetic energy as well as the unsplit reference order does, and what does it cost?

usage: python tools/primed_check.py n steps out.json [B:S ...] [--no-tree]
Traces: ref (NBX_KERNEL_EXACT, = CPU ver7 bit for bit), seq (reference order, one accumulator), tree (default tree sums),
split<S> (order 3 with S splits).  Prints max relative kinetic-energy deviation from ref and ms/step of each.
"""
import json
import sys
import time

sys.path.insert(0, "nbody-demo-2023_amd")
import nbx  # noqa: E402
import numpy as np  # noqa: E402


def main():
    n, steps, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    splits = [x for x in sys.argv[4:] if not x.startswith("--")] or ["2:2", "4:8", "4:16"]   # bodies_per_lane:j_split
    ctx = {
        "ref": nbx.Context(n, 32, kernel_variant=nbx.KERNEL_EXACT),
        "seq": nbx.Context(n, 32, summation_order=nbx.ORDER_REFERENCE),
        "tree": nbx.Context(n, 32, summation_order=nbx.ORDER_TREE),
    }
    for s in splits:
        b, k = (int(v) for v in s.split(":"))
        ctx["splitB%dS%d" % (b, k)] = nbx.Context(n, 32, summation_order=nbx.ORDER_REFERENCE_SPLIT, bodies_per_lane=b, j_split=k)
    if "--no-tree" in sys.argv:
        ctx.pop("tree").close()
    ic = nbx.initial_conditions(n, 32)
    for c in ctx.values():
        c.upload(ic)
    tr = {k: [] for k in ctx}
    ms = {k: 0.0 for k in ctx}
    chunk = 1 if "--every" in sys.argv else 10
    done = 0
    t0 = time.time()
    while done < steps:
        k = min(chunk, steps - done)
        for name, c in ctx.items():
            c.sync()
            t = time.time()
            tr[name] += list(c.step_trace(k))
            c.sync()
            ms[name] += (time.time() - t) * 1e3
        done += k
        r = tr["ref"][-1]
        print("step %4d/%d ref %.9g | " % (done, steps, r)
              + "  ".join("%s %+.2e" % (k, tr[k][-1] / r - 1) for k in tr if k != "ref") + "  (%.0f s)" % (time.time() - t0), flush=True)
    a = {k: np.array(v) for k, v in tr.items()}
    res = {"n": n, "steps": steps, "ms_per_step": {k: v / steps for k, v in ms.items()},
           "stats": {k: c.stats() for k, c in ctx.items()} if hasattr(ctx["ref"], "stats") else None,
           "max_rel_dev_vs_ref": {k: float(np.max(np.abs(a[k] - a["ref"]) / np.abs(a["ref"]))) for k in a if k != "ref"}}
    res["stale_per_step"] = {k: c.stats().get("stale_reevaluations", 0) / steps for k, c in ctx.items()}
    print(json.dumps({k: res[k] for k in ("ms_per_step", "max_rel_dev_vs_ref", "stale_per_step")}, indent=1))
    json.dump(res, open(out, "w"), indent=1, default=str)


if __name__ == "__main__":
    main()
