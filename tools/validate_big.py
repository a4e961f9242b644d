#!/usr/bin/env python3
"""Fast path vs the reference's arithmetic at sizes the CPU reference cannot reach in reasonable time, and how far two
legitimate builds of the REFERENCE drift apart there.

NBX_KERNEL_EXACT reproduces the reference binary bit for bit (tests: CRC-32 of whole state arrays on every fixture up to
n = 262144), so its kinetic-energy trace stands in for `./nbody.x n steps` of CPU ver7 (3.6 h at n = 262144 x 200, ~19 h
at n = 1048576 x 100 on 8 cores).  Four traces are stepped side by side:
  fast      default kernel, fp32 (the product path)
  ref       NBX_KERNEL_EXACT        = the pinned reference build (g++ -O2, no FMA)
  ref_fma   NBX_KERNEL_EXACT_FMA    = the same loop with FMA contraction (an -march=native / icpc -xAVX2 build)
  fp64      default kernel, fp64 arithmetic on the same particles (the closest thing to the true trajectory)
  seq       fast kernel in the reference's SUMMATION ORDER (j_split = 1: one fp32 accumulator per body, j ascending)
usage: python tools/validate_big.py n steps out.json [chunk]
"""
import json
import sys
import time

sys.path.insert(0, "nbody-demo-2023_amd")
import nbx  # noqa: E402
import numpy as np  # noqa: E402


def main():
    n, steps, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    ctx = {
        "fast": nbx.Context(n, 32),
        "ref": nbx.Context(n, 32, kernel_variant=nbx.KERNEL_EXACT),
        "ref_fma": nbx.Context(n, 32, kernel_variant=nbx.KERNEL_EXACT_FMA),
        "fp64": nbx.Context(n, 64),
        # the reference's SUMMATION ORDER at speed: one fp32 accumulator per body, j strictly ascending (no splits)
        "seq": nbx.Context(n, 32, j_split=1, kernel_variant=nbx.KERNEL_SGPR, bodies_per_lane=2),
    }
    if "--no-fma" in sys.argv:
        ctx.pop("ref_fma").close()
    for k, c in ctx.items():
        c.upload(nbx.initial_conditions(n, 64 if k == "fp64" else 32))
    tr = {k: [] for k in ctx}
    t0 = time.time()
    done = 0
    while done < steps:
        k = min(chunk, steps - done)
        for name, c in ctx.items():
            tr[name] += list(c.step_trace(k))
        done += k
        r = tr["ref"][-1]
        print("step %4d/%d  ref %.9g | " % (done, steps, r) + "  ".join("%s %+.2e" % (k, tr[k][-1] / r - 1) for k in tr if k != "ref")
              + "   (%.0f s)" % (time.time() - t0), flush=True)
    a = {k: np.array(v) for k, v in tr.items()}
    rel = lambda x, y: np.abs(a[x] - a[y]) / np.abs(a[y])
    sel = list(range(50, steps + 1, 50)) or [steps]
    res = {"n": n, "steps": steps, "seconds": time.time() - t0, "printed_steps": sel, "kenergy": {k: v.tolist() for k, v in a.items()},
           "printed_5digits": {k: ["%.5g" % np.float32(v[s - 1]) for s in sel] for k, v in a.items()}}
    for x, y in (("fast", "ref"), ("ref_fma", "ref"), ("fast", "fp64"), ("ref", "fp64"), ("ref_fma", "fp64"),
                 ("seq", "ref"), ("seq", "fp64")):
        if x not in a or y not in a:
            continue
        e = rel(x, y)
        res["%s_vs_%s" % (x, y)] = {"max": float(e.max()), "printed": {str(s): float(e[s - 1]) for s in sel},
                                    "first_step_over_1e-4": int(np.argmax(e > 1e-4)) + 1 if (e > 1e-4).any() else None}
    json.dump(res, open(out, "w"))
    for k in ("fast_vs_ref", "ref_fma_vs_ref", "fast_vs_fp64", "ref_vs_fp64", "ref_fma_vs_fp64", "seq_vs_ref", "seq_vs_fp64"):
        if k not in res:
            continue
        print(k, "max %.2e" % res[k]["max"], "first>1e-4:", res[k]["first_step_over_1e-4"], {s: "%.1e" % v for s, v in res[k]["printed"].items()})


if __name__ == "__main__":
    main()
