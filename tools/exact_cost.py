"""Cost of the validation kernel (NBX_KERNEL_EXACT: the reference's arithmetic bit for bit, one thread per body) against the
default kernel, per n: ms per step of each and the ratio (replace the "~50x" / "~500x" in the docs by numbers).
usage: python tools/exact_cost.py [n ...]   (GPU box, repo root)"""
import sys
import time

sys.path.insert(0, 'nbody-demo-2023_amd')
import nbx

sizes = [int(x) for x in sys.argv[1:]] or [2000, 16384, 32768, 65536, 131072, 262144, 1048576]
print("%9s %14s %14s %8s   %s" % ("n", "default ms", "exact ms", "ratio", "exact-mode grid (workgroups of 256 threads on 256 CUs)"))
for n in sizes:
    ic = nbx.initial_conditions(n)
    t = {}
    for name, kw in (("default", {}), ("exact", dict(kernel_variant=nbx.KERNEL_EXACT))):
        with nbx.Context(n, 32, use_graph=2, **kw) as c:
            c.upload(ic)
            c.step(2, kenergy=False)
            c.sync()
            per = 1e-3 if name == "default" else t["default"] * 6
            steps = max(2, min(200, int(1.0 / max(per, 1e-5))))
            t0 = time.perf_counter()
            c.step(steps, kenergy=False)
            c.sync()
            t[name] = (time.perf_counter() - t0) / steps
            if name == "exact":
                g = c.stats()["force_grid_x"]
    print("%9d %14.4f %14.4f %8.1f   %d" % (n, 1e3 * t["default"], 1e3 * t["exact"], t["exact"] / t["default"], g), flush=True)
