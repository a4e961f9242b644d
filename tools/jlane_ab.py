"""One-launch-per-step kernel (NBX_KERNEL_JLANE) against the two-launch tree shape (SGPRW): us per step, graph replay on."""
import sys, time
sys.path.insert(0, 'nbody-demo-2023_amd')
import numpy as np
import nbx
sizes = [int(x) for x in sys.argv[1:]] or [2000, 2048, 4096, 8192, 16384, 32768, 65536]
for n in sizes:
    ic = nbx.initial_conditions(n)
    row = []
    ref = None
    for name, kw in (("sgprw", dict(kernel_variant=nbx.KERNEL_SGPRW)), ("jlane2", dict(kernel_variant=nbx.KERNEL_JLANE, bodies_per_lane=2)),
                     ("jlane4", dict(kernel_variant=nbx.KERNEL_JLANE, bodies_per_lane=4)), ("jlane8", dict(kernel_variant=nbx.KERNEL_JLANE, bodies_per_lane=8)),
                     ("jlane16", dict(kernel_variant=nbx.KERNEL_JLANE, bodies_per_lane=16)), ("auto", dict()),
                     ("auto/cxx", dict(inner_loop=nbx.LOOP_CXX))):
        if name.startswith("jlane") and n / int(name[5:]) > 16384:
            continue
        with nbx.Context(n, 32, **kw) as c:
            c.upload(ic)
            ke = c.step(20)
            if ref is None:
                ref = ke
            steps = max(20, min(2000, int(0.3 / (n * n / 4e12 + 5e-6))))
            c.step(steps, kenergy=False); c.sync()
            t0 = time.perf_counter(); c.step(steps, kenergy=False); c.sync(); t1 = time.perf_counter()
            st = c.stats()
        us = (t1 - t0) / steps * 1e6
        row.append("%s %7.1f us (%4.1f%%, dKE %.1e, B%d %dx%d)" % (name, us, 100 * 20.0 * n * n / (us * 1e-6) / 157.3e12, abs(ke / ref - 1), st['bodies_per_lane'], st['force_grid_x'], st['force_grid_y']))
    print("n=%6d  " % n + "  ".join(row), flush=True)
