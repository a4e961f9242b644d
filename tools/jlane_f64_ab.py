"""fp64: one-launch jlane kernel vs the two-launch SGPRW shape, us per step (graph replay on), energy agreement."""
import sys, time
sys.path.insert(0, 'nbody-demo-2023_amd')
import nbx
for n in ([int(x) for x in sys.argv[1:]] or [2000, 2048, 4096, 8192, 12288, 16384]):
    ic = nbx.initial_conditions(n, 64)
    row, ref = [], None
    for name, kw in (("sgprw", dict(kernel_variant=nbx.KERNEL_SGPRW)), ("jlane2", dict(kernel_variant=nbx.KERNEL_JLANE, bodies_per_lane=2)),
                     ("jlane4", dict(kernel_variant=nbx.KERNEL_JLANE, bodies_per_lane=4)), ("jlane8", dict(kernel_variant=nbx.KERNEL_JLANE, bodies_per_lane=8)), ("auto", dict())):
        with nbx.Context(n, 64, **kw) as c:
            c.upload(ic)
            ke = c.step(20)
            ref = ke if ref is None else ref
            steps = max(20, min(2000, int(0.3 / (n * n / 2e12 + 5e-6))))
            c.step(steps, kenergy=False); c.sync()
            t0 = time.perf_counter(); c.step(steps, kenergy=False); c.sync(); t1 = time.perf_counter()
            st = c.stats()
        us = (t1 - t0) / steps * 1e6
        row.append("%s %7.1f us (%4.1f%%, dKE %.1e, %dx%d)" % (name, us, 100 * 20.0 * n * n / (us * 1e-6) / 78.6e12, abs(ke / ref - 1), st['force_grid_x'], st['force_grid_y']))
    print("n=%6d  " % n + "  ".join(row), flush=True)
