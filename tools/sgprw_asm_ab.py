"""SGPRW (tree order) with the hand-scheduled loop vs the compiler-scheduled one: us per step, bit-equality of the energies."""
import sys, time
sys.path.insert(0, 'nbody-demo-2023_amd')
import nbx
for n in (16384, 32768, 65536, 131072):
    ic = nbx.initial_conditions(n)
    res = {}
    for name, loop in (("asm", nbx.LOOP_ASM), ("cxx", nbx.LOOP_CXX)):
        with nbx.Context(n, 32, kernel_variant=nbx.KERNEL_SGPRW, inner_loop=loop) as c:
            c.upload(ic)
            ke = c.step(10)
            steps = max(20, min(2000, int(0.5 / (n * n / 4e12 + 5e-6))))
            best = 1e9
            for _ in range(3):
                c.sync(); t0 = time.perf_counter(); c.step(steps, kenergy=False); c.sync(); best = min(best, (time.perf_counter() - t0) / steps)
            st = c.stats()
            res[name] = (best * 1e6, ke, st)
    a, b = res["asm"], res["cxx"]
    print("n=%7d  asm %8.1f us (%4.1f%%)  cxx %8.1f us (%4.1f%%)  bit-equal energy: %s  shape B%d S%d %dx%d inner %d/%d" % (
        n, a[0], 100 * 20.0 * n * n / (a[0] * 1e-6) / 157.3e12, b[0], 100 * 20.0 * n * n / (b[0] * 1e-6) / 157.3e12, a[1] == b[1],
        a[2]['bodies_per_lane'], a[2]['j_split'], a[2]['force_grid_x'], a[2]['force_grid_y'], a[2]['inner_loop'], b[2]['inner_loop']), flush=True)
