"""A/B of the hand-scheduled j loop (inner_loop=ASM) against the compiler-scheduled one, same process, interleaved rounds."""
import sys
sys.path.insert(0, 'nbody-demo-2023_amd')
import nbx
CASES = [(1048576, 131072, dict(bodies_per_lane=2)), (262144, 262144, dict(bodies_per_lane=4)), (262144, 262144, dict(bodies_per_lane=2)),
         (1048576, 1048576, dict(bodies_per_lane=4)), (1048576, 65536, dict(bodies_per_lane=2)), (1048576, 262144, dict(bodies_per_lane=4))]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for n, own, kw in CASES:
    ic = nbx.initial_conditions(n)
    ctx = {}
    for name, loop in (("asm", nbx.LOOP_ASM), ("cxx", nbx.LOOP_CXX)):
        c = nbx.Context(n, 32, i_begin=0, i_count=own, n_alloc=n, summation_order=nbx.ORDER_REFERENCE, kernel_variant=nbx.KERNEL_SGPR, inner_loop=loop, **kw)
        c.upload(ic)
        ctx[name] = c
    steps = max(3, int(1.5e11 / (float(n) * own)))
    best = {}
    for r in range(rounds):
        for name, c in ctx.items():
            for _ in range(2):
                c.step_local(); c.commit()
            c.sync(); c.profile(True)
            for _ in range(steps):
                c.step_local(); c.commit()
            c.sync()
            st = c.stats(); c.profile(False)
            ms = st['force_ms_total'] / st['force_launches_timed']
            best.setdefault(name, []).append(ms)
    for name, c in ctx.items():
        st = c.stats(); c.close()
        ms = sorted(best[name])[len(best[name]) // 2]
        print("n=%8d own=%8d B%d grid %4dx%d loop %s  median %8.3f ms (min %8.3f)  %5.1f %%" % (n, own, st['bodies_per_lane'], st['force_grid_x'], st['force_grid_y'], name,
              ms, min(best[name]), 100 * 20.0 * float(n) * own / (ms * 1e-3) / 157.3e12), flush=True)
