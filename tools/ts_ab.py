"""A/B of the time-sliced wave priority (inner_loop = ASM_TS) against the plain hand-scheduled loop, reference-order shapes with
1, 2, 4 and 8 waves per SIMD; same process, interleaved rounds; slice lengths 2^k x 10 ns through NBX_SLICE_BIT."""
import os
import sys
sys.path.insert(0, 'nbody-demo-2023_amd')
import nbx
CASES = [(262144, 262144, 2), (524288, 524288, 2), (1048576, 1048576, 2), (262144, 262144, 4), (524288, 524288, 4), (1048576, 131072, 2),
         (393216, 393216, 2)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if len(sys.argv) > 3:  # n:own:B,n:own:B,...
    CASES = [tuple(int(x) for x in c.split(":")) for c in sys.argv[3].split(",")]
ks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [12, 13, 14, 15, 16]
for n, own, B in CASES:
    ic = nbx.initial_conditions(n)
    ctx = {}
    variants = [("asm", nbx.LOOP_ASM, None)] + [("ts k=%d" % k, nbx.LOOP_ASM_TS, k) for k in ks]
    for name, loop, k in variants:
        if k is not None:
            os.environ["NBX_SLICE_BIT"] = str(k)
        c = nbx.Context(n, 32, i_begin=0, i_count=own, n_alloc=n, summation_order=nbx.ORDER_REFERENCE, kernel_variant=nbx.KERNEL_SGPR, inner_loop=loop,
                        bodies_per_lane=B)
        c.upload(ic)
        ctx[name] = c
    steps = max(3, int(1.0e11 / (float(n) * own)))
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
            best.setdefault(name, []).append(st['force_ms_total'] / st['force_launches_timed'])
    base = sorted(best["asm"])[len(best["asm"]) // 2]
    for name, c in ctx.items():
        st = c.stats(); c.close()
        ms = sorted(best[name])[len(best[name]) // 2]
        print("n=%8d own=%8d B%d grid %4dx%d %-8s median %8.3f ms (min %8.3f)  %5.2f %%  %+5.2f %% vs asm" % (
            n, own, st['bodies_per_lane'], st['force_grid_x'], st['force_grid_y'], name, ms, min(best[name]),
            100 * 20.0 * float(n) * own / (ms * 1e-3) / 157.3e12, 100 * (base / ms - 1)), flush=True)
