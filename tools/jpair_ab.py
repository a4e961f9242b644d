"""Reference-order slices that leave less than one wave per SIMD: one body per lane with the compiled loop (plain VALU
ops), one body per lane with the two-j-records-per-operation loop (sgpr_loop_asm_jpair) and two bodies per lane with the
hand-scheduled loop, same process, interleaved rounds; bit-equality of the owned velocities after 2 steps.
usage: jpair_ab.py [n:own ...]"""
import sys
sys.path.insert(0, 'nbody-demo-2023_amd')
import nbx
import numpy as np

cases = [(262144, 65536), (262144, 32768), (1048576, 65536), (262144, 131072), (1048576, 131072), (262144, 49152), (262144, 98304)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(':')) for a in sys.argv[1:]]
SHAPES = (("B1 compiled", dict(bodies_per_lane=1, inner_loop=nbx.LOOP_CXX)),
          ("B1 jpair asm", dict(bodies_per_lane=1, inner_loop=nbx.LOOP_ASM)),
          ("B2 asm", dict(bodies_per_lane=2, inner_loop=nbx.LOOP_ASM)),
          ("auto", dict()))
for n, own in cases:
    ic = nbx.initial_conditions(n)
    ctxs, ref = [], None
    for name, kw in SHAPES:
        c = nbx.Context(n, 32, i_begin=0, i_count=own, n_alloc=n, summation_order=nbx.ORDER_REFERENCE, kernel_variant=nbx.KERNEL_SGPR, **kw)
        c.upload(ic)
        for _ in range(2):
            c.step_local(); c.commit()
        c.sync()
        d = c.download()
        sig = tuple(np.ascontiguousarray(d[f][:own]).tobytes() for f in ("vel_x", "vel_y", "vel_z", "pos_x"))
        same = "ref" if ref is None else ("bit-equal" if sig == ref else "DIFFERENT")
        ref = ref or sig
        ctxs.append((name, c, same, []))
    steps = max(2, int(6e10 / (float(n) * own)))
    for rnd in range(3):
        for name, c, same, ms in ctxs:
            c.profile(True)
            import time
            t0 = time.perf_counter()
            for _ in range(steps):
                c.step_local(); c.commit()
            c.sync()
            wall = (time.perf_counter() - t0) * 1e3 / steps
            st = c.stats()
            c.profile(False)
            ms.append((st['force_ms_total'] / st['force_launches_timed'], wall))
    for name, c, same, ms in ctxs:
        st = c.stats()
        k = min(m[0] for m in ms)
        w = min(m[1] for m in ms)
        print("n=%8d own=%8d %-13s B%d loop%d grid %4dx%d  kernel %8.3f ms  step %8.3f ms  %5.1f %%  %s" % (
            n, own, name, st['bodies_per_lane'], st['inner_loop'], st['force_grid_x'], st['force_grid_y'], k, w,
            100 * 20.0 * float(n) * own / (k * 1e-3) / 157.3e12, same), flush=True)
        c.close()
