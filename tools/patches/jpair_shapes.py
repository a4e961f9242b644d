"""Reference-order kernel: two i-bodies per packed op (j_pairing=2, the shipped default) vs two j-records per packed op
(j_pairing=1), by owned bodies and bodies per lane; and a bit-equality check of the two after 2 steps."""
import sys
sys.path.insert(0, 'nbody-demo-2023_amd')
import nbx
import numpy as np
cases = [(262144, 262144), (1048576, 131072), (524288, 524288), (1048576, 1048576), (262144, 65536)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(':')) for a in sys.argv[1:]]
for n, own in cases:
    ic = nbx.initial_conditions(n)
    ref = None
    for name, kw in (("ibody B2", dict(j_pairing=2, bodies_per_lane=2)), ("ibody B4", dict(j_pairing=2, bodies_per_lane=4)),
                     ("jpair B1", dict(j_pairing=1, bodies_per_lane=1)), ("jpair B2", dict(j_pairing=1, bodies_per_lane=2))):
        c = nbx.Context(n, 32, i_begin=0, i_count=own, n_alloc=n, summation_order=nbx.ORDER_REFERENCE, kernel_variant=nbx.KERNEL_SGPR, **kw)
        c.upload(ic)
        def run(k):
            for _ in range(k):
                c.step_local(); c.commit()
        run(2); c.sync()
        d = c.download()
        sig = tuple(np.ascontiguousarray(d[f][:own]).tobytes() for f in ("vel_x", "vel_y", "vel_z"))
        same = "ref" if ref is None else ("bit-equal" if sig == ref else "DIFFERENT")
        ref = ref or sig
        c.profile(True)
        steps = max(3, int(2e11 / (float(n) * own)))
        run(steps); c.sync()
        st = c.stats(); c.close()
        ms = st['force_ms_total'] / st['force_launches_timed']
        print("n=%8d own=%8d %s grid %4dx%d  %8.3f ms  %5.1f %%  %s" % (n, own, name, st['force_grid_x'], st['force_grid_y'], ms,
              100 * 20.0 * float(n) * own / (ms * 1e-3) / 157.3e12, same), flush=True)
