import sys
sys.path.insert(0, 'nbody-demo-2023_amd')
import nbx
n = 1048576
ic = nbx.initial_conditions(n)
for own in (32768, 65536, 81920, 98304, 131072, 163840, 196608, 262144, 327680, 393216, 458752, 524288, 786432, 1048576):
    row = []
    for B in (1, 2, 4):
        c = nbx.Context(n, 32, i_begin=0, i_count=own, n_alloc=n, summation_order=nbx.ORDER_REFERENCE, bodies_per_lane=B)
        c.upload(ic)
        def run(k):
            for _ in range(k):
                c.step_local(); c.commit()
        run(2); c.sync(); c.profile(True); run(max(3, int(6e11 / (float(n) * own)))); c.sync()
        st = c.stats(); c.close()
        ms = st['force_ms_total'] / st['force_launches_timed']
        row.append("B%d %7.3f ms %5.1f%%" % (B, ms, 100 * 20.0 * float(n) * own / (ms * 1e-3) / 157.3e12))
    with nbx.Context(n, 32, i_begin=0, i_count=own, n_alloc=n) as c:
        auto = c.stats()['bodies_per_lane']
    print("own=%7d  %s   auto=B%d" % (own, "   ".join(row), auto), flush=True)
