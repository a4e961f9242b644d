"""Throughput of NBX_ORDER_REFERENCE_SPLIT shapes by owned-body count, bodies/lane and split count (force kernel ms from
HIP events, whole step ms from the host clock), next to the unsplit reference order and the tree default."""
import sys, time
sys.path.insert(0, 'nbody-demo-2023_amd')
import nbx
cases = [(1048576, 131072), (262144, 262144), (524288, 524288), (1048576, 1048576)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(':')) for a in sys.argv[1:]]
for n, own in cases:
    ic = nbx.initial_conditions(n)
    shapes = [('seq', dict(summation_order=nbx.ORDER_REFERENCE)), ('tree', dict(summation_order=nbx.ORDER_TREE))]
    for B in (2, 4):
        for S in (2, 4, 8, 16):
            shapes.append(('split B%d S%d' % (B, S), dict(summation_order=nbx.ORDER_REFERENCE_SPLIT, bodies_per_lane=B, j_split=S)))
    for name, kw in shapes:
        c = nbx.Context(n, 32, i_begin=0, i_count=own, n_alloc=n, **kw)
        c.upload(ic)
        steps = max(3, int(3e11 / (float(n) * own)))
        def run(k):
            for _ in range(k):
                c.step_local(); c.commit()
        run(2); c.sync(); c.profile(True)
        t = time.time(); run(steps); c.sync(); wall = (time.time() - t) * 1e3 / steps
        st = c.stats(); c.close()
        ms = st['force_ms_total'] / st['force_launches_timed']
        print("n=%8d own=%8d %-14s grid %4dx%-2d force %8.3f ms %5.1f %%   step %8.3f ms %5.1f %%" % (
            n, own, name, st['force_grid_x'], st['force_grid_y'], ms, 100 * 20.0 * float(n) * own / (ms * 1e-3) / 157.3e12,
            wall, 100 * 20.0 * float(n) * own / (wall * 1e-3) / 157.3e12), flush=True)
