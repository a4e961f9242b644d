import sys
sys.path.insert(0, 'nbody-demo-2023_amd')
import nbx
for n, own, kw in [(1048576, 131072, {}), (262144, 262144, {}), (262144, 262144, dict(bodies_per_lane=2)), (1048576, 1048576, {}),
                   (262144, 262144, dict(summation_order=nbx.ORDER_TREE)), (16384, 16384, {}), (65536, 65536, {})]:
    ic = nbx.initial_conditions(n)
    c = nbx.Context(n, 32, i_begin=0, i_count=own, n_alloc=n, **kw)
    c.upload(ic)
    steps = max(3, int(2e11 / (float(n) * own)))
    def run(k):
        for _ in range(k):
            c.step_local(); c.commit()
    run(2); c.sync(); c.profile(True)
    run(steps); c.sync()
    st = c.stats(); c.close()
    ms = st['force_ms_total'] / st['force_launches_timed']
    print("n=%8d own=%8d %s order %d kernel %d B%d grid %4dx%d  %8.3f ms  %5.1f %%" % (n, own, kw, st['summation_order'], st['kernel_variant'],
          st['bodies_per_lane'], st['force_grid_x'], st['force_grid_y'], ms, 100 * 20.0 * float(n) * own / (ms * 1e-3) / 157.3e12), flush=True)
