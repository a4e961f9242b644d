import sys
sys.path.insert(0, 'nbody-demo-2023_amd')
import nbx
for n, own in [(262144, 131072), (262144, 65536), (262144, 32768), (1048576, 65536)]:
    ic = nbx.initial_conditions(n)
    for var, vn in ((2, 'sgpr'), (1, 'lds')):
        for B in (1, 2):
            c = nbx.Context(n, 32, i_begin=0, i_count=own, n_alloc=n, j_split=1, kernel_variant=var, bodies_per_lane=B)
            c.upload(ic)
            steps = max(2, int(1e11 / (float(n) * own)))
            def run(k):
                for _ in range(k):
                    c.step_local(); c.commit()
            run(2); c.sync(); c.profile(True)
            run(steps); c.sync()
            st = c.stats(); c.close()
            ms = st['force_ms_total'] / st['force_launches_timed']
            print("n=%8d own=%8d %s B%d grid %4dx%d  %8.3f ms  %5.1f %%" % (n, own, vn, B, st['force_grid_x'], st['force_grid_y'], ms,
                  100 * 20.0 * float(n) * own / (ms * 1e-3) / 157.3e12), flush=True)
