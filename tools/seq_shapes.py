"""Throughput of reference-order shapes (j_split = 1) by owned-body count, bodies/lane and j source."""
import sys
sys.path.insert(0, 'nbody-demo-2023_amd')
import nbx
cases = [(131072, 131072), (262144, 262144), (524288, 524288), (1048576, 1048576), (1048576, 131072), (1048576, 262144)]
for n, own in cases:
    ic = nbx.initial_conditions(n)
    for var, vn in ((2, 'sgpr'), (1, 'lds')):
        for B in (1, 2, 4, 8):
            try:
                c = nbx.Context(n, 32, i_begin=0, i_count=own, n_alloc=n, j_split=1, kernel_variant=var, bodies_per_lane=B)
            except nbx.NbxError as e:
                continue
            c.upload(ic)
            steps = max(2, int(2e11 / (float(n) * own)))
            def run(k):
                for _ in range(k):
                    c.step_local(); c.commit()
            run(2); c.sync(); c.profile(True)
            run(steps); c.sync()
            st = c.stats(); c.close()
            ms = st['force_ms_total'] / st['force_launches_timed']
            print("n=%8d own=%8d %s B%d grid %4dx%d fused=%d  %8.3f ms  %6.1f Gpair/s  %5.1f %%" % (
                n, own, vn, B, st['force_grid_x'], st['force_grid_y'], st['fused_epilogue'], ms, float(n) * own / ms * 1e-6,
                100 * 20.0 * float(n) * own / (ms * 1e-3) / 157.3e12), flush=True)
