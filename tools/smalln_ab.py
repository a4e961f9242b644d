import sys, time
sys.path.insert(0, 'nbody-demo-2023_amd')
import nbx
for n in (2000, 4096, 8192, 16384, 32768):
    ic = nbx.initial_conditions(n)
    row = []
    for g in (2, 1):
        with nbx.Context(n, use_graph=g) as c:
            c.upload(ic); c.step(100)
            t0 = time.perf_counter(); c.step(1000); t1 = time.perf_counter()
            st = c.stats()
        row.append((t1 - t0) / 1000 * 1e6)
    print("n=%6d  plain %.1f us/step  graph %.1f us/step  B%d S%d grid %dx%d" % (n, row[0], row[1], st['bodies_per_lane'], st['j_split'], st['force_grid_x'], st['force_grid_y']))
