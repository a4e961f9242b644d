"""the 16384-65536 band.  us per step (whole step: force + integrate, graph replay where the library uses
it) for j-split S x bodies-per-lane B x kernel, hand-scheduled loop wherever it exists, against the default shape.
usage: python tools/band_sweep.py [n ...]     (on the GPU box, from the repo root)"""
import sys
import time

sys.path.insert(0, 'nbody-demo-2023_amd')
import nbx

PREC = 32
if len(sys.argv) > 1 and sys.argv[1] == "--fp64":
    PREC = 64
    del sys.argv[1]
sizes = [int(x) for x in sys.argv[1:]] or [16384, 24576, 32768, 49152, 65536]
PEAK = 157.3e12 if PREC == 32 else 78.6e12


def measure(n, ic, **kw):
    try:
        c = nbx.Context(n, PREC, **kw)
    except nbx.NbxError as e:
        return None, str(e)
    with c:
        c.upload(ic)
        c.step(20)
        steps = max(20, min(2000, int(0.4 / (n * n / (4e12 if PREC == 32 else 1.9e12) + 5e-6))))
        best = 1e30
        c.step(steps, kenergy=False)
        c.sync()
        for _ in range(3):
            t0 = time.perf_counter()
            c.step(steps, kenergy=False)
            c.sync()
            best = min(best, (time.perf_counter() - t0) / steps)
        st = c.stats()
    return best, st


for n in sizes:
    ic = nbx.initial_conditions(n, PREC)
    t_def, st = measure(n, ic)
    print("n=%6d default: %8.1f us  %5.2f %%  kernel %d B%d S%d loop %d grid %dx%d" % (
        n, t_def * 1e6, 100 * 20.0 * n * n / t_def / PEAK, st["kernel_variant"], st["bodies_per_lane"], st["j_split"], st["inner_loop"],
        st["force_grid_x"], st["force_grid_y"]), flush=True)
    rows = []
    for kname, kv in (("sgprw", nbx.KERNEL_SGPRW), ("sgpr", nbx.KERNEL_SGPR)):
        for B in (2, 4):
            for S in (1, 2, 4, 8, 16, 32):
                kw = dict(kernel_variant=kv, bodies_per_lane=B, j_split=S, summation_order=nbx.ORDER_TREE)
                t, st = measure(n, ic, **kw)
                if t is None:
                    continue
                rows.append((t, "%-5s B%d S%-2d -> S%-2d loop %d grid %4dx%-2d epi %d" % (kname, B, S, st["j_split"], st["inner_loop"], st["force_grid_x"],
                                                                                     st["force_grid_y"], st["fused_epilogue"])))
    for NB in ((8, 16) if PREC == 32 else (8,)):
        t, st = measure(n, ic, kernel_variant=nbx.KERNEL_JLANE, bodies_per_lane=NB)
        if t is not None:
            rows.append((t, "jlane NB%d loop %d grid %dx%d" % (NB, st["inner_loop"], st["force_grid_x"], st["force_grid_y"])))
    rows.sort()
    for t, what in rows[:12]:
        print("    %8.1f us  %5.2f %%  %+5.1f %% vs default   %s" % (t * 1e6, 100 * 20.0 * n * n / t / PEAK, 100 * (t_def / t - 1), what), flush=True)
