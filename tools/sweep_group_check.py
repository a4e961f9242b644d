"""sweep_group_check.py -- exercises scripts/sweep.py's MEASURED branch (nbx.Group over k devices) on a box that has only one, by
letting the k ranks share device 0 (logical ranks: same code path, copies instead of RCCL); not a measurement."""
import importlib.util
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("sweep_mod", os.path.join(ROOT, "scripts", "sweep.py"))
sw = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sw)
nbx = sw.nbx
orig = nbx.Group


def logical(n, precision=32, n_ranks=1, devices=None, **kw):
    return orig(n, precision, n_ranks=n_ranks, devices=[0] * n_ranks, **kw)


nbx.Group = logical
for n, k in ((4096, 2), (65536, 4), (262144, 8)):
    r = sw.time_group(n, k, 32, nbx.initial_conditions(n), target_s=0.2)
    print(n, k, r["ranks_used"], "%.1f us/step" % (1e6 * r["s_per_step"]), r["shape"], "kenergy %.9g" % r["kenergy"])
