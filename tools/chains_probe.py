"""How many partial chains per body may a kernel use at configs[2] before the kinetic energy leaves the reference's?  n = 262144 x 200
steps against the fixture from the reference's own binary (tests/golden/ver7_f32_n262144_s200.json); chains = contiguous j sub-ranges
summed separately and added in order (SGPR kernel with j_split = S; SGPRW: 4 S quarter-chains)."""
import json
import sys
sys.path.insert(0, 'nbody-demo-2023_amd')
import numpy as np
import nbx
g = json.load(open('tests/golden/ver7_f32_n262144_s200.json'))
ref = np.array(g['kenergy'][:200])
n = 262144
ic = nbx.initial_conditions(n)
for name, kw in (("1 chain (reference order)", dict(summation_order=nbx.ORDER_REFERENCE)),
                 ("2 chains", dict(kernel_variant=nbx.KERNEL_SGPR, j_split=2, bodies_per_lane=2, summation_order=nbx.ORDER_TREE)),
                 ("3 chains", dict(kernel_variant=nbx.KERNEL_SGPR, j_split=3, bodies_per_lane=2, summation_order=nbx.ORDER_TREE)),
                 ("4 chains", dict(kernel_variant=nbx.KERNEL_SGPR, j_split=4, bodies_per_lane=2, summation_order=nbx.ORDER_TREE)),
                 ("4 chains (wave split)", dict(kernel_variant=nbx.KERNEL_SGPRW, j_split=1, bodies_per_lane=4, summation_order=nbx.ORDER_TREE)),
                 ("8 chains", dict(kernel_variant=nbx.KERNEL_SGPR, j_split=8, bodies_per_lane=2, summation_order=nbx.ORDER_TREE)),
                 ("16 chains", dict(kernel_variant=nbx.KERNEL_SGPR, j_split=16, bodies_per_lane=2, summation_order=nbx.ORDER_TREE)),
                 ("default tree", dict(summation_order=nbx.ORDER_TREE))):
    with nbx.Context(n, 32, **kw) as c:
        c.upload(ic)
        ke = c.step_trace(200)
        st = c.stats()
    err = np.abs(ke - ref) / ref
    print("%-26s kernel %d B%d grid %dx%d: max rel err over 200 steps %.2e; at steps 50/100/150/200: %.2e %.2e %.2e %.2e" % (
        name, st['kernel_variant'], st['bodies_per_lane'], st['force_grid_x'], st['force_grid_y'], err.max(), err[49], err[99], err[149], err[199]), flush=True)
