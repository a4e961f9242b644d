import sys, json
sys.path.insert(0, 'nbody-demo-2023_amd')
import numpy as np, nbx
g = json.load(open('tests/golden/ver7_f32_n16384_s500.json'))
ref = np.array(g['kenergy'])
n = 16384
ic = nbx.initial_conditions(n)
for name, kw in (("sgprw", dict(kernel_variant=nbx.KERNEL_SGPRW)), ("sgprw S4", dict(kernel_variant=nbx.KERNEL_SGPRW, j_split=4)), ("sgprw S8", dict(kernel_variant=nbx.KERNEL_SGPRW, j_split=8)),
                 ("sgprw S16", dict(kernel_variant=nbx.KERNEL_SGPRW, j_split=16)), ("sgprw S2", dict(kernel_variant=nbx.KERNEL_SGPRW, j_split=2)),
                 ("sgpr S8", dict(kernel_variant=nbx.KERNEL_SGPR, j_split=8)), ("sgpr S16", dict(kernel_variant=nbx.KERNEL_SGPR, j_split=16)), ("lds", dict(kernel_variant=nbx.KERNEL_LDS)), ("jlane8", dict(kernel_variant=nbx.KERNEL_JLANE, bodies_per_lane=8)),
                 ("jlane16", dict(kernel_variant=nbx.KERNEL_JLANE, bodies_per_lane=16)), ("jlane4", dict(kernel_variant=nbx.KERNEL_JLANE, bodies_per_lane=4)),
                 ("ref-order B2", dict(summation_order=nbx.ORDER_REFERENCE))):
    with nbx.Context(n, 32, **kw) as c:
        c.upload(ic)
        ke = c.step_trace(500)
    e = np.abs(ke - ref) / ref
    print("%-12s printed max %.2e  all max %.2e  first200 max %.2e  printed: %s" % (name, max(e[k-1] for k in range(50, 501, 50)), e.max(), e[:200].max(),
          " ".join("%.1e" % e[k-1] for k in range(50, 501, 50))), flush=True)
