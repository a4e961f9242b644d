"""slice_run.py n own [cxx|asm|asm2|auto] [steps] -- steps ONE rank's slice (bodies [0, own) of n, reference summation order) a few times:
the program rocprofv3 is pointed at for the slice shapes (scripts/profile_slice.sh): what one rank of a multi-GPU run launches per step.
cxx = one body per lane, compiled loop (round 3's shape for slices of up to 65536 bodies); asm2 = two bodies per lane, plain hand-scheduled
loop (round 3's shape for 65537 ... 131072 bodies); auto = what the library takes now."""
import sys
sys.path.insert(0, 'nbody-demo-2023_amd')
import nbx
n, own = int(sys.argv[1]), int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else "auto"
loop = {"cxx": nbx.LOOP_CXX, "asm": nbx.LOOP_ASM, "asm2": nbx.LOOP_ASM, "auto": nbx.LOOP_AUTO}[which]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
kw = dict(bodies_per_lane=1) if which == "cxx" else dict(bodies_per_lane=2) if which == "asm2" else {}
with nbx.Context(n, 32, i_begin=0, i_count=own, n_alloc=n, summation_order=nbx.ORDER_REFERENCE, kernel_variant=nbx.KERNEL_SGPR, inner_loop=loop, **kw) as c:
    c.upload(nbx.initial_conditions(n))
    c.profile(True)
    for _ in range(steps):
        c.step_local(); c.commit()
    c.sync()
    st = c.stats()
ms = st["force_ms_total"] / st["force_launches_timed"]
print("n=%d own=%d B%d loop %d grid %dx%d: %.3f ms per launch = %.1f %% of the fp32 roofline" % (
    n, own, st["bodies_per_lane"], st["inner_loop"], st["force_grid_x"], st["force_grid_y"], ms, 100 * 20.0 * n * own / (ms * 1e-3) / 157.3e12))
