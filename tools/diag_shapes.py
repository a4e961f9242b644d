"""One launch shape per child process, in order, stop at the first one that dies (diagnostic)."""
import subprocess, sys, os
shapes = [(b, s, k) for b in (1, 2, 4, 8) for s in (1, 3, 16) for k in (1, 2, 3) if not (k == 3 and b == 8)]
code = r'''
import sys; sys.path.insert(0, "nbody-demo-2023_amd")
import nbx, numpy as np
b, s, k = %d, %d, %d
n = 4099
with nbx.Context(n, 32, bodies_per_lane=b, j_split=s, kernel_variant=k, fused_epilogue=2) as c:
    c.upload(nbx.initial_conditions(n))
    st = c.stats()
    print("shape", b, s, k, "->", st["bodies_per_lane"], st["j_split"], st["kernel_variant"], st["force_grid_x"], st["force_grid_y"], flush=True)
    ax, ay, az = c.accel()
    print("   ok", float(np.abs(ax).max()), flush=True)
'''
env = dict(os.environ, LIBC_FATAL_STDERR_="1", AMD_LOG_LEVEL="1")
for b, s, k in shapes:
    p = subprocess.run([sys.executable, "-c", code % (b, s, k)], env=env, capture_output=True, text=True, timeout=120)
    sys.stdout.write(p.stdout)
    if p.returncode != 0:
        print("DIED rc", p.returncode)
        print(p.stderr[-3000:])
        sys.exit(1)
