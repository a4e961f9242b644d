import sys, os, json
sys.path.insert(0, 'nbody-demo-2023_amd'); sys.path.insert(0, 'oracle')
import numpy as np, nbx, oracle as O
n = 4099
s = O.init_state(n).astype(np.float64); O.accel(s)
with nbx.Context(n, 64) as c:
    c.upload(nbx.initial_conditions(n, 64)); ax, ay, az = c.accel()
sc = max(abs(s.acc_x).max(), abs(s.acc_y).max(), abs(s.acc_z).max())
print(os.environ.get('NBX_LIB'), 'acc err', max(abs(ax - s.acc_x).max(), abs(ay - s.acc_y).max(), abs(az - s.acc_z).max()) / sc)
g = json.load(open('tests/golden/ver7_f64_n2000_s500.json'))
with nbx.Context(2000, 64) as c:
    c.upload(nbx.initial_conditions(2000, 64)); ke = c.step_trace(500)
r = np.array(g['kenergy']); print('ke err', (abs(ke - r) / r).max())
with nbx.Context(262144, 64) as c:
    c.upload(nbx.initial_conditions(262144, 64)); c.profile(True); c.step(3); st = c.stats()
print('ms', st['force_ms_total'] / 3, 'frac', 20 * 262144.0**2 / (st['force_ms_total'] / 3 * 1e-3) / 78.6e12)
