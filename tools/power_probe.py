"""Package power and shader clock (rocm-smi, sampled from a side thread) while one kernel variant steps n = 262144 for ~6 s each:
energy per pair = power / pair rate.  Observational: the hot kernel is power-limited, so the variant with the
fewest joules per pair is the one a power-capped box runs fastest.  usage: python tools/power_probe.py   (GPU box, repo root)"""
import json
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, 'nbody-demo-2023_amd')
import nbx

n = 262144
ic = nbx.initial_conditions(n)


def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        c = d[sorted(d)[0]]
        p = [float(v) for k, v in c.items() if "ower" in k and re.match(r"^[\d.]+$", str(v))]
        s = [int(m.group(1)) for k, v in c.items() if k.startswith("sclk") for m in [re.search(r"\((\d+)Mhz\)", str(v))] if m]
        return (p[0] if p else None), (s[0] if s else None)
    except Exception as e:
        return None, None


variants = [("reference B2 asm_ts (default)", dict(summation_order=nbx.ORDER_REFERENCE)),
            ("reference B2 asm (no time slicing)", dict(summation_order=nbx.ORDER_REFERENCE, inner_loop=nbx.LOOP_ASM)),
            ("reference B4 asm, 1 wave/SIMD", dict(summation_order=nbx.ORDER_REFERENCE, bodies_per_lane=4)),
            ("reference B2 compiled loop", dict(summation_order=nbx.ORDER_REFERENCE, inner_loop=nbx.LOOP_CXX)),
            ("tree SGPRW B4 S8, 8 waves/SIMD", dict(summation_order=nbx.ORDER_TREE)),
            ("tree LDS tile B8 S8", dict(summation_order=nbx.ORDER_TREE, kernel_variant=nbx.KERNEL_LDS, bodies_per_lane=8, j_split=8)),
            ("idle", None)]
print("%-40s %10s %8s %8s %12s %14s" % ("variant", "ms/step", "frac", "W", "sclk MHz", "J per G pair"))
for name, kw in variants:
    samples = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(sample())
            stop.wait(0.7)
    if kw is None:
        th = threading.Thread(target=sampler); th.start(); time.sleep(4); stop.set(); th.join()
        ws = [p for p, _ in samples if p]
        print("%-40s %10s %8s %8.0f" % (name, "-", "-", sum(ws) / max(1, len(ws))))
        continue
    with nbx.Context(n, 32, use_graph=2, **kw) as c:
        c.upload(ic)
        c.step(30, kenergy=False); c.sync()
        th = threading.Thread(target=sampler); th.start()
        t0 = time.perf_counter(); steps = 0
        while time.perf_counter() - t0 < 6.0:
            c.step(40, kenergy=False); c.sync(); steps += 40
        dt = (time.perf_counter() - t0) / steps
        stop.set(); th.join()
    ws = [p for p, _ in samples[1:] if p]
    cs = [s for _, s in samples[1:] if s]
    W = sum(ws) / max(1, len(ws)); rate = n * float(n) / dt
    print("%-40s %10.3f %8.4f %8.0f %12s %14.2f" % (name, dt * 1e3, 20 * rate / 157.3e12, W, ("%d" % (sum(cs) / len(cs))) if cs else "-", W / (rate * 1e-9)), flush=True)
