#!/usr/bin/env python3
"""Pair-rate sweep n = 2048 * 2^k (2k ... 1M) on one GPU: the table north_star asks for
(absolute pair/s and fraction of the fp32 vector-FMA roofline), plus P logical ranks on the one device
(same partition/exchange code path as the multi-GPU run, exchange by device-to-device copies).

usage: python scripts/sweep.py [--out profiles/r01_sweep.json] [--max-n 1048576] [--precision 32]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-demo-2023_amd"))
import nbx  # noqa: E402

PEAK = {32: 157.3e12, 64: 78.6e12}


def time_steps(n, precision, target_s=1.0, order=0):
    ic = nbx.initial_conditions(n, precision)
    with nbx.Context(n, precision, summation_order=order) as c:
        c.upload(ic)
        c.step(3, kenergy=False)
        c.sync()
        t0 = time.perf_counter()
        c.step(2, kenergy=False)
        c.sync()
        per = (time.perf_counter() - t0) / 2
        steps = max(5, min(2000, int(target_s / max(per, 1e-6))))
        c.step(min(steps, 100), kenergy=False)  # untimed: lets nbx_step capture and instantiate its hipGraph (launch-bound sizes)
        c.sync()
        # wall time first, as a user's run sees it (hipGraph replay of the launch-bound sizes is disabled while the
        # per-launch events of nbx_profile are recorded, so the two measurements are taken separately)
        t0 = time.perf_counter()
        c.step(steps, kenergy=False)
        c.sync()
        wall = time.perf_counter() - t0
        c.profile(True)
        c.step(min(steps, 200), kenergy=False)
        c.sync()
        st = c.stats()
    kms = st["force_ms_total"] / max(1, st["force_launches_timed"])
    return {"n": n, "steps": steps, "us_per_step": 1e6 * wall / steps, "pair_per_s": float(n) * n * steps / wall,
            "roofline_frac": 20.0 * float(n) * n * steps / wall / PEAK[precision],
            "force_kernel_us": 1e3 * kms, "force_kernel_frac": 20.0 * float(n) * n / (kms * 1e-3) / PEAK[precision] if kms else None,
            "bodies_per_lane": st["bodies_per_lane"], "j_split": st["j_split"], "grid": [st["force_grid_x"], st["force_grid_y"]],
            "kernel": {1: "lds", 2: "sgpr", 3: "sgprw", 6: "jlane"}[st["kernel_variant"]], "graph_replay": bool(st["use_graph"]),
            "inner_loop": {1: "cxx", 2: "asm", 3: "asm_ts"}.get(st["inner_loop"], "?"),
            "order": {1: "reference", 2: "tree"}[st["summation_order"]]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep.json"))
    ap.add_argument("--max-n", type=int, default=1048576)
    ap.add_argument("--precision", type=int, default=32)
    ap.add_argument("--order", default="auto", choices=("auto", "reference", "tree"))
    a = ap.parse_args()
    rows = []
    n = 2048
    print("%9s %8s %12s %14s %9s %12s %9s  shape" % ("n", "steps", "us/step", "G pair/s", "roof %", "kernel us", "kern %"))
    while n <= a.max_n:
        r = time_steps(n, a.precision, order={"auto": 0, "reference": 1, "tree": 2}[a.order])
        rows.append(r)
        print("%9d %8d %12.1f %14.1f %9.2f %12.1f %9.2f  %s %s%s B%d S%d %dx%d" % (
            r["n"], r["steps"], r["us_per_step"], r["pair_per_s"] * 1e-9, 100 * r["roofline_frac"], r["force_kernel_us"],
            100 * (r["force_kernel_frac"] or 0), r["order"], r["kernel"], ("/" + r["inner_loop"]) if r["inner_loop"] in ("asm", "asm_ts") else "", r["bodies_per_lane"], r["j_split"],
            r["grid"][0], r["grid"][1]), flush=True)
        n *= 2
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump({"precision": a.precision, "peak_flops": PEAK[a.precision], "flop_per_pair": 20, "rows": rows}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
