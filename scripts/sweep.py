#!/usr/bin/env python3
"""Pair-rate sweep n = 2048 * 2^k (2k ... 1M): the table north_star asks for (absolute pair/s and fraction of the fp32
vector-FMA roofline) -- on one GPU, and with `--gpus 1,2,4,8` the whole SURVEY.md 8(d) grid n x {1, 2, 4, 8} GPUs in one command:

  * a column k for which the box has >= k devices is MEASURED: the product's own single-process form, nbx.Group(n, n_ranks=k)
    over devices 0..k-1 (nbx_group_create -> ncclCommInitAll -> one grouped in-place ncclAllGather per step);
  * otherwise the cell is a PROXY, marked "measured": false: what ONE rank of k computes per step (rank 0's block of the
    library's partition, all n_alloc records resident) timed on this one GPU -- the k-GPU step before communication and skew.

usage: python scripts/sweep.py [--out profiles/r01_sweep.json] [--max-n 1048576] [--precision 32] [--gpus 1,2,4,8]
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
            "inner_loop": {1: "cxx", 2: "asm", 3: "asm_ts", 4: "asm_pf"}.get(st["inner_loop"], "?"),
            "order": {1: "reference", 2: "tree"}[st["summation_order"]]}


LOOPS = {1: "cxx", 2: "asm", 3: "asm_ts", 4: "asm_pf"}
KERNELS = {1: "lds", 2: "sgpr", 3: "sgprw", 4: "exact", 6: "jlane"}


def _shape(st):
    return {"bodies_owned": st["i_count"], "bodies_per_lane": st["bodies_per_lane"], "j_split": st["j_split"], "grid": [st["force_grid_x"], st["force_grid_y"]],
            "kernel": KERNELS.get(st["kernel_variant"], "?"), "inner_loop": LOOPS.get(st["inner_loop"], "?"),
            "order": {1: "reference", 2: "tree"}[st["summation_order"]]}


def time_slice(n, k, precision, ic, target_s=0.5, order=0):
    """One rank of k on this GPU: rank 0's block against all n_alloc resident records, step_local + commit per step."""
    P, block, i_begin, i_count, n_alloc = nbx.partition(n, k, 0)
    with nbx.Context(n, precision, i_begin=i_begin, i_count=i_count, n_alloc=n_alloc, summation_order=order) as c:
        c.upload(ic)

        def run(steps):
            for _ in range(steps):
                c.step_local()
                c.commit()
            c.sync()
        run(3)
        t0 = time.perf_counter()
        run(2)
        per = (time.perf_counter() - t0) / 2
        steps = max(5, min(2000, int(target_s / max(per, 1e-6))))
        t0 = time.perf_counter()
        run(steps)
        wall = (time.perf_counter() - t0) / steps
        c.profile(True)
        run(min(steps, 100))
        st = c.stats()
    return {"ranks_used": P, "block": block, "steps": steps, "s_per_step": wall, "force_kernel_us": 1e3 * st["force_ms_total"] / max(1, st["force_launches_timed"]),
            "shape": _shape(st)}


def time_group(n, k, precision, ic, target_s=0.5, order=0):
    """The product's single-process multi-GPU form on k distinct devices (measured)."""
    with nbx.Group(n, precision, n_ranks=k, devices=list(range(k)), summation_order=order) as g:
        g.upload(ic)
        P, rccl, st = g.info(0)
        g.step(3)
        t0 = time.perf_counter()
        g.step(2)
        per = (time.perf_counter() - t0) / 2
        steps = max(5, min(2000, int(target_s / max(per, 1e-6))))
        t0 = time.perf_counter()
        ke = g.step(steps)
        wall = (time.perf_counter() - t0) / steps
    return {"ranks_used": P, "steps": steps, "s_per_step": wall, "uses_rccl": rccl, "kenergy": ke, "shape": _shape(st)}


def cpu_baseline(precision):
    """north_star: "alongside the reference ver7 OpenMP CPU path timed on the same box's host cores (core count stated) in the same
    run" -- bench.py's cpu_baseline leg (the reference's own ver7 binary from oracle/_ref when it is there, else the oracle's C
    restatement), run BEFORE this process touches the GPU."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench.cpu_baseline("auto", 262144, precision)


def multi_gpu_grid(a, gpus):
    """n x gpus grid; every cell says whether it was measured on that many devices or is the one-GPU slice proxy."""
    cpu = None if a.no_cpu else cpu_baseline(a.precision)
    if cpu:
        print("CPU baseline: %.3g pair/s on %d threads (%s)" % (cpu["value"], cpu["cores"], cpu["sample"]), flush=True)
    try:
        import torch
        ndev = torch.cuda.device_count()
    except Exception:
        ndev = 1
    order = {"auto": 0, "reference": 1, "tree": 2}[a.order]
    cells, n = [], 2048
    print("devices on this box: %d; columns with more GPUs than that are one-GPU slice proxies (marked *)" % ndev)
    print("%9s %5s %12s %14s %9s %9s  shape of one rank" % ("n", "gpus", "us/step", "G pair/s", "roof %", "speed-up"))
    while n <= a.max_n:
        ic = nbx.initial_conditions(n, a.precision)
        one = time_steps(n, a.precision, target_s=0.5, order=order)
        t1 = one["us_per_step"] * 1e-6
        for k in gpus:
            if k == 1:
                cell = {"n": n, "gpus": 1, "measured": True, "ranks_used": 1, "s_per_step": t1, "force_kernel_us": one["force_kernel_us"],
                        "shape": {kk: one[kk] for kk in ("bodies_per_lane", "j_split", "grid", "kernel", "inner_loop", "order")}, "graph_replay": one["graph_replay"]}
            elif k <= ndev:
                cell = dict(time_group(n, k, a.precision, ic, order=order), n=n, gpus=k, measured=True)
            else:
                cell = dict(time_slice(n, k, a.precision, ic, order=order), n=n, gpus=k, measured=False,
                            note="one rank's block timed on ONE GPU (plain launches, no exchange): the %d-GPU step before communication and skew" % k)
            t = cell["s_per_step"]
            cell["pair_per_s"] = float(n) * n / t
            cell["roofline_frac_of_all_gpus"] = 20.0 * float(n) * n / t / (PEAK[a.precision] * cell["ranks_used"])
            cell["speedup_vs_one_gpu" if cell["measured"] else "implied_speedup_before_communication"] = t1 / t
            cells.append(cell)
            sh = cell["shape"]
            print("%9d %4d%s %12.1f %14.1f %9.2f %9.2f  %s %s/%s B%d S%d %dx%d%s" % (
                n, k, " " if cell["measured"] else "*", 1e6 * t, cell["pair_per_s"] * 1e-9, 100 * cell["roofline_frac_of_all_gpus"], t1 / t, sh["order"], sh["kernel"],
                sh["inner_loop"], sh["bodies_per_lane"], sh["j_split"], sh["grid"][0], sh["grid"][1],
                "" if cell["ranks_used"] == k else "  (%d ranks: no rank may be empty)" % cell["ranks_used"]), flush=True)
        n *= 2
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump({"precision": a.precision, "peak_flops_per_gpu": PEAK[a.precision], "flop_per_pair": 20, "devices_on_box": ndev, "cpu_baseline": cpu,
               "what": "SURVEY.md 8(d) grid n = 2048 * 2^k x GPUs; measured = run on that many devices (nbx.Group, single process, RCCL), otherwise the one-GPU "
                       "slice proxy: rank 0's block of the library's partition against all resident records, before communication and skew",
               "cells": cells}, open(a.out, "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep.json"))
    ap.add_argument("--max-n", type=int, default=1048576)
    ap.add_argument("--precision", type=int, default=32)
    ap.add_argument("--order", default="auto", choices=("auto", "reference", "tree"))
    ap.add_argument("--gpus", default="", help="comma-separated GPU counts, e.g. 1,2,4,8: the n x GPUs grid (measured where the box has the devices, else the slice proxy)")
    ap.add_argument("--no-cpu", action="store_true", help="--gpus grid: skip the CPU baseline (the reference's ver7 on this box's cores, ~20 s)")
    a = ap.parse_args()
    if a.gpus:
        return multi_gpu_grid(a, [int(x) for x in a.gpus.split(",")])
    rows = []
    n = 2048
    print("%9s %8s %12s %14s %9s %12s %9s  shape" % ("n", "steps", "us/step", "G pair/s", "roof %", "kernel us", "kern %"))
    while n <= a.max_n:
        r = time_steps(n, a.precision, order={"auto": 0, "reference": 1, "tree": 2}[a.order])
        rows.append(r)
        print("%9d %8d %12.1f %14.1f %9.2f %12.1f %9.2f  %s %s%s B%d S%d %dx%d" % (
            r["n"], r["steps"], r["us_per_step"], r["pair_per_s"] * 1e-9, 100 * r["roofline_frac"], r["force_kernel_us"],
            100 * (r["force_kernel_frac"] or 0), r["order"], r["kernel"], ("/" + r["inner_loop"]) if r["inner_loop"].startswith("asm") else "", r["bodies_per_lane"], r["j_split"],
            r["grid"][0], r["grid"][1]), flush=True)
        n *= 2
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump({"precision": a.precision, "peak_flops": PEAK[a.precision], "flop_per_pair": 20, "rows": rows}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
