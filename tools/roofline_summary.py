#!/usr/bin/env python3
"""Condense the rocprofv3 passes of scripts/profile.sh into profiles/<tag>_*.{csv,json,md}.

usage: python tools/roofline_summary.py gpurun_out/prof r01 [n] [gpus] [precision]

HBM-side bytes follow MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are KiB per dispatch;
on gfx950 FETCH_SIZE tallies the 128-B requests of wide (16 B/lane) coalesced reads at 64 B, so the
guide prescribes doubling the read side; WRITE_SIZE is exact for 16-B-per-lane stores.  That x2 is
calibrated for vector loads: the SGPR kernels fetch their j records with 64-B SCALAR loads, for which
it does not apply, so BOTH figures are reported (hbm_bytes_per_launch = guide-corrected upper bound,
hbm_bytes_per_launch_fetch_x1 = raw FETCH_SIZE + WRITE_SIZE).  The counters sit on the L2's fabric
side, so Infinity-Cache hits are included: this is "bytes that left the XCD L2s", an upper bound on
HBM bytes.

Algorithmic bytes are computed for the kernel instance that was actually profiled (template arguments
and grid from the trace), not from a fixed shape: row epilogue (EPI 1) 16n + 48*own,
slab epilogue (EPI 0/2) 16n + 16*own*S for the force kernel alone (fp64: records twice as large).
"""
import re
import collections
import csv
import glob
import json
import os
import shutil
import sys


def dominant_kernel(path, match):
    """A run may launch several instances of the kernel template (e.g. bench.py's parity probe uses the default
    shape): keep the instance with the most dispatches."""
    names = collections.Counter(r["Kernel_Name"] for r in csv.DictReader(open(path)) if match in r["Kernel_Name"])
    return names.most_common(1)[0][0] if names else match


def per_kernel(path, match):
    match = dominant_kernel(path, match)
    acc = collections.defaultdict(list)
    dur = []
    meta = {}
    for r in csv.DictReader(open(path)):
        if match != r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                                  "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size")}
    out = {k: sum(v) / len(v) for k, v in acc.items()}
    return out, (sum(dur) / len(dur) / 1e6 if dur else None), meta


def newest(pattern):
    """gpurun merges every run into the same directories: take the most recent file that matches."""
    f = glob.glob(pattern)
    return [max(f, key=os.path.getmtime)] if f else []


def main():
    src, tag = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
    gpus = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    precision = int(sys.argv[5]) if len(sys.argv) > 5 else 32
    rec = 16 if precision == 32 else 32
    peak_flops = 20.0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = os.path.join(root, "profiles")
    os.makedirs(dst, exist_ok=True)
    match = "force_kernel"

    stats = newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
    rows = []
    if stats:
        shutil.copy(stats[0], os.path.join(dst, "%s_kernel_stats.csv" % tag))
        rows = list(csv.DictReader(open(stats[0])))
    counters, meta, durs = {}, {}, {}
    for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_grbm"):
        f = newest(os.path.join(src, d, "*", "*_counter_collection.csv"))
        if not f:
            continue
        c, ms, m = per_kernel(f[0], match)
        counters.update(c)
        durs[d] = ms
        meta = m or meta

    fetch_kib = counters.get("FETCH_SIZE")
    write_kib = counters.get("WRITE_SIZE")
    read_bytes = None if fetch_kib is None else 2.0 * fetch_kib * 1024.0
    write_bytes = None if write_kib is None else write_kib * 1024.0
    traffic = None if (read_bytes is None or write_bytes is None) else read_bytes + write_bytes
    force_rows = sorted((r for r in rows if match in r["Name"]), key=lambda r: -int(r["Calls"]))
    force_row = force_rows[0] if force_rows else None
    avg_ms = float(force_row["AverageNs"]) / 1e6 if force_row else None
    clock_ghz = None
    if "GRBM_GUI_ACTIVE" in counters and durs.get("pmc_grbm"):
        clock_ghz = counters["GRBM_GUI_ACTIVE"] / 8.0 / (durs["pmc_grbm"] * 1e-3) * 1e-9
    pairs = float(n) * float(n) / gpus
    valu_insts = counters.get("SQ_INSTS_VALU")
    # shape of the profiled instance: force_kernel<T, B, JSRC, EPI, MINW, MATH, WSPLIT, LOOP>
    own = n // gpus
    shape = {"bodies_per_lane": None, "epilogue": None, "j_split": None, "wave_split": None, "loop": None}
    alg_bytes = None
    m = re.search(r"force_kernel<(\w+), (\d+), (\d+), (\d+), (\d+), (\d+), (\w+)(?:, (\d+))?>", meta.get("Kernel_Name", ""))
    if m:
        B, epi, ws = int(m.group(2)), int(m.group(4)), m.group(7) in ("true", "1")
        wgx = -(-own // ((64 if ws else 256) * B))
        S = max(1, int(meta.get("Grid_Size", 0)) // (256 * wgx)) if meta.get("Grid_Size") else None
        shape = {"bodies_per_lane": B, "epilogue": {0: "slab", 1: "row"}.get(epi), "j_split": S, "wave_split": ws,
                 "loop": {None: "cxx", "0": "cxx", "1": "asm", "2": "asm_ts"}.get(m.group(8), m.group(8))}
        alg_bytes = float(rec) * n + (3.0 * rec * own if epi == 1 else float(rec) * own * (S or 1))
    read_x1 = None if fetch_kib is None else fetch_kib * 1024.0
    traffic_x1 = None if (read_x1 is None or write_bytes is None) else read_x1 + write_bytes
    summary = {
        "tag": tag, "n": n, "gpus": gpus, "kernel": meta.get("Kernel_Name"), "launch": meta,
        "avg_launch_ms_kernel_trace": avg_ms, "avg_launch_ms_pmc_passes": durs,
        "counters_per_launch": counters,
        "hbm_side_read_bytes_per_launch": read_bytes, "hbm_side_write_bytes_per_launch": write_bytes,
        "hbm_bytes_per_launch": traffic,
        "precision": precision,
        "hbm_bytes_per_launch_fetch_x1": traffic_x1,
        "profiled_shape": shape,
        "algorithmic_bytes_per_launch": alg_bytes,
        "traffic_over_algorithmic": None if not (traffic and alg_bytes) else traffic / alg_bytes,
        "traffic_fetch_x1_over_algorithmic": None if not (traffic_x1 and alg_bytes) else traffic_x1 / alg_bytes,
        "effective_clock_ghz": clock_ghz,
        "valu_wave_insts_per_64_pairs": None if not valu_insts else valu_insts / (pairs / 64.0),
        "valu_busy_fraction": None,
        "achieved_tflops_20flop_per_pair": None if not avg_ms else 20.0 * pairs / (avg_ms * 1e-3) * 1e-12,
    }
    if "SQ_ACTIVE_INST_VALU" in counters and clock_ghz and durs.get("pmc_sq"):
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over all SIMDs (MI355X_MICROARCH.md cycle-constants)
        busy_cycles = counters["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0
        summary["valu_busy_fraction"] = busy_cycles / (durs["pmc_sq"] * 1e-3 * clock_ghz * 1e9)
    if traffic and avg_ms:
        summary["hbm_side_GBps"] = traffic / (avg_ms * 1e-3) * 1e-9
    json.dump(summary, open(os.path.join(dst, "%s_roofline_summary.json" % tag), "w"), indent=1)
    if traffic and precision == 32:
        json.dump({"n": n, "gpus": gpus, "precision": precision, "hbm_bytes_per_launch": traffic,
                   "hbm_bytes_per_launch_fetch_x1": traffic_x1, "algorithmic_bytes_per_launch": alg_bytes, "profiled_shape": shape,
                   "source": "profiles/%s_roofline_summary.json" % tag},
                  open(os.path.join(dst, "roofline_traffic.json"), "w"))
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
