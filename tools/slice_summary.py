#!/usr/bin/env python3
"""slice_summary.py DIR TAG N OWN -- condenses the rocprofv3 passes of scripts/profile_slice.sh (one rank's slice of a reference-order
run; `cxx` = one body per lane with the compiled loop, round 3's shape, `auto` = what the library takes now) into
profiles/TAG_slice_pmc_summary.json: per launch of the force kernel, averaged over the profiled launches.

Units (MI355X_MICROARCH.md, profiling section): SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / SQ_ACTIVE_INST_VALU / SQ_WAIT_INST_ANY count in units of
four shader cycles, summed over the waves (resp. SIMDs / SEs); GRBM_GUI_ACTIVE counts shader-clock cycles summed over the 8 XCDs;
FETCH_SIZE is KiB entering the XCD L2s from the fabric."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def counters(d, variant, pas):
    f = sorted(glob.glob(os.path.join(d, "%s_pmc_%s" % (variant, pas), "*", "*counter_collection.csv")), key=os.path.getmtime)[-1]
    acc, dur, name, grid = collections.defaultdict(list), [], None, None
    seen = set()
    for r in csv.DictReader(open(f)):
        if "force_kernel" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        name, grid = r["Kernel_Name"].split("(")[0], int(r["Grid_Size"])
    return {k: sum(v) / len(v) for k, v in acc.items()}, sum(dur) / len(dur) * 1e-6, name, grid, len(seen)


def main():
    d, tag, n, own = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    before = sys.argv[5] if len(sys.argv) > 5 else "cxx"
    out = {"what": "one rank's slice: bodies [0, %d) of n = %d, reference summation order, force kernel with the row epilogue; per launch" % (own, n),
           "n": n, "bodies_owned": own, "simds_on_chip": 1024, "variants": {}}
    labels = {"cxx": "before (round 3): one body per lane, compiled loop, plain VALU instructions",
              "asm2": "before (round 3): two bodies per lane, plain hand-scheduled loop (64-record trips, no L2 prefetch)"}
    for v, label in ((before, labels[before]), ("auto", "after (round 4): what the library takes now")):
        sq, ms_sq, name, grid, launches = counters(d, v, "sq")
        grbm, ms_g, _, _, _ = counters(d, v, "grbm")
        fetch, ms_f, _, _, _ = counters(d, v, "fetch")
        stats = sorted(glob.glob(os.path.join(d, "%s_stats" % v, "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1]
        shutil.copy(stats, os.path.join("profiles", "%s_slice_%s_kernel_stats.csv" % (tag, v)))
        avg_ns = None
        for r in csv.DictReader(open(stats)):
            if "force_kernel" in r["Name"]:
                avg_ns = float(r["AverageNs"])
        waves = sq["SQ_WAVES"]
        clk = grbm["GRBM_GUI_ACTIVE"] / 8.0 / (ms_g * 1e-3) * 1e-9      # GHz: cycles per XCD / duration of the same pass
        cyc_per_wave = 4.0 * sq["SQ_WAVE_CYCLES"] / waves
        t = avg_ns * 1e-9
        out["variants"][v] = {
            "label": label, "kernel": name, "grid_workgroups": grid // 256, "launches_profiled": launches,
            "avg_launch_ms_kernel_trace": avg_ns * 1e-6, "roofline_frac": 20.0 * n * own / t / 157.3e12,
            "SQ_WAVES": waves, "waves_per_simd": waves / 1024.0, "simds_without_a_wave": max(0.0, 1024.0 - waves),
            "SQ_WAVE_CYCLES_x4_per_wave": cyc_per_wave, "cycles_per_j_record_per_wave": cyc_per_wave / n,
            "GRBM_GUI_ACTIVE_per_xcd": grbm["GRBM_GUI_ACTIVE"] / 8.0, "shader_clock_ghz": clk,
            "SQ_INSTS_VALU_per_wave_per_record": sq["SQ_INSTS_VALU"] / waves / n,
            "SQ_INSTS_SALU_per_wave_per_record": sq["SQ_INSTS_SALU"] / waves / n,
            "SQ_INSTS_SMEM_per_wave_per_record": sq["SQ_INSTS_SMEM"] / waves / n,
            "valu_busy_of_wave_cycles": sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"],
            "waiting_of_wave_cycles": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
            "chip_valu_utilisation": sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"] * min(1.0, waves / 1024.0),
            "FETCH_SIZE_MB": fetch["FETCH_SIZE"] * 1024 / 1e6, "raw": {**sq, **grbm, **fetch},
        }
    a, b = out["variants"][before], out["variants"]["auto"]
    out["speedup"] = a["avg_launch_ms_kernel_trace"] / b["avg_launch_ms_kernel_trace"]
    path = os.path.join("profiles", "%s_slice_pmc_summary.json" % tag)
    json.dump(out, open(path, "w"), indent=1)
    for v in (before, "auto"):
        x = out["variants"][v]
        print("%-5s %s: %d workgroups, %.3f ms, %.1f %% of the roofline; %.2f waves/SIMD; %.1f cycles per j record per wave at %.2f GHz; VALU %.2f, SALU %.3f, SMEM %.3f instructions "
              "per record; VALU busy %.3f of the wave's cycles, waiting %.3f; FETCH %.1f MB" % (
                  v, x["kernel"][-40:], x["grid_workgroups"], x["avg_launch_ms_kernel_trace"], 100 * x["roofline_frac"], x["waves_per_simd"], x["cycles_per_j_record_per_wave"],
                  x["shader_clock_ghz"], x["SQ_INSTS_VALU_per_wave_per_record"], x["SQ_INSTS_SALU_per_wave_per_record"], x["SQ_INSTS_SMEM_per_wave_per_record"],
                  x["valu_busy_of_wave_cycles"], x["waiting_of_wave_cycles"], x["FETCH_SIZE_MB"]))
    print("wrote", path)


if __name__ == "__main__":
    main()
