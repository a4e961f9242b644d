#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REAL reference (oracle/_ref) -- build container only.

TEST INFRASTRUCTURE.  Runs the wrapper binaries that `make -C oracle ref` builds from the
reference's unmodified ver7 sources (which stay under /root/reference) and stores their output
as small JSON fixtures: per-step kinetic energy, CRC/sum/samples of the initial arrays, samples
of the final state.  The fixtures are data; no reference source text is stored.

usage: python oracle/gen_golden.py [--only f32:2000:500 ...]
"""
import argparse
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")

# (precision, n, nsteps): sized so the whole set regenerates in well under an hour on 8 vCPUs
CASES = [
    ("f32", 5, 20),
    ("f32", 63, 20),
    ("f32", 64, 20),
    ("f32", 65, 20),
    ("f32", 1000, 100),
    ("f32", 2000, 500),     # BASELINE.json configs[0]
    ("f32", 4099, 40),      # n not a multiple of any tile
    ("f32", 4096, 200),     # sizes the one-launch jlane kernel serves by default (round 2): NB = 2 ...
    ("f32", 8192, 200),     # ... NB = 4 ...
    ("f32", 12288, 100),    # ... and the largest one of round 2, NB = 4
    ("f32", 15000, 100),    # round 3: the one-launch kernel's range now ends at 16383 (NB = 8)
    ("f32", 16384, 500),    # BASELINE.json configs[1]
    ("f32", 65536, 20),
    ("f64", 5, 20),
    ("f64", 2000, 500),
    ("f64", 4099, 40),
    ("f64", 16384, 60),
]
# BASELINE.json configs[2] / configs[4] size: ~65 s per step here, generated once with
#   python oracle/gen_golden.py --only f32:262144:7 f64:262144:3
#   python oracle/gen_golden.py --only f32:32768:500 f32:65536:500    (310 s + 25 min: the chaotic regime)
#   OMP_NUM_THREADS=7 python oracle/gen_golden.py --only f32:262144:200   (2.7 h: the whole of configs[2])
#   python oracle/gen_golden.py --only f32:1048576:3     (40 min: the first steps at configs[3]'s size)
#   python oracle/gen_golden.py --only f64:262144:50     (over an hour: configs[4] up to its first printed row)
#   OMP_NUM_THREADS=6 python oracle/gen_golden.py --only f32:1048576:10    (2.5 h: ten steps at configs[3]'s size, round 2)
#   python oracle/gen_golden.py --only f32o3:16384:500    (1-2 min: configs[1] from the -O3 / AVX2 / FMA build of the reference, round 3)
BIG_CASES = [("f32", 262144, 7), ("f64", 262144, 3), ("f32", 32768, 500), ("f32", 65536, 500), ("f32", 262144, 200), ("f32", 1048576, 3), ("f64", 262144, 50),
             ("f32", 1048576, 10)]


def run_case(prec, n, steps, nsample=8):
    # "f32o3": the SAME unmodified source built with the best-effort flags (_ref/ver7_trace_o3.x: -O3 -march=x86-64-v3, FMA
    # contraction on) -- a second legitimate build of the reference, kept to show how far two builds of the reference drift
    # apart in the chaotic regime (n = 16384 x 500: tests/test_parity_gpu.py, profiles/r03_config1_divergence.json); never the oracle
    exe = os.path.join(HERE, "_ref", {"f32": "ver7_trace.x", "f64": "ver7_trace_f64.x", "f32o3": "ver7_trace_o3.x"}[prec])
    if not os.path.exists(exe):
        sys.exit("missing %s -- run `make -C oracle ref` in the build container first" % exe)
    out = os.path.join(GOLD, "ver7_%s_n%d_s%d.json" % (prec, n, steps))
    tmp = out + ".tmp"
    t0 = time.time()
    subprocess.check_call([exe, str(n), str(steps), tmp, str(nsample), "1"])
    d = json.load(open(tmp))
    os.remove(tmp)
    d["_provenance"] = {
        "source": "reference ver7/GSimulation.cpp compiled unmodified via oracle/ref_wrapper/ver7_trace.cpp",
        "flags": ("g++ -std=c++11 -O3 -march=x86-64-v3 -fopenmp -include mm_malloc.h" if prec == "f32o3" else
                  "g++ -std=c++11 -O2 -fopenmp -ffp-contract=off -include mm_malloc.h" + (" -DREF_F64" if prec == "f64" else "")),
        "variant": "fp32 (typedef float real_type)" if prec == "f32" else
                   "fp32, best-effort build (AVX2 + FMA contraction): a second build of the same source, NOT the pinned oracle" if prec == "f32o3" else
                   "fp64 arithmetic on the fp32-drawn initial conditions (SURVEY 8c variant B), dt=(double)0.1f",
        "generator": "oracle/gen_golden.py",
    }
    with open(out, "w") as f:
        json.dump(d, f, indent=None, separators=(",", ":"))
        f.write("\n")
    print("%-40s %.1fs" % (os.path.basename(out), time.time() - t0), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None, help="prec:n:steps ...")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    cases = CASES
    if a.only:
        cases = []
        for c in a.only:
            p, n, s = c.split(":")
            cases.append((p, int(n), int(s)))
    for c in cases:
        run_case(*c)


if __name__ == "__main__":
    main()
