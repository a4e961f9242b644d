"""bench.py's host-side pieces that need no GPU: core counting and the CPU-baseline leg (the oracle /
oracle/_ref are allowed here: this is the cpu_baseline leg itself)."""
import importlib.util
import os

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_usable_cpus_is_positive_and_bounded(bench):
    n = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)


@pytest.mark.parametrize("kind", ["port", "reference"])
def test_cpu_baseline_leg(bench, oracle, kind):
    if kind == "reference" and not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ver7_trace.x")):
        pytest.skip("oracle/_ref is only built where /root/reference exists")
    cores = bench.usable_cpus()
    c = bench.cpu_baseline(kind, 32768, 32)
    assert c["kind"] == kind and c["unit"] == "pair/s" and c["cores"] == cores
    assert bench.usable_cpus() == cores     # the leg must not re-pin the benchmark process
    assert 1e8 < c["value"] < 1e12          # a CPU does 0.1-1 G pair/s per core on this loop
    assert "n=" in c["sample"] and "steps" in c["sample"]


def test_constants_match_the_scope_contract(bench):
    assert bench.FLOP_PER_PAIR == 20 and bench.PEAK_FP32_VECTOR_TFLOPS == 157.3 and bench.PEAK_FP64_VECTOR_TFLOPS == 78.6


def test_divergence_curve_of_the_two_reference_builds(bench):
    """The fixtures of BASELINE.json configs[1] from two builds of the reference's unmodified source (pinned -O2 / SSE2 and
    -O3 / AVX2 / FMA): the reference is only defined to ~1e-4 late in this chaotic run (1.3e-4 at
    step 450), which is the yardstick the GPU's own curve is read against (tests/test_parity_gpu.py, bench.py --bodies 16384)."""
    import json
    a = json.load(open(os.path.join(ROOT, "tests", "golden", "ver7_f32_n16384_s500.json")))
    b = json.load(open(os.path.join(ROOT, "tests", "golden", "ver7_f32o3_n16384_s500.json")))
    assert b["n"] == 16384 and b["nsteps"] == 500 and "x86-64-v3" in b["_provenance"]["flags"]
    assert b["init"] == a["init"]  # same particles, bit for bit
    d = bench.divergence_vs_reference_builds(a["kenergy"], a["kenergy"], b["kenergy"])  # "gpu" := the pinned build itself
    assert d["max_over_all_steps"]["gpu_vs_pinned_build"] == 0.0
    spread = d["max_over_all_steps"]["second_build_vs_pinned_build"]
    assert 1e-4 < spread < 1e-3                                                     # measured 3.0e-4 at step 480
    assert abs(d["printed_rows"]["450"]["second_build_vs_pinned_build"] - 1.285e-4) < 2e-6  # SURVEY G2's 1.3e-4
    assert d["printed_rows"]["50"]["second_build_vs_pinned_build"] < 1e-6 and sorted(map(int, d["printed_rows"])) == list(range(50, 501, 50))
    assert 150 < d["first_step_above_1e-5"]["second_build_vs_pinned_build"] < 300
