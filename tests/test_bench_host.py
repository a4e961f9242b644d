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
