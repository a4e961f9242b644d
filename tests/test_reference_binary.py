"""T1 (build container only -- skipped where oracle/_ref does not exist): the UNMODIFIED reference program
(oracle/_ref/nbody_ver7.x = /root/reference/ver7/{GSimulation,main}.cpp compiled as they lie) against the oracle and
against the wrapper-TU fixtures, so the chain reference -> fixtures -> oracle -> GPU has no unverified link."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

EXE = os.path.join(ROOT, "oracle", "_ref", "nbody_ver7.x")
pytestmark = pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref is only built where /root/reference exists")


def _rows(out):
    return [m.groups() for m in (re.match(r"^ (\d+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s*$", ln) for ln in out.splitlines()) if m]


def test_unmodified_reference_prints_what_the_fixture_and_the_oracle_say(oracle):
    p = subprocess.run([EXE, "1000", "100"], capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="4"))
    assert p.returncode == 0
    rows = _rows(p.stdout)
    assert [r[0] for r in rows] == ["50", "100"]
    g = load_golden("ver7_f32_n1000_s100.json")
    s = oracle.init_state(1000)
    ke = oracle.run(s, 100)
    for (step, _, kcol, _, _), idx in zip(rows, (49, 99)):
        assert kcol == "%.5g" % np.float32(g["kenergy"][idx])        # wrapper TU == unmodified program
        assert kcol == "%.5g" % ke[idx]                               # oracle == unmodified program (5 printed digits)
    lines = p.stdout.splitlines()
    assert lines[0] == "=" * 31 and lines[2] == " nPart = 1000; nSteps = 100; dt = 0.1"
    assert any(ln.startswith("# Average Perfomance : ") for ln in lines)


def test_reference_default_run_column(oracle):
    """`./nbody.x` with no arguments: the column BASELINE.md section 2 quotes for every version of the reference."""
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="8"))
    assert [r[2] for r in _rows(p.stdout)] == ["0.1432", "2.4341", "8.1256", "17.877", "32.966", "55.786", "91.132", "150.12", "264.78", "571.53"]
