"""Drop-in proof (build container only: skipped where /root/reference does not exist, e.g. on the GPU box).

The reference's OWN, unmodified callers -- ver7/main.cpp, ver8/main.cpp (ver7/main.cpp:25-46) and ver5_all/main.cpp
(:23-66, which needs init_mpi(), world_rank and the <iostream>/<string> its GSimulation.hpp provides,
ver5_all/GSimulation.hpp:25-30,60-65) -- are compiled against host/GSimulation.{hpp,cpp} + libnbx and run.

The sources are read where they lie, at compile time only: each main.cpp is fed to g++ on stdin, so its
`#include "GSimulation.hpp"` cannot resolve to the header next to it and must find host/GSimulation.hpp through -I
(checked with -H).  Nothing of the reference is copied into the repo and no binary built here travels.
"""
import os
import subprocess

import pytest

from conftest import PKG, ROOT, has_gpu

REF = "/root/reference"
HOST = os.path.join(PKG, "host")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "ver7")), reason="the reference sources exist only in the build container")

MAINS = [("ver7", []), ("ver8", []), ("ver5_all", ["-DNBX_BANNER_IN_MAIN"])]


def _build(tmp_path, version, defs):
    if not os.path.exists(os.path.join(PKG, "libnbx.so")):
        subprocess.check_call(["make", "-s", "-C", ROOT, "lib"])
    exe = str(tmp_path / ("nbody_%s_main.x" % version))
    obj = str(tmp_path / "main.o")
    src = open(os.path.join(REF, version, "main.cpp")).read()
    # main.cpp from stdin, cwd = an empty directory: the only GSimulation.hpp in reach is host/'s
    p = subprocess.run(["g++", "-std=c++14", "-O2", "-H", "-x", "c++", "-", "-I", HOST, "-c", "-o", obj], input=src, text=True,
                       capture_output=True, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr
    headers = [ln.split()[-1] for ln in p.stderr.splitlines() if ln.startswith(".")]
    assert os.path.join(HOST, "GSimulation.hpp") in [os.path.normpath(h) for h in headers], headers
    assert not any(os.path.normpath(h).startswith(REF) for h in headers), headers
    subprocess.check_call(["g++", "-std=c++14", "-O2"] + defs + [os.path.join(HOST, "GSimulation.cpp"), obj, "-o", exe,
                                                                  "-L" + PKG, "-lnbx", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


@pytest.mark.parametrize("version,defs", MAINS, ids=[m[0] for m in MAINS])
def test_unmodified_reference_main_builds_and_runs_against_the_drop_in(tmp_path, version, defs):
    exe = _build(tmp_path, version, defs)
    args = ["300", "100"] + (["gpu", "0.5", "256", "2"] if version == "ver5_all" else [])
    p = subprocess.run([exe] + args, capture_output=True, text=True, timeout=120)
    lines = p.stdout.splitlines()
    if version == "ver5_all":
        assert lines[0] == "gpu"  # ver5_all/main.cpp:42 echoes the device word before the banner
        lines = lines[1:]
    assert lines[0] == "=" * 31 and lines[1] == " Initialize Gravity Simulation"
    assert lines[2] == " nPart = 300; nSteps = 100; dt = 0.1"
    assert lines[3] == "-" * 48
    if has_gpu():
        assert p.returncode == 0 and any(ln.startswith("# Average Perfomance : ") for ln in lines), p.stderr
    else:  # the product has no CPU path: the reference's caller gets the header, then a loud failure
        assert p.returncode != 0 and "no HIP device" in p.stderr


def test_reference_ver5_main_device_word_cpu_is_refused(tmp_path):
    exe = _build(tmp_path, "ver5_all", ["-DNBX_BANNER_IN_MAIN"])
    p = subprocess.run([exe, "300", "100", "cpu"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 1 and "no CPU engine" in p.stderr
    assert p.stdout.splitlines()[:3] == ["cpu", "=" * 31, " Initialize Gravity Simulation"]
