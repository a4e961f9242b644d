"""Shared fixtures.  `-m "not gpu"` runs the oracle / host-logic / ABI-surface tests on CPU;
`-m gpu` runs the parity tests proper on an MI355X, through the C-ABI of libnbx.so."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "nbody-demo-2023_amd")
GOLD = os.path.join(ROOT, "tests", "golden")
for p in (PKG, os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure), built on demand with the pinned flags."""
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def nbx():
    """ctypes binding of libnbx.so; builds the library if the tree has none (hipcc cross-compiles)."""
    if not os.path.exists(os.path.join(PKG, "libnbx.so")):
        subprocess.check_call(["make", "-s", "-C", ROOT, "lib"])
    import nbx as N
    N.load()
    return N


@pytest.fixture(scope="session", autouse=True)
def _built_tree():
    """A fresh checkout has no binaries (they are git-ignored): build what the tests execute."""
    host = os.path.join(PKG, "host")
    if not all(os.path.exists(os.path.join(host, x)) for x in ("nbody.x", "nbody_fp64.x", "nbody_v5.x")) \
            or not os.path.exists(os.path.join(PKG, "libnbx.so")):
        subprocess.check_call(["make", "-s", "-C", ROOT, "lib", "host"])


def load_golden(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
