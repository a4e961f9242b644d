"""T3 (CPU part): libnbx.so loads, exports every symbol include/nbx.h declares, rejects bad
arguments with the documented codes, fails LOUDLY without a GPU, and its host-side initial
conditions are bit-exact against the reference fixtures.  No GPU compute here."""
import ctypes
import os
import re
import subprocess
import zlib

import numpy as np
import pytest

from conftest import ROOT, PKG, has_gpu, load_golden


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "nbx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nbx_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported(nbx):
    names = declared_functions()
    assert set(names) == set(nbx.SYMBOLS)
    L = nbx.load()
    for s in names:
        assert getattr(L, s) is not None
    out = subprocess.check_output(["nm", "-D", "--defined-only", nbx.LIB_PATH]).decode()
    for s in names:
        assert re.search(r" T %s$" % s, out, flags=re.M), s


def test_abi_version_and_struct_sizes(nbx):
    assert nbx.load().nbx_abi_version() == 1
    assert ctypes.sizeof(nbx.Opts) == 72
    assert ctypes.sizeof(nbx.Stats) % 8 == 0


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "nbx.h"\nint main(void){nbx_opts o; nbx_stats_t s; (void)o; (void)s; return sizeof(o)==72?0:1;}\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    subprocess.check_call([str(exe)])


@pytest.mark.parametrize("n,prec", [(0, 32), (-5, 32), (100, 16), (100, 0)])
def test_create_rejects_bad_arguments(nbx, n, prec):
    with pytest.raises(nbx.NbxError) as e:
        nbx.Context(n, prec)
    assert e.value.code == nbx.NBX_ERR_ARG


def test_create_rejects_bad_slice_and_null(nbx):
    for kw in ({"i_begin": 100, "i_count": 1}, {"i_begin": 10, "i_count": 95}, {"n_alloc": 50}, {"i_begin": -1}):
        with pytest.raises(nbx.NbxError) as e:
            nbx.Context(100, 32, **kw)
        assert e.value.code == nbx.NBX_ERR_ARG, kw
    L = nbx.load()
    assert L.nbx_create(None, 10, 32, None) == nbx.NBX_ERR_ARG
    assert L.nbx_upload(None, *([None] * 7)) == nbx.NBX_ERR_ARG
    assert L.nbx_step(None, 0.1, 1, None) == nbx.NBX_ERR_ARG
    assert L.nbx_stats(None, None) == nbx.NBX_ERR_ARG
    L.nbx_destroy(None)  # NULL-safe
    L.nbx_group_destroy(None)
    assert L.nbx_group_create(None, 10, 32, 2, None, None) == nbx.NBX_ERR_ARG
    assert L.nbx_group_step(None, 0.1, 1, None) == nbx.NBX_ERR_ARG
    for args in ((0, 32, 2), (100, 32, 0), (100, 32, 65), (100, 8, 2)):
        with pytest.raises(nbx.NbxError) as e:
            nbx.Group(args[0], args[1], n_ranks=args[2])
        assert e.value.code == nbx.NBX_ERR_ARG, args
    assert b"ctx is NULL" in L.nbx_last_error() or L.nbx_last_error() != b""


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly_no_cpu_fallback(nbx):
    with pytest.raises(nbx.NbxError) as e:
        nbx.Context(1000, 32)
    assert e.value.code == nbx.NBX_ERR_DEVICE
    assert "no HIP device" in str(e.value)


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_cli_without_gpu_prints_header_then_fails():
    exe = os.path.join(PKG, "host", "nbody.x")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", ROOT, "host"])
    p = subprocess.run([exe, "100", "10"], capture_output=True, text=True)
    assert p.returncode != 0
    assert "no HIP device" in p.stderr
    lines = p.stdout.splitlines()
    assert lines[0] == "=" * 31 and lines[1] == " Initialize Gravity Simulation"
    assert lines[2] == " nPart = 100; nSteps = 10; dt = 0.1"
    assert lines[3] == "-" * 48
    assert lines[4] == " " + "s".ljust(8) + "dt".ljust(8) + "kenergy".ljust(12) + "time (s)".ljust(12) + "GFlops".ljust(12)


ARR = ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "mass")


@pytest.mark.parametrize("name", ["ver7_f32_n5_s20.json", "ver7_f32_n2000_s500.json", "ver7_f32_n4099_s40.json",
                                  "ver7_f32_n16384_s500.json", "ver7_f32_n65536_s20.json", "ver7_f64_n2000_s500.json"])
def test_library_initial_conditions_bit_exact_vs_reference(nbx, name):
    g = load_golden(name)
    s = nbx.initial_conditions(g["n"], g["precision"])
    for f in ARR:
        assert "%08x" % zlib.crc32(s[f].tobytes()) == g["init"][f]["crc32"], f
        assert float(s[f].astype(np.float64).sum()) == g["init"][f]["sum"], f


def test_library_initial_conditions_equal_oracle(nbx, oracle):
    for n in (1, 7, 624, 625, 1048576):  # 624*k draws cross the MT state refill
        s = nbx.initial_conditions(n, 32)
        o = oracle.init_state(n)
        for f in ARR:
            assert np.array_equal(s[f], getattr(o, f)), (n, f)


def test_large_n_checksums_from_survey(nbx):
    # SURVEY.md App. B G0 (captured from the reference by the survey): n = 262144 and 1048576
    s = nbx.initial_conditions(262144, 32)
    assert "%08x" % zlib.crc32(s["pos_x"].tobytes()) == "bc09b2f3"
    assert "%08x" % zlib.crc32(s["mass"].tobytes()) == "b5e33d2d"
    s = nbx.initial_conditions(1048576, 32)
    assert "%08x" % zlib.crc32(s["vel_z"].tobytes()) == "3e180fe8"
    assert "%08x" % zlib.crc32(s["mass"].tobytes()) == "380ad203"


def test_ic_rejects_bad_arguments(nbx):
    L = nbx.load()
    a = np.zeros(4, dtype=np.float32)
    p = a.ctypes.data_as(ctypes.c_void_p)
    assert L.nbx_ic_pos(4, 16, p, p, p) == nbx.NBX_ERR_ARG
    assert L.nbx_ic_pos(-1, 32, p, p, p) == nbx.NBX_ERR_ARG
    assert L.nbx_ic_mass(4, 32, None) == nbx.NBX_ERR_ARG
    assert L.nbx_ic_pos(0, 32, p, p, p) == nbx.NBX_OK


def test_every_entry_point_rejects_null_and_nonsense_without_aborting(nbx):
    """each symbol of include/nbx.h called with NULL handles / NULL outputs / out-of-range scalars
    returns a negative status (never aborts, never throws across the C boundary) and leaves a message."""
    L = nbx.load()
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    null = vp()
    h = vp()
    st = nbx.Stats()
    d = ctypes.c_double()
    sz = ctypes.c_size_t()
    buf = (ctypes.c_char * 128)()
    calls = {
        "nbx_create": [lambda: L.nbx_create(None, 10, 32, None), lambda: L.nbx_create(ctypes.byref(h), 0, 32, None),
                       lambda: L.nbx_create(ctypes.byref(h), -5, 32, None), lambda: L.nbx_create(ctypes.byref(h), 10, 16, None)],
        "nbx_upload": [lambda: L.nbx_upload(null, *([null] * 7))],
        "nbx_step": [lambda: L.nbx_step(null, 0.1, 1, None)],
        "nbx_step_trace": [lambda: L.nbx_step_trace(null, 0.1, 1, ctypes.byref(d)), lambda: L.nbx_step_trace(null, 0.1, 1, None)],
        "nbx_step_local": [lambda: L.nbx_step_local(null, 0.1)],
        "nbx_exchange_buffer": [lambda: L.nbx_exchange_buffer(null, ctypes.byref(h), ctypes.byref(sz), ctypes.byref(sz), ctypes.byref(sz))],
        "nbx_commit": [lambda: L.nbx_commit(null)],
        "nbx_kenergy_partial": [lambda: L.nbx_kenergy_partial(null, ctypes.byref(d))],
        "nbx_accel": [lambda: L.nbx_accel(null, null, null, null)],
        "nbx_sync": [lambda: L.nbx_sync(null)],
        "nbx_download": [lambda: L.nbx_download(null, *([null] * 6))],
        "nbx_ic_pos": [lambda: L.nbx_ic_pos(4, 32, null, null, null), lambda: L.nbx_ic_pos(4, 7, buf, buf, buf)],
        "nbx_ic_vel": [lambda: L.nbx_ic_vel(4, 32, null, null, null)],
        "nbx_ic_mass": [lambda: L.nbx_ic_mass(4, 32, null), lambda: L.nbx_ic_mass(-1, 32, buf)],
        "nbx_profile": [lambda: L.nbx_profile(null, 1)],
        "nbx_stats": [lambda: L.nbx_stats(null, ctypes.byref(st))],
        "nbx_group_create": [lambda: L.nbx_group_create(None, 10, 32, 2, None, None), lambda: L.nbx_group_create(ctypes.byref(h), 10, 32, 0, None, None),
                             lambda: L.nbx_group_create(ctypes.byref(h), 10, 32, 65, None, None), lambda: L.nbx_group_create(ctypes.byref(h), 0, 32, 2, None, None)],
        "nbx_group_upload": [lambda: L.nbx_group_upload(null, *([null] * 7))],
        "nbx_group_step": [lambda: L.nbx_group_step(null, 0.1, 1, None)],
        "nbx_group_download": [lambda: L.nbx_group_download(null, *([null] * 6))],
        "nbx_group_info": [lambda: L.nbx_group_info(null, None, None, 0, None)],
        "nbx_partition": [lambda: L.nbx_partition(0, 2, 0, *([None] * 5)), lambda: L.nbx_partition(10, 2, 2, *([None] * 5)),
                          lambda: L.nbx_partition(10, 0, 0, *([None] * 5))],
        "nbx_comm_unique_id": [lambda: L.nbx_comm_unique_id(null)],
        "nbx_group_create_rank": [lambda: L.nbx_group_create_rank(None, 10, 32, 1, 0, buf, -1, None),
                                  lambda: L.nbx_group_create_rank(ctypes.byref(h), 10, 32, 2, 2, buf, -1, None),
                                  lambda: L.nbx_group_create_rank(ctypes.byref(h), 10, 32, 1, 0, null, -1, None),
                                  lambda: L.nbx_group_create_rank(ctypes.byref(h), 300, 32, 3, 0, buf, -1, None)],  # rank 2 would be empty
        "nbx_collective_timeout": [lambda: L.nbx_collective_timeout(float("nan"))],
        "nbx_partition_weighted": [lambda: L.nbx_partition_weighted(0, 2, None, 0, *([None] * 4)), lambda: L.nbx_partition_weighted(10, 2, None, 2, *([None] * 4)),
                                   lambda: L.nbx_partition_weighted(1000, 2, (ctypes.c_double * 2)(1.0, -1.0), 0, *([None] * 4))],
        "nbx_group_create_weighted": [lambda: L.nbx_group_create_weighted(None, 10, 32, 2, None, None, None),
                                      lambda: L.nbx_group_create_weighted(ctypes.byref(h), 10, 32, 0, None, None, None),
                                      lambda: L.nbx_group_create_weighted(ctypes.byref(h), 0, 32, 2, None, None, None)],
        "nbx_group_shares": [lambda: L.nbx_group_shares(null, None, None, None)],
        "nbx_tune_weights": [lambda: L.nbx_tune_weights(0, None, None, None), lambda: L.nbx_tune_weights(2, None, None, None),
                             lambda: L.nbx_tune_weights(2, (i32 * 2)(5, 5), (ctypes.c_double * 2)(1.0, 0.0), (ctypes.c_double * 2)())],
        "nbx_group_retune": [lambda: L.nbx_group_retune(null, None, None)],
    }
    # the remaining symbols cannot fail: they are exercised for "does not crash on NULL"
    L.nbx_destroy(null)
    L.nbx_group_destroy(null)
    assert L.nbx_abi_version() == 1 and isinstance(L.nbx_last_error(), bytes)
    assert set(calls) | {"nbx_destroy", "nbx_group_destroy", "nbx_abi_version", "nbx_last_error"} == set(nbx.SYMBOLS)
    for name, fs in calls.items():
        for k, f in enumerate(fs):
            rc = f()
            assert rc < 0, (name, k, rc)
            assert L.nbx_last_error(), (name, k)
    assert not h.value  # no handle was ever produced


def test_opts_and_stats_layout_match_what_a_c_compiler_sees(nbx, tmp_path):
    """The ctypes mirrors in nbx.py against the header itself: sizes and the offsets of the newest fields."""
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "nbx.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", '
                   'sizeof(nbx_opts), sizeof(nbx_stats_t), offsetof(nbx_opts, inner_loop), offsetof(nbx_opts, summation_order), '
                   'offsetof(nbx_stats_t, inner_loop), offsetof(nbx_stats_t, graph_replays)); return 0; }\n')
    exe = str(tmp_path / "layout.x")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    got = [int(x) for x in subprocess.check_output([exe], text=True).split()]
    assert got == [ctypes.sizeof(nbx.Opts), ctypes.sizeof(nbx.Stats), nbx.Opts.inner_loop.offset, nbx.Opts.summation_order.offset,
                   nbx.Stats.inner_loop.offset, nbx.Stats.graph_replays.offset]
    assert got[0] == 72  # documented in INTEGRATION.md; reserved[] shrinks when a field is added, the size does not move
