"""AddressSanitizer + UBSan over the product's host-only code (CPU build; GPU ASan is unavailable on this pool)."""
import os
import subprocess

from conftest import ROOT


def test_host_code_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "san_driver")
    subprocess.check_call(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-ffp-contract=off", os.path.join(ROOT, "tests", "san_driver.cpp"),
                           os.path.join(ROOT, "nbody-demo-2023_amd", "csrc", "nbx_ic.cpp"), "-o", exe])
    p = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert p.returncode == 0, p.stdout + p.stderr
    assert "sanitized host code: ok" in p.stdout
