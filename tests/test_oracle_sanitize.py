"""The CPU oracle's C restatement (oracle/mws.c — what pins the GPU mutex watershed) under AddressSanitizer + UBSan: built here with
gcc -fsanitize=address,undefined together with tests/mws_selftest.c and run on random volumes / edge lists (sanitizers run on the CPU
build only; the GPU pool has no ASan)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not found")
def test_mws_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "mws_selftest")
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                           os.path.join(ROOT, "tests", "mws_selftest.c"), os.path.join(ROOT, "oracle", "mws.c")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1"})
    assert r.returncode == 0 and "selftest ok" in r.stdout, (r.returncode, r.stdout[-400:], r.stderr[-2000:])
