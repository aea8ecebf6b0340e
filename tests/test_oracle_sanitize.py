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


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not found")
def test_lz4_block_decoder_under_asan_ubsan(tmp_path):
    """The library's host-side LZ4 decoder (csrc/io.hip — product code on the chunk reader's path, the one piece that parses untrusted bytes) built
    alone with g++ -fsanitize=address,undefined and fuzzed: round trips, truncations, bit flips, random streams, short destination buffers."""
    exe = str(tmp_path / "lz4_fuzz")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-DRG_IO_STANDALONE", "-o", exe,
                           "-x", "c++", os.path.join(ROOT, "raygun_amd", "csrc", "io.hip"), os.path.join(ROOT, "tests", "lz4_fuzz.c")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "lz4 fuzz ok" in r.stdout, (r.returncode, r.stdout[-400:], r.stderr[-2000:])
