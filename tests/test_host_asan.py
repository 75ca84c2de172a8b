"""CPU sanitizer runs (SURVEY 5 "race detection / sanitizers": ASAN on the CPU restatement; GPU ASAN is not available on
this pool): the device-free host code of libseigen_hip - reference elements, mesh tables, MFMA fragment tables, the
device-free C-ABI entry points (seigen_amd/csrc: refelem.cpp, mesh_tables.cpp, mfma_tables.cpp, hostapi.cpp) - and the
oracle's C port (oracle/c/seigen_oracle.c), both built with -fsanitize=address,undefined and run on the CPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAD = ("AddressSanitizer", "runtime error:", "LeakSanitizer")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None or shutil.which("gcc") is None, reason="needs gcc / g++")


def _instrumented(path):
    out = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True).stdout
    return "__asan_init" in out and "__ubsan_handle" in out


def test_host_code_clean_under_asan_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "seigen_amd", "csrc"), "host-asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = os.path.join(ROOT, "build_tools", "host_asan_driver")
    assert _instrumented(exe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host_asan_driver: clean" in r.stdout
    assert not any(b in r.stdout + r.stderr for b in BAD), r.stderr


def test_oracle_c_port_clean_under_asan_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle", "c"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lib = os.path.join(ROOT, "oracle", "c", "libseigen_oracle_asan.so")
    assert _instrumented(lib)
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.isabs(asan_rt) and os.path.exists(asan_rt), asan_rt
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0", SEIGEN_ORACLE_LIB=lib, OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "oracle_asan_check.py")], capture_output=True, text=True,
                       env=env, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "oracle_asan_check: clean" in r.stdout
    assert not any(b in r.stdout + r.stderr for b in BAD), r.stderr
