"""The plain-C caller of the boundary on a MI355X (run with `pytest -m gpu`): tests/c/abi_smoke.c built with gcc -std=c11
against include/bellman_hip.h, `gpu` mode - a 4096-term G1 multiexp through bh_msm_async / bh_msm_wait_stats against
[sum s_i t_i]G, the EOF code of a short base vector (src/multiexp.rs:55-61), a coset FFT round trip and the degree check
(src/domain.rs:57-59)."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_plain_c_caller_on_the_gpu(tmp_path):
    from bellman_amd import _lib

    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe, "-L", libdir, "-lbellman_hip",
                    "-Wl,-rpath," + libdir], check=True)
    r = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "host + gpu checks passed" in r.stdout, r.stdout + r.stderr
