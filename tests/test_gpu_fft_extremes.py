"""Extreme input vectors through every FFT pass plan and both kinds of twiddle tables (run with `pytest -m gpu` on a
MI355X) - src/domain.rs:81-125, :261-314.

The butterflies of csrc/fft.hip compute lazily reduced in [0, 2q) on the argument "a value < 2q times a canonical table
entry leaves the Montgomery multiplier below 1.91 q without its final subtraction"; the one-level tables fold the 1/n of
the inverse transforms into their first twiddle table.  Random vectors (what every other FFT test feeds) sit in the
middle of every range, so here are the corners: all limbs at q - 1 (the largest stored value), all zero, all one, all
minus one, a single non-zero coefficient at index 0 / 1 / n - 1, alternating 0 / q - 1, and (q - 1)(-1)^i (at 2^21 and
above without the all-zero, all-one and index-0 vectors, whose oracle transforms would only add minutes) - all four
transforms, at the sizes where the pass plan and the table kind switch (2^11: one pass; 2^12, 2^13: two passes, the
smallest one-level sizes; 2^21, 2^22: two passes with tiles one or two elements wide; 2^23: three passes), each with the
one-level tables (default) and with the hi x lo tables forced (BELLMAN_HIP_FFT_ONE_LEVEL=0: what sizes above 2^24, or
a table cache over its budget, run).  Checked against the restated best_fft (oracle/c), every limb.  The switch is read
once per process, hence one subprocess per table kind; the oracle's outputs are computed once and shared through a
temporary directory."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SNIPPET = r"""
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import bellman_amd
from oracle import cref

Q = cref.Q
R = (1 << 256) %% Q
log_n, cache = %(log_n)d, %(cache)r
n = 1 << log_n

def limbs(x):
    return np.array([(x >> (64 * k)) & cref.MASK64 for k in range(4)], dtype=np.uint64)

def const(x):
    return np.tile(limbs(x), (n, 1))

def delta(i, x):
    v = np.zeros((n, 4), dtype=np.uint64)
    v[i] = limbs(x)
    return v

def alternating(even, odd):
    v = const(even)
    v[1::2] = limbs(odd)
    return v

vectors = {
    "all_stored_q_minus_1": const(Q - 1),
    "all_zero": const(0),
    "all_one": const(R),                       # the field's one (Montgomery form)
    "all_minus_one": const(Q - R),             # the field's q - 1
    "delta_0": delta(0, Q - 1),
    "delta_1": delta(1, R),
    "delta_last": delta(n - 1, 1),
    "alternating_0_qm1": alternating(0, Q - 1),
    "minus_one_alternating_sign": alternating(Q - R, R),   # (q - 1) * (-1)^i
}
# the sizes whose oracle transform takes seconds keep the vectors that sit on the range bounds (the others add nothing a
# smaller size does not already cover); their oracle outputs are computed concurrently (the C oracle releases the GIL;
# 8 threads each: its parallel_fft costs m * P extra products, so more threads per transform only slow it down)
if log_n >= 20:
    vectors = {k: vectors[k] for k in ("all_stored_q_minus_1", "all_minus_one", "delta_1", "delta_last", "alternating_0_qm1",
                                       "minus_one_alternating_sign")}
from concurrent.futures import ThreadPoolExecutor

def want_of(job):
    name, mode = job
    path = os.path.join(cache, "%%d_%%s_%%d.npy" %% (log_n, name, mode))
    if not os.path.exists(path):
        np.save(path, cref.fft(vectors[name], mode, threads=8))
    return path

cref.lib()
jobs = [(name, mode) for name in vectors for mode in (0, 1, 2, 3)]
with ThreadPoolExecutor(max_workers=max(1, min(12, (os.cpu_count() or 8) // 8))) as ex:
    paths = dict(zip(jobs, ex.map(want_of, jobs)))
w = bellman_amd.Worker(0)
for name, data in vectors.items():
    for mode in (0, 1, 2, 3):
        want = np.load(paths[(name, mode)])
        d = bellman_amd.EvaluationDomain.from_coeffs(w, data)
        [d.fft, d.ifft, d.coset_fft, d.icoset_fft][mode]()
        got = d.into_coeffs()
        assert np.array_equal(got, want), (log_n, name, mode, int((got != want).any(axis=1).sum()))
print("ok", flush=True)
"""


@pytest.fixture(scope="module")
def oracle_cache(tmp_path_factory):
    return str(tmp_path_factory.mktemp("fft_extremes"))


@pytest.mark.parametrize("log_n", [11, 12, 13, 21, 22, 23])
def test_fft_extreme_vectors(log_n, oracle_cache):
    for one_level in ("1", "0"):
        env = dict(os.environ, BELLMAN_HIP_FFT_ONE_LEVEL=one_level)
        code = _SNIPPET % {"root": ROOT, "log_n": log_n, "cache": oracle_cache}
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0 and "ok" in r.stdout, "one_level=%s\n" % one_level + r.stdout[-2000:] + r.stderr[-3000:]
