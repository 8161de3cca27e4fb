"""GPU parity of what round 4 changed on the hot path (run with `pytest -m gpu` on a MI355X).  The kernels themselves
(fused last line of the mixed addition, G2 accumulation on lane pairs, lazily reduced FFT butterflies, one-level FFT
tables) sit under every multiexp / FFT test of the suite; here are the cases that need a special set-up:

  * the FFT's two-level tables (what sizes above 2^24 and a failed table allocation use) at sizes that take the
    one-level tables by default (src/domain.rs:81-125).
Integer work: every limb equal, no tolerances."""

import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.test_gpu_groth16 import worker  # noqa: E402,F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SNIPPET = r"""
import ctypes, sys
import numpy as np
sys.path.insert(0, %r)
import bellman_amd
from bellman_amd import _lib
from oracle import cref
lib = _lib.load()
w = bellman_amd.Worker(0)
for log_n in (12, 13, 17, 22):
    n = 1 << log_n
    data = cref.random_fr(n, 4400 + log_n)
    for mode in (0, 1, 2, 3):
        d = bellman_amd.EvaluationDomain.from_coeffs(w, data)
        [d.fft, d.ifft, d.coset_fft, d.icoset_fft][mode]()
        assert np.array_equal(d.into_coeffs(), cref.fft(data, mode)), (log_n, mode)
print("ok", flush=True)
"""


@pytest.mark.parametrize("one_level", ["0", "1"])
def test_fft_with_two_level_and_one_level_tables(one_level):
    """BELLMAN_HIP_FFT_ONE_LEVEL=0 forces the hi x lo tables (two products per twiddle / coset factor, the 1/n as a
    separate product) at sizes that by default read one-level tables in element order (one product, 1/n folded into the
    inverse twiddles): all four transforms == the restated best_fft at 2^12, 2^13, 2^17, 2^22, either way.  (The switch is
    read once per process, hence the subprocess.)"""
    env = dict(os.environ, BELLMAN_HIP_FFT_ONE_LEVEL=one_level)
    r = subprocess.run([sys.executable, "-c", _SNIPPET % ROOT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_large_host_scalar_multiexp_error_semantics(worker):
    """bh_msm_async with 2^19 + 777 HOST scalars (what a Rust host with only multiexp.rs patched issues per multiexp):
    result and error semantics of src/multiexp.rs:210-332 at a size where the full pipeline runs - == the oracle with a
    density map and skip; running out of bases near the end, or a quarter of the way in; an identity base early or late
    with a full-size scalar (top window: wins over a later EOF) or a small one (the EOF wins), or under a zero scalar
    (never seen).  (Written for the round-4 experiment that issued such a multiexp as two halves so that the second
    half's upload would overlap the first half's kernels: no gain - each half pays the latency-bound stages and a host
    tail of its own, profiles/r4_call8.txt - removed; the cases stay.)"""
    import bellman_amd
    from bellman_amd import UnexpectedEof, UnexpectedIdentity
    from oracle import cref

    n = (1 << 19) + 777
    rnd = np.random.default_rng(41)
    bases = cref.gen_bases(1, n + 100, a=3, b=5)
    sc = cref.random_fr(n, 4100)
    dens = rnd.random(n) < 0.6
    dm = bellman_amd.DensityTracker(dens)
    hb = bellman_amd.Bases(worker, 1, bases)

    def run(hbases, scalars, density, skip, host_bases):
        rc, want = cref.multiexp(1, host_bases, skip, None if density is None else cref.density_bitmap(density), scalars,
                                 threads=cref.lib().orc_max_threads())
        d = bellman_amd.FullDensity() if density is None else bellman_amd.DensityTracker(density)
        try:
            got = bellman_amd.multiexp(worker, hbases, d, scalars, skip=skip).wait()
            assert rc == 0 and np.array_equal(got, want)
            out = 0
        except UnexpectedIdentity:
            out = 1
        except UnexpectedEof:
            out = 2
        assert out == rc, (out, rc)
        return rc

    assert run(hb, sc, None, 0, bases) == 0
    assert run(hb, sc, dens, 37, bases) == 0
    del dm
    short = bases[: n - 5]                                   # the upper half runs out of bases
    hs = bellman_amd.Bases(worker, 1, short)
    assert run(hs, sc, None, 0, short) == 2
    tiny = bases[: n // 4]                                   # ... the lower half already does
    ht = bellman_amd.Bases(worker, 1, tiny)
    assert run(ht, sc, None, 0, tiny) == 2
    for pos in (1000, n - 2000):                             # an identity in the lower / the upper half
        b2 = bases.copy()
        b2[pos] = 0
        s2 = sc.copy()
        s2[pos] = cref.ints_to_arr([cref.Q - 1], 4)[0]       # non-zero top window
        h2 = bellman_amd.Bases(worker, 1, b2)
        assert run(h2, s2, None, 0, b2) == 1
        h2s = bellman_amd.Bases(worker, 1, b2[: n - 5])
        assert run(h2s, s2, None, 0, b2[: n - 5]) == 1       # ... wins over the EOF that comes later
        s2[pos] = cref.ints_to_arr([5], 4)[0]                # small scalar: zero top digit, the EOF wins
        assert run(h2s, s2, None, 0, b2[: n - 5]) == 2
        s2[pos] = 0                                          # zero scalar: the identity is skipped unseen
        assert run(h2, s2, None, 0, b2) == 0
        h2.release()
        h2s.release()
    for h in (hb, hs, ht):
        h.release()
