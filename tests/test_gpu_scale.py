"""GPU parity at BASELINE config C5 scale (run with `pytest -m gpu` on a MI355X).

The everyday parity suite (test_gpu_parity.py) stops at 2^20 terms / 2^22 points; this file covers the
sizes the multi-GPU configuration hands to one rank and above, where the 32-bit position arithmetic of
the sort and the multi-GiB workspaces are stressed:

  * G1 MSM 2^23 (the per-rank shard of C5's 2^26-base MSM on 8 GPUs) and 2^26 (the whole of it) against
    the restated multiexp (src/multiexp.rs:210-332) on all host cores, plus the size-independent identity
    sum_i s_i [t_i]G = [sum_i s_i t_i mod q]G (the bases are generated as known multiples of G);
  * G2 MSM 2^20 and 2^22 likewise;
  * FFT / iFFT / coset variants for EVERY domain size 2^0 ... 2^25 not already in test_gpu_parity.py
    (src/domain.rs:81-125; every pass-plan of csrc/fft.hip is exercised on hardware);
  * a 2^22-constraint create_proof against the restated prover (groth16/src/prover.rs:217-360).

CPU oracle time on the 128-thread GPU box: about 1 min for the 2^26 MSM and for the 2^22 proof, seconds
for the rest.  Integer work: exact equality of every limb, no tolerances."""

import ctypes
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import cprover, cref  # noqa: E402


@pytest.fixture(scope="module")
def worker():
    import bellman_amd

    w = bellman_amd.Worker(0)
    yield w
    w.close()


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _splitmix(n, seed):
    from bench import splitmix_scalars

    return splitmix_scalars(n, seed)


def _device_bases(worker, group, t):
    """P_i = [t_i]G made on the device (bh_fixed_base_mul_dev; itself checked against the oracle's
    point_mul below and in test_gpu_parity.py) -> (Bases handle, host copy of the records)."""
    import bellman_amd
    from bellman_amd import _lib

    lib = _lib.load()
    n = t.shape[0]
    words = 12 if group == 1 else 24
    gen = cref.g1_generator() if group == 1 else cref.g2_generator()
    dt, dout = worker.alloc(n * 32), worker.alloc(n * 8 * words)
    worker.upload(dt, t)
    assert lib.bh_fixed_base_mul_dev(worker.ctx, group, _p(gen), dt, n, 0, dout, None) == 0
    worker.synchronize()
    worker.free(dt)
    bases = bellman_amd.Bases.wrap_device(worker, group, dout, n)
    host = bases.download()
    for i in (0, 1, n // 2, n - 1):   # spot-check the generator kernel against the oracle
        assert np.array_equal(host[i], cref.point_mul(group, gen, cref.limbs_to_int(t[i])))
    return bases, host, gen


@pytest.mark.parametrize("group,log_n", [(1, 23), (1, 26), (2, 20), (2, 22)])
def test_msm_c5_scale_matches_oracle(worker, group, log_n):
    """multiexp over 2^log_n terms == restated multiexp_inner on all host cores, and == [sum s_i t_i]G."""
    import bellman_amd

    n = 1 << log_n
    t = _splitmix(n, 1000 + log_n)
    bases, host_bases, gen = _device_bases(worker, group, t)
    sc = _splitmix(n, 2000 + log_n)
    sc[1] = 0
    sc[2] = cref.ints_to_arr([1], 4)[0]
    sc[3] = cref.ints_to_arr([cref.Q - 1], 4)[0]
    sc[n - 1] = sc[n - 2]
    ds = worker.alloc(n * 32)
    worker.upload(ds, sc)
    t0 = time.time()
    got, ms = bellman_amd.multiexp(worker, bases, bellman_amd.FullDensity(), None, scalars_dev=ds, n=n, timed=True).wait()
    gpu_s = time.time() - t0
    worker.free(ds)
    # (1) size-independent identity
    k = cref.fr_dot(sc, t)
    assert np.array_equal(got, cref.point_mul(group, gen, k)), "MSM != [sum s_i t_i]G"
    # (2) the restated reference algorithm on the same inputs
    t0 = time.time()
    rc, want = cref.multiexp(group, host_bases, 0, None, sc, threads=cref.lib().orc_max_threads())
    cpu_s = time.time() - t0
    assert rc == 0
    assert np.array_equal(got, want)
    print("G%d MSM 2^%d: device %.1f ms (wall %.2f s), oracle on %d threads %.1f s" %
          (group, log_n, ms[0], gpu_s, cref.lib().orc_max_threads(), cpu_s))
    bases.release()


def test_msm_2_23_density_and_skip(worker):
    """the same shard size with a DensityTracker and a base offset (groth16/src/lib.rs:451-473): positions
    of a density-compacted query at C5 scale."""
    import bellman_amd

    n = 1 << 23
    rnd = np.random.default_rng(5)
    bits = rnd.random(n) < 0.5
    skip = 3
    nb = int(bits.sum()) + skip
    t = _splitmix(nb, 77)
    bases, host_bases, gen = _device_bases(worker, 1, t)
    sc = _splitmix(n, 78)
    dt = bellman_amd.DensityTracker()
    dt.bv = bits   # numpy bool vector (words() packs it); avoids 2^23 Python objects
    got = bellman_amd.multiexp(worker, bases, dt, sc, skip=skip).wait()
    dense_sc = sc[bits]
    k = cref.fr_dot(dense_sc, t[skip:])
    assert np.array_equal(got, cref.point_mul(1, gen, k))
    rc, want = cref.multiexp(1, host_bases, skip, cref.density_bitmap(bits), sc, threads=cref.lib().orc_max_threads())
    assert rc == 0 and np.array_equal(got, want)
    bases.release()


# sizes not covered by test_gpu_parity.py::test_fft_all_modes_bit_exact / test_fft_2_22_config_c3
@pytest.mark.parametrize("log_n", [4, 6, 7, 9, 14, 19, 21, 23, 24, 25])
def test_fft_every_remaining_size_bit_exact(worker, log_n):
    """fft / ifft / coset_fft / icoset_fft == restated best_fft at every remaining domain size up to 2^25
    (2^24 = C5's domain), all four modes, every limb; plus the round trip."""
    import bellman_amd

    n = 1 << log_n
    data = cref.random_fr(n, 300 + log_n)
    # the restated parallel_fft pays n * P extra multiplications for P = 2^floor(log2 threads) (domain.rs:340-349):
    # 16 threads keep the oracle in seconds at 2^25
    threads = 16 if log_n >= 16 else 8
    big = log_n >= 23
    # above 2^22 only fft and icoset_fft are compared with the oracle; ifft and coset_fft are then pinned as the
    # inverse maps of verified maps by the two round trips below (halves the oracle time of the 3-pass sizes)
    for mode, name in enumerate(("fft", "ifft", "coset_fft", "icoset_fft")):
        if big and name in ("ifft", "coset_fft"):
            continue
        d = bellman_amd.EvaluationDomain.from_coeffs(worker, data)
        getattr(d, name)()
        got = d.into_coeffs()
        assert np.array_equal(got, cref.fft(data, mode, threads=threads)), (log_n, name)
    d = bellman_amd.EvaluationDomain.from_coeffs(worker, data)
    d.coset_fft()
    d.icoset_fft()
    assert np.array_equal(d.as_ref(), data)
    d.fft()
    d.ifft()
    assert np.array_equal(d.into_coeffs(), data)


def test_proof_2_22_constraints_matches_oracle(worker):
    """create_proof on the synthetic chain circuit with 2^22 constraints (R1CS resident in HBM) == the
    restated prover (prover.rs:217-360) on all host cores: proof A, B, C bit-identical."""
    from bellman_amd import groth16 as pg
    from tests import circuits

    log_n = 22
    rounds = (1 << log_n) - 3
    seed, x0, r, s = 4242, 1234567, 0xABCDEF0123456789, 0x1234567890ABCDEF
    m = 1 << log_n
    n_aux, nb = rounds + 1, (rounds + 1) // 2 + 2
    h, l = cref.gen_bases(1, m - 1, a=11, b=3), cref.gen_bases(1, n_aux, a=5, b=7)
    a, b1, b2 = cref.gen_bases(1, n_aux + 2, a=2, b=9), cref.gen_bases(1, nb, a=13, b=4), cref.gen_bases(2, nb, a=17, b=6)
    g1, g2 = cref.g1_generator(), cref.g2_generator()
    vk = dict(alpha_g1=cref.point_mul(1, g1, 101), beta_g1=cref.point_mul(1, g1, 102), beta_g2=cref.point_mul(2, g2, 102),
              delta_g1=cref.point_mul(1, g1, 103), delta_g2=cref.point_mul(2, g2, 103))
    pp = pg.Parameters(worker, vk["alpha_g1"], vk["beta_g1"], vk["beta_g2"], vk["delta_g1"], vk["delta_g2"], h, l, a, b1, b2)
    r1cs = pg.R1CS.from_demo(worker, 1, rounds, seed)
    tm = [0, 0, 0, 0]
    got = pg.create_proof_demo_r1cs(pp, r1cs, 1, rounds, seed, [x0], None, r, s, tm)
    f = circuits.chain_assignment_fast(rounds, seed, x0)
    tc = {}
    want = cprover.prove_assignment(f["a"], f["b"], f["c"], f["input_assignment"], f["aux_assignment"], f["a_aux_density"],
                                    f["b_input_density"], f["b_aux_density"], vk, h, l, a, b1, b2, r, s,
                                    threads=cref.lib().orc_max_threads(), concurrent=True, timing=tc)
    print("2^22 proof: device host-ms [witness, h, msm, total] = %s; oracle %.1f s" % ([round(x, 1) for x in tm], tc["total_s"]))
    assert got.a.tobytes() == want[0].tobytes()
    assert got.b.tobytes() == want[1].tobytes()
    assert got.c.tobytes() == want[2].tobytes()
    pp.release()
