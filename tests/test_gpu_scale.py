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
from tests import golden_cache  # noqa: E402


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


def _device_bases(worker, group, t, table=False):
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
    bases = bellman_amd.Bases.wrap_device(worker, group, dout, n)   # a live view: no window table unless asked for
    if table:
        bases.precompute()
    host = bases.download()
    for i in (0, 1, n // 2, n - 1):   # spot-check the generator kernel against the oracle
        assert np.array_equal(host[i], cref.point_mul(group, gen, cref.limbs_to_int(t[i])))
    return bases, host, gen


@pytest.mark.parametrize("group,log_n,table", [(1, 19, True), (1, 21, True), (1, 23, False), (1, 26, False), (2, 20, False), (2, 22, True)])
def test_msm_c5_scale_matches_oracle(worker, group, log_n, table):
    """multiexp over 2^log_n terms == restated multiexp_inner on all host cores, and == [sum s_i t_i]G.  G2 2^20 runs the
    classic 16-window plan (one lane per point, 2^19 buckets), G2 2^22 the window-table plan a registered CRS query gets;
    G1 2^19 and 2^21 the 20-bit window table kept at a 128-byte record stride (api.hip bh_bases::table_padded) that G1
    queries of 2^19 ... 2^22 points get (automatically since round 6)."""
    import bellman_amd

    n = 1 << log_n
    t = _splitmix(n, 1000 + log_n)
    bases, host_bases, gen = _device_bases(worker, group, t, table)
    assert (bases.table_info()[1] > 0) == table
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
    # (2) the restated reference algorithm on the same inputs (tests/golden_cache.py: its answer for these seeded inputs is
    # stored; BELLMAN_GOLDEN_REGEN=1 re-runs the oracle - about a minute at 2^26 on the 128-thread box)
    def compute():
        rc, want = cref.multiexp(group, host_bases, 0, None, sc, threads=cref.lib().orc_max_threads())
        assert rc == 0
        return [want]

    t0 = time.time()
    (want,), src = golden_cache.oracle_answer("msm_c5:g%d:2^%d" % (group, log_n), [host_bases, sc], compute)
    cpu_s = time.time() - t0
    assert np.array_equal(got, want)
    print("G%d MSM 2^%d: device %.1f ms (wall %.2f s), oracle answer (%s) on %d threads %.1f s" %
          (group, log_n, ms[0], gpu_s, src, cref.lib().orc_max_threads(), cpu_s))
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
    def compute():
        rc, want = cref.multiexp(1, host_bases, skip, cref.density_bitmap(bits), sc, threads=cref.lib().orc_max_threads())
        assert rc == 0
        return [want]

    (want,), _ = golden_cache.oracle_answer("msm_2^23:density0.5:skip3", [host_bases, sc, bits.astype(np.uint8)], compute)
    assert np.array_equal(got, want)
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
    # the restated prover's answer for these seeded inputs is stored (tests/golden_cache.py); the inputs that pin it: the
    # assignment as the PRODUCT's host mirror synthesises it (equal to tests.circuits.chain_assignment_fast, which the
    # oracle is fed with when it runs - test_round3_cpu.py and the 2^20 case of test_gpu_groth16.py compare the two) and the CRS
    asg = pg.demo_assignment(1, rounds, seed, [x0])

    def compute():
        f = circuits.chain_assignment_fast(rounds, seed, x0)
        assert np.array_equal(asg["aux_assignment"], cref.fr_to_mont(cref.ints_to_arr(f["aux_assignment"], 4)))
        tc = {}
        want = cprover.prove_assignment(f["a"], f["b"], f["c"], f["input_assignment"], f["aux_assignment"], f["a_aux_density"],
                                        f["b_input_density"], f["b_aux_density"], vk, h, l, a, b1, b2, r, s,
                                        threads=cref.lib().orc_max_threads(), concurrent=True, timing=tc)
        print("oracle proof %.1f s" % tc["total_s"])
        return list(want)

    want, src = golden_cache.oracle_answer("proof_chain:2^22:seed4242", [asg["aux_assignment"], asg["a"], h, l, a, b1, b2,
                                                                         np.array([r, s], dtype=np.uint64)], compute)
    print("2^22 proof: device host-ms [witness, h, msm, total] = %s; oracle answer: %s" % ([round(x, 1) for x in tm], src))
    assert got.a.tobytes() == want[0].tobytes()
    assert got.b.tobytes() == want[1].tobytes()
    assert got.c.tobytes() == want[2].tobytes()
    pp.release()


def test_proof_2_24_config_c5(worker):
    """BASELINE configs[4], proof leg: create_proof at 2^24 constraints on one GPU (CRS from the device generator, R1CS
    resident).  (a) the proof satisfies the verification equation (oracle/pyref/pairing.py restates verify_proof,
    groth16/src/verifier.rs:23-58) for the right public input only; (b) the proof assembled from 2 and from 8 parts
    (prove_witness_part + sums_add + assemble: what 2 / 8 ranks compute) is the same proof; (c) its largest G2 multiexp -
    b_g2 over 2^23 points, the first size without a window table: the classic 16-window plan with one lane per point in
    accumulation AND reduction - equals the restated multiexp (src/multiexp.rs:210-332) on all host cores.
    The full comparison with the restated prover is tools/check_proof_large.py 24 (profiles/archive/r3_proof_2p24.txt)."""
    import bellman_amd
    from bellman_amd import groth16 as pg
    from oracle.pyref import pairing
    from tests.test_gpu_generator import _recs
    from oracle.cengine import CBls12

    log_n = 24
    rounds = (1 << log_n) - 3
    seed, x0, r, s = 2424, 31337, 0xABCDEF0123456789ABCDEF, 0x1234567890ABCDEF123
    t0 = time.time()
    r1cs = pg.R1CS.from_demo(worker, 1, rounds, seed)
    g1, g2 = _recs(1, [CBls12.G1.gen])[0], _recs(2, [CBls12.G2.gen])[0]
    params = pg.Parameters.generate(worker, r1cs, g1, g2, 48577, 22580, 53332, 5481, 3673)
    t_setup = time.time() - t0
    tm = [0, 0, 0, 0]
    t0 = time.time()
    proof = pg.create_proof_demo_r1cs(params, r1cs, 1, rounds, seed, [x0], None, r, s, tm)
    t_proof = time.time() - t0
    print("2^24 proof: setup %.1f s (capture + generate), proof %.0f ms wall, host ms [witness, h, msm, total] = %s"
          % (t_setup, t_proof * 1e3, [round(x, 1) for x in tm]))
    # (a) verification equation
    asg = pg.demo_assignment(1, rounds, seed, [x0])
    image = cref.arr_to_ints(cref.fr_from_mont(asg["input_assignment"][1:2]))[0]
    alpha_g1, _, beta_g2, _, delta_g2 = params.vk()
    gamma_g2, ic = params.vk_ext()
    vk = dict(alpha_g1=cref.g1_to_py(alpha_g1)[0], beta_g2=cref.g2_to_py(beta_g2)[0], gamma_g2=cref.g2_to_py(gamma_g2)[0],
              delta_g2=cref.g2_to_py(delta_g2)[0], ic=cref.g1_to_py(ic))
    pr = (cref.g1_to_py(proof.a)[0], cref.g2_to_py(proof.b)[0], cref.g1_to_py(proof.c)[0])
    assert pairing.verify_proof(vk, pr, [image])
    assert not pairing.verify_proof(vk, pr, [(image + 1) % cref.Q])
    # (b) parts
    for parts in (2, 8):
        total = None
        for part in range(parts):
            sums = pg.prove_demo_part(params, r1cs, 1, rounds, seed, [x0], None, part, parts)
            total = sums if total is None else pg.sums_add(total, sums)
        folded = pg.assemble(params, total, r, s)
        assert folded.a.tobytes() == proof.a.tobytes() and folded.b.tobytes() == proof.b.tobytes() and \
            folded.c.tobytes() == proof.c.tobytes(), parts
    # (c) the b_g2 multiexp of this proof against the oracle
    hb = params.bases("b_g2")
    assert len(hb) > (1 << 22) and hb.table_info()[1] == 0          # no window table: the classic plan
    n_aux = asg["aux_assignment"].shape[0]
    bits = np.unpackbits(asg["b_aux_density"].view(np.uint8), bitorder="little")[:n_aux].astype(bool)
    b_in_total = int(np.unpackbits(asg["b_input_density"].view(np.uint8), bitorder="little")[:2].sum())
    dt = bellman_amd.DensityTracker()
    dt.bv = bits
    got, ms = bellman_amd.multiexp(worker, hb, dt, asg["aux_assignment"], skip=b_in_total, mont=True, timed=True).wait()
    b2_host = params.query("b_g2")

    def compute():
        rc, want = cref.multiexp(2, b2_host, b_in_total, cref.density_bitmap(bits), cref.fr_from_mont(asg["aux_assignment"]),
                                 threads=cref.lib().orc_max_threads())
        assert rc == 0
        return [want]

    t0 = time.time()
    (want,), src = golden_cache.oracle_answer("msm_b_g2:proof2^24:seed2424", [b2_host, asg["aux_assignment"], bits.astype(np.uint8)], compute)
    print("b_g2 multiexp (2^23 points, classic plan): device %.1f ms [sort %.1f, accumulate %.1f, reduce %.1f]; oracle answer (%s) %.1f s"
          % (ms[0], ms[1], ms[2], ms[3], src, time.time() - t0))
    assert np.array_equal(got, want)
    r1cs.release()
    params.release()
    worker.trim()


@pytest.mark.parametrize("log_n", [26, 28])
def test_fft_above_2_25(worker, log_n):
    """EvaluationDomain at 2^26 ... 2^28 (src/domain.rs:57-59 allows up to 2^31; three passes, 2 - 8 GiB vectors; the
    restated best_fft is too slow here, so the checks are size-independent):
      * a SPARSE polynomial with coefficients at low, middle and top indices: every sampled output equals
        sum_j a_j w^(i_j k) computed with Python integers - wrong index arithmetic above 2^25 (32-bit products, the
        three-pass digit reversal) cannot survive this;
      * coset_fft of the same polynomial at sampled points (a_j 7^(i_j) w^(i_j k));
      * random data: ifft(fft(x)) == x and icoset_fft(coset_fft(x)) == x on every element.
    2^29 ... 2^31 (16 - 64 GiB vectors + as much scratch + a copy): the same method with everything generated and
    compared on the device side, tools/fft_huge.py - test_fft_2_29_device_side below runs 2^29, the output of all three
    sizes is profiles/archive/r4_fft_2p29_2p31.txt."""
    import bellman_amd

    n = 1 << log_n
    q = cref.Q
    rnd = np.random.default_rng(log_n)
    omega = pow(pow(7, (q - 1) >> 32, q), 1 << (32 - log_n), q)
    pos = [0, 1, 2047, 2048, (1 << 22) + 5, n // 2 - 1, n // 2, n - 2049, n - 1] + [int(x) for x in rnd.integers(0, n, 7)]
    pos = sorted(set(pos))
    coef = [int(x) for x in rnd.integers(1, 1 << 62, len(pos))]
    data = np.zeros((n, 4), dtype=np.uint64)
    data[pos] = cref.fr_to_mont(cref.ints_to_arr(coef, 4))
    ks = [0, 1, n - 1, n // 2, (1 << 25) + 3] + [int(x) for x in rnd.integers(0, n, 40)]
    for mode, shift in (("fft", 1), ("coset_fft", 7)):
        d = bellman_amd.EvaluationDomain.from_coeffs(worker, data)
        getattr(d, mode)()
        out = d.into_coeffs()
        got = cref.arr_to_ints(cref.fr_from_mont(out[ks]))
        for k, g in zip(ks, got):
            want = sum(c * pow(shift, i, q) % q * pow(omega, (i * k) % n, q) for c, i in zip(coef, pos)) % q
            assert g == want, (log_n, mode, k)
        del out
    del data
    x = _splitmix(n, 7000 + log_n)
    d = bellman_amd.EvaluationDomain.from_coeffs(worker, x)
    d.fft()
    d.ifft()
    assert np.array_equal(d.as_ref(), x)
    d.coset_fft()
    d.icoset_fft()
    assert np.array_equal(d.into_coeffs(), x)
    worker.trim()


def test_fft_2_29_device_side():
    """[r4] 2^29 points (16 GiB vector; three passes 10 + 10 + 9) by tools/fft_huge.py: sparse polynomial against Python
    integers at sampled outputs (fft, coset_fft), dense round trips compared on the device - the method of
    test_fft_above_2_25 without host-sized arrays.  2^30 and 2^31 (192 GiB of device memory in total) were run with the
    same tool: profiles/archive/r4_fft_2p29_2p31.txt (src/domain.rs:57-59 allows exp <= 31)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fft_huge.py"), "29"], cwd=root, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "2^29 done" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
