"""GPU parity tests (run with `pytest -m gpu` on a MI355X): every call goes through the C ABI
(libbellman_hip.so) and is compared BIT-EXACTLY with the CPU oracle (oracle/c, itself validated
against the KAT-pinned oracle/pyref).  Integer work => exact equality, no tolerances.

Covers SURVEY.md Appendix A: MSM items 1-9, FFT/domain items 10-15."""

import ctypes
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cref  # noqa: E402
from oracle.pyref import bls12_381 as bls  # noqa: E402

Q = bls.Q


@pytest.fixture(scope="module")
def worker():
    import bellman_amd

    w = bellman_amd.Worker(0)
    yield w
    w.close()


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _dev(worker, arr):
    d = worker.alloc(max(arr.nbytes, 16))
    if arr.nbytes:
        worker.upload(d, arr)
    return d


# ------------------------------------------------------------------------------ field / group
def test_field_mul_on_device(worker):
    from bellman_amd import _lib

    lib = _lib.load()
    n = 20000
    a, b = cref.random_fr(n, 11), cref.random_fr(n, 12)
    a[0], b[0] = 0, 0
    a[1] = cref.ints_to_arr([Q - 1], 4)[0]
    b[1] = a[1]
    da, db, dr = _dev(worker, a), _dev(worker, b), worker.alloc(a.nbytes)
    assert lib.bh_test_fr_mul_dev(worker.ctx, dr, da, db, n) == 0
    r = np.empty_like(a)
    worker.download(r, dr)
    assert np.array_equal(r, cref.mul_assign(a, b))
    # Fp: random values below p
    rnd = random.Random(5)
    xs = [rnd.randrange(bls.P) for _ in range(2000)] + [0, 1, bls.P - 1]
    ys = [rnd.randrange(bls.P) for _ in range(2000)] + [bls.P - 1] * 3
    xa, ya = cref.ints_to_arr(xs, 6), cref.ints_to_arr(ys, 6)
    dx, dy, dz = _dev(worker, xa), _dev(worker, ya), worker.alloc(xa.nbytes)
    assert lib.bh_test_fp_mul_dev(worker.ctx, dz, dx, dy, len(xs)) == 0
    z = np.empty_like(xa)
    worker.download(z, dz)
    rinv = pow(pow(2, 384, bls.P), -1, bls.P)
    assert cref.arr_to_ints(z) == [x * y * rinv % bls.P for x, y in zip(xs, ys)]


@pytest.mark.parametrize("group", [1, 2])
def test_point_add_on_device(worker, group):
    from bellman_amd import _lib

    lib = _lib.load()
    n = 300
    w = 12 if group == 1 else 24
    A = cref.gen_bases(group, n, a=5, b=3)
    B = cref.gen_bases(group, n, a=7, b=11)
    B[0] = A[0]
    B[1] = 0
    A[2] = 0
    B[3] = cref.point_mul(group, A[3], Q - 1)
    A[4] = 0
    B[4] = 0
    dA, dB, dR = _dev(worker, A), _dev(worker, B), worker.alloc(A.nbytes)
    assert lib.bh_test_point_add_dev(worker.ctx, group, dR, dA, dB, n) == 0
    out = np.empty_like(A)
    worker.download(out, dR)
    want = np.stack([cref.point_add(group, A[i], B[i]) for i in range(n)])
    assert np.array_equal(out, want)
    assert out.shape[1] == w and not out[3].any() and not out[4].any()


def test_g2_k3_group_law(worker):
    """The lane-triple (K3) form of Fp2 the G2 MSM kernels compute in (csrc/fp2k3.cuh): general addition, mixed
    addition and doubling of G2 points, including P + P, P + (-P) and the identity on either side, against the
    oracle's group law.  70 points: more than one wavefront of 21 triples, and a ragged tail."""
    from bellman_amd import _lib

    lib = _lib.load()
    n = 70
    A = cref.gen_bases(2, n, a=5, b=3)
    B = cref.gen_bases(2, n, a=7, b=11)
    B[0] = A[0]                                   # doubling through the addition paths
    B[1] = 0                                      # + identity
    A[2] = 0                                      # identity +
    B[3] = cref.point_mul(2, A[3], Q - 1)         # P + (-P)
    A[4] = 0
    B[4] = 0
    B[n - 1] = A[n - 1]
    dA, dB = _dev(worker, A), _dev(worker, B)
    outs = [np.zeros((n, 24), dtype=np.uint64) for _ in range(3)]
    assert lib.bh_test_g2_k3_dev(worker.ctx, _p(outs[0]), _p(outs[1]), _p(outs[2]), dA, dB, n) == 0
    want_add = np.stack([cref.point_add(2, A[i], B[i]) for i in range(n)])
    want_dbl = np.stack([cref.point_add(2, A[i], A[i]) for i in range(n)])
    assert np.array_equal(outs[0], want_add)
    assert np.array_equal(outs[1], want_add)      # mixed addition: same sums (identity B leaves A)
    assert np.array_equal(outs[2], want_dbl)
    assert not outs[0][3].any() and not outs[0][4].any()


def test_g2_lane_sextet_addition(worker):
    """[r6] The lane-sextet (K6) general addition the G2 merge kernels run (csrc/msm_ec.cuh 5'': the x / y split of a point
    over two lane triples, seven product slots instead of fourteen): a + b for G2 points including P + P (the doubling
    fallback), P + (-P) and the identity on either side, against the oracle's group law.  70 points: nine wavefronts of 8
    sextets, a ragged tail."""
    from bellman_amd import _lib

    lib = _lib.load()
    n = 70
    A = cref.gen_bases(2, n, a=19, b=3)
    B = cref.gen_bases(2, n, a=23, b=29)
    B[0] = A[0]                                   # doubling
    B[1] = 0                                      # + identity
    A[2] = 0                                      # identity +
    B[3] = cref.point_mul(2, A[3], Q - 1)         # P + (-P)
    A[4] = 0
    B[4] = 0
    B[9] = A[9]                                   # a doubling beside ordinary additions in one wavefront
    B[n - 1] = A[n - 1]
    dA, dB = _dev(worker, A), _dev(worker, B)
    out = np.zeros((n, 24), dtype=np.uint64)
    assert lib.bh_test_g2_k6_dev(worker.ctx, _p(out), dA, dB, n) == 0
    want = np.stack([cref.point_add(2, A[i], B[i]) for i in range(n)])
    assert np.array_equal(out, want)
    assert not out[3].any() and not out[4].any()


def test_g2_lane_pair_group_law(worker):
    """The lane-pair form of Fp2 the large G2 accumulations compute in (csrc/fp2pair.cuh: schoolbook products with one
    reduction per lane, operands exchanged by DPP): the same cases as the lane-triple test above.  70 points: more than
    two wavefronts of 32 pairs, and a ragged tail."""
    from bellman_amd import _lib

    lib = _lib.load()
    n = 70
    A = cref.gen_bases(2, n, a=9, b=13)
    B = cref.gen_bases(2, n, a=17, b=5)
    B[0] = A[0]                                   # doubling through the addition paths
    B[1] = 0                                      # + identity
    A[2] = 0                                      # identity +
    B[3] = cref.point_mul(2, A[3], Q - 1)         # P + (-P)
    A[4] = 0
    B[4] = 0
    B[n - 1] = A[n - 1]
    dA, dB = _dev(worker, A), _dev(worker, B)
    outs = [np.zeros((n, 24), dtype=np.uint64) for _ in range(3)]
    assert lib.bh_test_g2_pairs_dev(worker.ctx, _p(outs[0]), _p(outs[1]), _p(outs[2]), dA, dB, n) == 0
    want_add = np.stack([cref.point_add(2, A[i], B[i]) for i in range(n)])
    want_dbl = np.stack([cref.point_add(2, A[i], A[i]) for i in range(n)])
    assert np.array_equal(outs[0], want_add)
    assert np.array_equal(outs[1], want_add)      # mixed addition: same sums (identity B leaves A)
    assert np.array_equal(outs[2], want_dbl)
    assert not outs[0][3].any() and not outs[0][4].any()


@pytest.mark.parametrize("n", [1, 2, 3, 8])
@pytest.mark.parametrize("group", [1, 2])
def test_msm_tiny_host_path_matches_pipeline(worker, group, n):
    """Multiexps of at most 8 terms with host scalars are answered on the host (create_proof's `inputs`
    multiexps): same result as the kernel pipeline (BH_MSM_NO_SMALL_PATH) and as the oracle, with and
    without a density map / skip, zero and one scalars included; same error semantics."""
    import bellman_amd
    from bellman_amd import UnexpectedEof, UnexpectedIdentity
    from bellman_amd.multiexp import NO_SMALL_PATH

    bases = cref.gen_bases(group, n + 3, a=31, b=7)
    hb = bellman_amd.Bases(worker, group, bases)
    sc = cref.random_fr(n, 500 + n)
    sc[0] = cref.ints_to_arr([1], 4)[0]
    if n > 2:
        sc[1] = 0
    for skip, dens in ((0, None), (2, None), (1, np.array([True, False, True, True, False, True, True, True][:n]))):
        dm = bellman_amd.FullDensity() if dens is None else bellman_amd.DensityTracker(dens)
        rc, want = cref.multiexp(group, bases, skip, None if dens is None else cref.density_bitmap(dens), sc)
        assert rc == 0
        for mont in (False, True):
            s_in = cref.fr_to_mont(sc) if mont else sc
            got = bellman_amd.multiexp(worker, hb, dm, s_in, skip=skip, mont=mont).wait()
            ref = bellman_amd.multiexp(worker, hb, dm, s_in, skip=skip, mont=mont, flags=NO_SMALL_PATH).wait()
            assert np.array_equal(got, want) and np.array_equal(ref, want), (group, n, skip, mont)
    # errors: running out of bases, an identity under a non-zero scalar, and both (top window decides)
    with pytest.raises(UnexpectedEof):
        bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc, skip=4).wait()
    ib = bases.copy()
    ib[n - 1] = 0
    hib = bellman_amd.Bases(worker, group, ib)
    big = sc.copy()
    big[n - 1] = cref.ints_to_arr([Q - 1], 4)[0]          # non-zero top window
    with pytest.raises(UnexpectedIdentity):
        bellman_amd.multiexp(worker, hib, bellman_amd.FullDensity(), big).wait()
    if n >= 3:
        # both kinds of failure: n scalars over n - 1 bases (the last entry runs out) with an identity at n - 2.
        # The reference reports its top window's first failure (multiexp.rs:295-300): the identity if that scalar
        # has a non-zero top digit, else the EOF.
        jb = bases[: n - 1].copy()
        jb[n - 2] = 0
        hjb = bellman_amd.Bases(worker, group, jb)
        both = cref.random_fr(n, 77)
        both[n - 2] = cref.ints_to_arr([Q - 1], 4)[0]
        with pytest.raises(UnexpectedIdentity):
            bellman_amd.multiexp(worker, hjb, bellman_amd.FullDensity(), both).wait()
        both[n - 2] = cref.ints_to_arr([5], 4)[0]
        with pytest.raises(UnexpectedEof):
            bellman_amd.multiexp(worker, hjb, bellman_amd.FullDensity(), both).wait()
        for sc2 in (both,):   # and the oracle agrees on the precedence
            rc, _ = cref.multiexp(group, jb, 0, None, sc2)
            assert rc == 2


@pytest.mark.parametrize("n", [257, 5000])
def test_msm_g2_single_lane_kernels_still_agree(worker, n):
    """BH_MSM_G2_SINGLE_LANE / _LANE_PAIRS: the one-lane-per-point G2 kernels and the lane-pair accumulation (the default
    of large jobs) == the default K3 kernels of small jobs."""
    import bellman_amd
    from bellman_amd.multiexp import NO_TABLE

    bases = cref.gen_bases(2, n, a=3, b=19)
    sc = _scalars(n, 600 + n)
    hb = bellman_amd.Bases(worker, 2, bases)
    rc, want = cref.multiexp(2, bases, 0, None, sc)
    assert rc == 0
    # 16: one lane per point, 32: lane triples, 256: lane pairs accumulate (merge + reduction as the other flag says)
    for flags in (0, 16, 32, NO_TABLE, 16 | NO_TABLE, 32 | NO_TABLE, 256, 256 | NO_TABLE, 256 | 16, 256 | 16 | NO_TABLE):
        assert np.array_equal(bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc, flags=flags).wait(), want), flags


# ------------------------------------------------------------------------------ FFT
@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 8, 10, 11, 12, 13, 15, 16, 17, 18, 20])
def test_fft_all_modes_bit_exact(worker, log_n):
    """Appendix A 11,12,14: fft/ifft/coset_fft/icoset_fft == restated best_fft, every limb."""
    from bellman_amd import _lib

    lib = _lib.load()
    n = 1 << log_n
    data = cref.random_fr(n, 100 + log_n)
    for mode in (0, 1, 2, 3):
        got = data.copy()
        assert lib.bh_fft_fr(worker.ctx, _p(got), log_n, mode) == 0
        want = cref.fft(data, mode, threads=8)
        assert np.array_equal(got, want), (log_n, mode)


def test_fft_2_22_config_c3(worker):
    """BASELINE config C3: 2^22-point FFT/iFFT: exact vs the oracle + round-trip identities
    (domain.rs:427-463 fft_composition)."""
    import bellman_amd

    log_n = 22
    data = cref.random_fr(1 << log_n, 7)
    d = bellman_amd.EvaluationDomain.from_coeffs(worker, data)
    d.fft()
    got = d.as_ref()
    assert np.array_equal(got, cref.fft(data, 0, threads=cref.lib().orc_max_threads()))
    d.ifft()
    assert np.array_equal(d.as_ref(), data)
    d.coset_fft()
    d.icoset_fft()
    assert np.array_equal(d.as_ref(), data)
    d.icoset_fft()
    d.coset_fft()
    assert np.array_equal(d.into_coeffs(), data)


def test_fft_degree_too_large(worker):
    from bellman_amd import _lib

    buf = np.zeros((1, 4), dtype=np.uint64)
    assert _lib.load().bh_fft_fr(worker.ctx, _p(buf), 32, 0) == 3  # domain.rs:57-59


def test_domain_pointwise_ops(worker):
    """Appendix A 13: mul_assign, sub_assign, divide_by_z_on_coset, distribute_powers."""
    import bellman_amd

    n = 1 << 12
    a, b = cref.random_fr(n, 21), cref.random_fr(n, 22)
    da = bellman_amd.EvaluationDomain.from_coeffs(worker, a)
    db = bellman_amd.EvaluationDomain.from_coeffs(worker, b)
    da.mul_assign(worker, db)
    want = cref.mul_assign(a, b)
    assert np.array_equal(da.as_ref(), want)
    da.sub_assign(worker, db)
    want = cref.sub_assign(want, b)
    assert np.array_equal(da.as_ref(), want)
    da.divide_by_z_on_coset(worker)
    want = cref.divide_by_z_on_coset(want)
    assert np.array_equal(da.as_ref(), want)
    g = cref.fr_to_mont(cref.ints_to_arr([123456789], 4))
    da.distribute_powers(worker, g[0])
    w2 = want.copy()
    cref.lib().orc_distribute_powers(_p(w2), ctypes.c_size_t(n), _p(g), ctypes.c_int(8))
    assert np.array_equal(da.into_coeffs(), w2)
    db.into_coeffs()


@pytest.mark.parametrize("n_evals", [1, 5, 13, 646, 5000])
def test_h_poly_pipeline(worker, n_evals):
    """Appendix A 15: the fused h block of create_proof (prover.rs:221-240)."""
    from bellman_amd import _lib

    lib = _lib.load()
    a, b = cref.random_fr(n_evals, 31), cref.random_fr(n_evals, 32)
    c = cref.mul_assign(a, b)  # satisfied constraints: a*b = c on the domain
    m = 1
    while m < n_evals:
        m *= 2
    out = np.zeros((max(m - 1, 1), 4), dtype=np.uint64)
    hlen = ctypes.c_size_t(0)
    assert lib.bh_h_poly_fr(worker.ctx, _p(a), _p(b), _p(c), n_evals, _p(out), ctypes.byref(hlen)) == 0
    assert hlen.value == m - 1
    pad = lambda v: np.concatenate([v, np.zeros((m - n_evals, 4), dtype=np.uint64)])  # noqa: E731
    want = cref.h_coeffs(pad(a), pad(b), pad(c))
    assert np.array_equal(out[: m - 1], want)


# ------------------------------------------------------------------------------ MSM stages
@pytest.mark.parametrize("n,c", [(1, 4), (100, 4), (5000, 7), (70000, 11), (1 << 17, 16), (4097, 17), (50000, 19), (1 << 17, 20), (9000, 21)])
def test_msm_sort_stages(worker, n, c):
    """signed digits + stable radix sort + zero-digit count against numpy ([r5]: also the window sizes that take three
    passes, c = 17 ... 21 - written for the 10-bit passes of round 5, which sorted correctly and slower)."""
    from bellman_amd import _lib

    lib = _lib.load()
    sc = cref.random_fr(n, 40 + c)
    if n > 10:
        sc[3] = 0
        sc[4] = sc[5]
    W = (256 + c - 1) // c
    half = 1 << (c - 1)
    pairs = np.zeros(W * n, dtype=np.uint64)
    zstart = np.zeros(W, dtype=np.uint32)
    assert lib.bh_msm_debug_stages(worker.ctx, _p(sc), n, 0, c, _p(pairs), _p(zstart)) == 0
    ints = cref.arr_to_ints(sc)
    # signed-digit recoding: d in [-(2^(c-1)-1), 2^(c-1)], sum d_w 2^(c w) == scalar
    mags = np.zeros((W, n), dtype=np.uint64)
    negs = np.zeros((W, n), dtype=np.uint64)
    for i, v in enumerate(ints):
        carry, acc = 0, 0
        for w in range(W):
            d = ((v >> (c * w)) & ((1 << c) - 1)) + carry
            carry = 0
            if d > half:
                d -= 1 << c
                carry = 1
            acc += d << (c * w)
            mags[w, i], negs[w, i] = abs(d), 1 if d < 0 else 0
        assert carry == 0 and acc == v
    idx = np.arange(n, dtype=np.uint64)
    for w in range(W):
        order = np.argsort(mags[w], kind="stable")
        want = (mags[w][order] << np.uint64(32)) | (negs[w][order] << np.uint64(31)) | idx[order]
        got = pairs[w * n : (w + 1) * n]
        assert np.array_equal(got, want), (w,)
        assert zstart[w] == int((mags[w] == 0).sum())


# ------------------------------------------------------------------------------ MSM
def _scalars(n, seed, special=True):
    sc = cref.random_fr(n, seed)
    if special and n >= 12:
        sc[1] = 0
        sc[2] = cref.ints_to_arr([1], 4)[0]
        sc[3] = cref.ints_to_arr([Q - 1], 4)[0]
        sc[4] = cref.ints_to_arr([1 << 200], 4)[0]
        sc[5] = sc[6]
        sc[7] = cref.ints_to_arr([2], 4)[0]
    return sc


@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 100, 1000, 4113, 1 << 14])
def test_msm_full_density_matches_oracle(worker, group, n):
    """Appendix A 1,2,3,4,5,7,9 (n < 32 / >= 32 window rule boundary included).  Every size runs through the default
    plan (window table for vectors this small, host path for a handful of terms) AND through the classic
    W-window pipeline (NO_TABLE | NO_SMALL_PATH)."""
    import bellman_amd
    from bellman_amd.multiexp import NO_SMALL_PATH, NO_TABLE

    if group == 2 and n > 5000:
        pytest.skip("G2 large case covered by test_msm_g2_2_14")
    bases = cref.gen_bases(group, n, a=3, b=5)
    if n >= 12:
        bases[9] = bases[8]  # duplicate base
    sc = _scalars(n, 50 + n)
    hb = bellman_amd.Bases(worker, group, bases)
    got = bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc).wait()
    rc, want = cref.multiexp(group, bases, 0, None, sc)
    assert rc == 0
    assert np.array_equal(got, want)
    classic = bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc, flags=NO_TABLE | NO_SMALL_PATH).wait()
    assert np.array_equal(classic, want)


def test_msm_g2_2_14(worker):
    import bellman_amd

    n = 1 << 14
    bases = cref.gen_bases(2, n, a=9, b=2)
    sc = _scalars(n, 77)
    hb = bellman_amd.Bases(worker, 2, bases)
    got = bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc).wait()
    rc, want = cref.multiexp(2, bases, 0, None, sc)
    assert rc == 0 and np.array_equal(got, want)


def test_msm_empty_is_identity(worker):
    import bellman_amd

    hb = bellman_amd.Bases(worker, 1, cref.gen_bases(1, 4))
    got = bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), np.zeros((0, 4), dtype=np.uint64)).wait()
    assert not got.any()


@pytest.mark.parametrize("group", [1, 2])
def test_msm_density_skip_and_montgomery_scalars(worker, group):
    """Appendix A 1: dense entries only, base index = skip + rank; scalars given as Rust `Scalar`s."""
    import bellman_amd

    n = 3000
    rnd = np.random.default_rng(3)
    bits = rnd.random(n) < 0.55
    nb = int(bits.sum()) + 7
    bases = cref.gen_bases(group, nb, a=2, b=7)
    sc = _scalars(n, 91)
    hb = bellman_amd.Bases(worker, group, bases)
    dt = bellman_amd.DensityTracker(bits)
    got = bellman_amd.multiexp(worker, hb, dt, sc, skip=7).wait()
    rc, want = cref.multiexp(group, bases, 7, cref.density_bitmap(bits), sc)
    assert rc == 0 and np.array_equal(got, want)
    got2 = bellman_amd.multiexp(worker, hb, dt, cref.fr_to_mont(sc), skip=7, mont=True).wait()
    assert np.array_equal(got2, want)


@pytest.mark.parametrize("group", [1, 2])
def test_bases_register_rust_struct_layout(worker, group):
    """bh_bases_register with a record stride and an `infinity` flag byte: the in-memory layout of
    bls12_381's G1Affine/G2Affine { x, y, infinity: Choice }.  A flagged record is
    the identity whatever its coordinate bytes say."""
    import bellman_amd
    from bellman_amd import UnexpectedIdentity

    n = 500
    w = 12 if group == 1 else 24
    rec = w * 8
    stride = rec + 8
    bases = cref.gen_bases(group, n, a=21, b=2)
    raw = np.zeros((n, stride), dtype=np.uint8)
    raw[:, :rec] = bases.view(np.uint8).reshape(n, rec)
    raw[:, rec + 1 :] = 0xAB  # padding garbage
    raw[77, rec] = 1  # identity flag set on a record that still carries coordinates
    sc = _scalars(n, 23)
    hb = bellman_amd.Bases(worker, group, raw, stride=stride, inf_offset=rec)
    with pytest.raises(UnexpectedIdentity):
        bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc).wait()
    sc[77] = 0  # an identity under a zero scalar is skipped (multiexp.rs:245)
    got = bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc).wait()
    ref_bases = bases.copy()
    ref_bases[77] = 0
    rc, want = cref.multiexp(group, ref_bases, 0, None, sc)
    assert rc == 0 and np.array_equal(got, want)


@pytest.mark.parametrize("group", [1, 2])
def test_bases_from_serialized_crs_bytes(worker, group):
    """bh_bases_register_uncompressed: bellman's on-disk query format (uncompressed Zcash encoding,
    groth16/src/lib.rs:258-287) decoded on the device == the same points registered as records."""
    import bellman_amd

    n = 300
    bases = cref.gen_bases(group, n, a=31, b=8)
    bases[13] = 0  # serialises with the infinity flag
    pts = (cref.g1_to_py if group == 1 else cref.g2_to_py)(bases)
    enc = bls.g1_uncompressed if group == 1 else bls.g2_uncompressed
    blob = b"".join(enc(p) for p in pts)
    assert len(blob) == n * (96 if group == 1 else 192)
    sc = _scalars(n, 29)
    sc[13] = 0
    hb = bellman_amd.Bases.from_uncompressed(worker, group, blob)
    got = bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc).wait()
    rc, want = cref.multiexp(group, bases, 0, None, sc)
    assert rc == 0 and np.array_equal(got, want)
    with pytest.raises(AssertionError):  # compressed-form flag is rejected (BH_ERR_INVALID_ARG)
        bad = bytearray(blob)
        bad[0] |= 0x80
        bellman_amd.Bases.from_uncompressed(worker, group, bytes(bad))


def test_msm_skewed_scalars_split_buckets(worker):
    """All scalars equal / boolean-heavy witnesses: exercises the split-bucket path."""
    import bellman_amd

    n = 20000
    bases = cref.gen_bases(1, n, a=11, b=13)
    hb = bellman_amd.Bases(worker, 1, bases)
    sc = np.tile(cref.ints_to_arr([0x1234567 + (5 << 100) + (7 << 250)], 4), (n, 1))
    got = bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc).wait()
    rc, want = cref.multiexp(1, bases, 0, None, sc)
    assert rc == 0 and np.array_equal(got, want)
    rnd = np.random.default_rng(4)
    sc2 = cref.random_fr(n, 5)
    kind = rnd.integers(0, 4, size=n)
    sc2[kind == 0] = 0
    sc2[kind == 1] = cref.ints_to_arr([1], 4)[0]
    got = bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc2).wait()
    rc, want = cref.multiexp(1, bases, 0, None, sc2)
    assert rc == 0 and np.array_equal(got, want)


def test_msm_error_semantics(worker):
    """Appendix A 6: EOF / UnexpectedIdentity, incl. which one wins when both occur."""
    import bellman_amd
    from bellman_amd import UnexpectedEof, UnexpectedIdentity

    n = 200
    bases = cref.gen_bases(1, n, a=4, b=9)
    sc = cref.random_fr(n, 8)
    sc[:, 3] |= np.uint64(1 << 60)  # top-window digit non-zero for every scalar

    def run(bs, skip, density, scalars):
        hb = bellman_amd.Bases(worker, 1, bs)
        dm = bellman_amd.FullDensity() if density is None else bellman_amd.DensityTracker(density)
        try:
            bellman_amd.multiexp(worker, hb, dm, scalars, skip=skip).wait()
            got = 0
        except UnexpectedIdentity:
            got = 1
        except UnexpectedEof:
            got = 2
        want, _ = cref.multiexp(1, bs, skip, None if density is None else cref.density_bitmap(density), scalars)
        assert got == want
        return got

    assert run(bases[:-1], 0, None, sc) == 2
    assert run(bases, 1, None, sc) == 2
    b2 = bases.copy()
    b2[17] = 0
    assert run(b2, 0, None, sc) == 1
    s2 = sc.copy()
    s2[17] = 0
    assert run(b2, 0, None, s2) == 0
    s3 = sc.copy()
    s3[17] = cref.ints_to_arr([5], 4)[0]  # identity only met in window 0; EOF reported by the top window
    assert run(b2[:-1], 0, None, s3) == 2
    assert run(b2[:-1], 0, None, sc) == 1  # identity met first in the top window
    assert run(bases[:0], 0, [False] * n, sc) == 0  # nothing dense: nothing consumed
    with pytest.raises(AssertionError):  # density length mismatch panics in the reference
        hb = bellman_amd.Bases(worker, 1, bases)
        bellman_amd.multiexp(worker, hb, bellman_amd.DensityTracker([True] * (n - 1)), sc).wait()


def test_msm_concurrent_jobs(worker):
    """Appendix A 8: create_proof keeps 8 multiexps in flight before waiting on any."""
    import bellman_amd

    n = 5000
    jobs, wants = [], []
    for i in range(8):
        g = 1 if i < 6 else 2
        bases = cref.gen_bases(g, n, a=i + 1, b=3)
        sc = cref.random_fr(n, 60 + i)
        hb = bellman_amd.Bases(worker, g, bases)
        jobs.append((hb, bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc)))
        wants.append(cref.multiexp(g, bases, 0, None, sc)[1])
    for (hb, j), want in zip(jobs, wants):
        assert np.array_equal(j.wait(), want)


def test_msm_window_bits_do_not_change_result(worker):
    """Appendix A 7: c is unobservable."""
    import bellman_amd
    from bellman_amd import _lib

    n = 3000
    bases = cref.gen_bases(1, n, a=6, b=1)
    sc = cref.random_fr(n, 9)
    hb = bellman_amd.Bases(worker, 1, bases)
    rc, want = cref.multiexp(1, bases, 0, None, sc)
    for c in (2, 3, 5, 8, 9, 13, 16):   # per-job override (bh_msm_opts): nothing on the context changes
        assert np.array_equal(bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc, window_bits=c).wait(), want), c


def test_msm_chunk_size_does_not_change_result(worker):
    """The accumulation chunk K only moves the boundaries of partial sums."""
    import bellman_amd
    from bellman_amd import _lib

    n = 6000
    bases = cref.gen_bases(1, n, a=8, b=5)
    sc = _scalars(n, 19)
    sc[100:400] = sc[100]  # a bucket spanning many chunks in every window
    hb = bellman_amd.Bases(worker, 1, bases)
    rc, want = cref.multiexp(1, bases, 0, None, sc)
    for k in (1, 2, 3, 7, 32, 100, 10000):
        assert np.array_equal(bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc, chunk=k).wait(), want), k


def test_msm_2_20_config_c2(worker):
    """BASELINE config C2: 2^20-base G1 MSM, bit-exact vs the restated multiexp, plus linearity."""
    import bellman_amd

    n = 1 << 20
    bases = cref.gen_bases(1, n, a=1, b=1)
    sc = cref.random_fr(n, 2020)
    hb = bellman_amd.Bases(worker, 1, bases)
    assert hb.table_info()[:2] == (20, 13)     # [r6] registered with its 13-row window table: the default plan of a 2^20-point query
    got, ms = bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc, timed=True).wait()
    rc, want = cref.multiexp(1, bases, 0, None, sc, threads=cref.lib().orc_max_threads())
    assert rc == 0 and np.array_equal(got, want)
    print("G1 MSM 2^20 device ms [total, sort, accumulate, reduce]:", ms)
    # the classic plan (16 windows over the plain vector) on the same inputs
    from bellman_amd.multiexp import NO_TABLE
    got, ms = bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc, timed=True, flags=NO_TABLE).wait()
    assert np.array_equal(got, want)
    print("... classic plan:", ms)
    # linearity: MSM(s, B[:h]) + MSM(s, B[h:]) == MSM(s, B)
    h = n // 2
    lo = bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc[:h]).wait()
    hi = bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc[h:], skip=h).wait()
    assert np.array_equal(cref.point_add(1, lo, hi), want)


@pytest.mark.parametrize("group", [1, 2])
def test_fixed_base_mul(worker, group):
    from bellman_amd import _lib

    lib = _lib.load()
    n = 500
    w = 12 if group == 1 else 24
    sc = _scalars(n, 13)
    gen = cref.g1_generator() if group == 1 else cref.g2_generator()
    dsc, dout = _dev(worker, sc), worker.alloc(n * w * 8)
    assert lib.bh_fixed_base_mul_dev(worker.ctx, group, _p(np.ascontiguousarray(gen)), dsc, n, 0, dout, None) == 0
    worker.synchronize()
    out = np.zeros((n, w), dtype=np.uint64)
    worker.download(out, dout)
    ints = cref.arr_to_ints(sc)
    for i in list(range(12)) + [100, 499]:
        assert np.array_equal(out[i], cref.point_mul(group, gen, ints[i])), i


@pytest.mark.parametrize("seed", range(12 + int(os.environ.get("BH_FUZZ_EXTRA", "0"))))   # BH_FUZZ_EXTRA=n: n more seeds
def test_msm_fuzz_random_shapes(worker, seed):
    """Randomised sweep over sizes, densities, skips, scalar mixes and the tuning knobs (c, K): chunk /
    bucket boundary handling (runs ending exactly on chunk borders, single-bucket chunks, long runs,
    empty windows) must never change the group element."""
    import bellman_amd
    from bellman_amd import _lib

    lib = _lib.load()
    rnd = np.random.default_rng(1000 + seed)
    for _ in range(5):
        group = 1 if rnd.random() < 0.75 else 2
        n = int(rnd.choice([3, 17, 64, 65, 127, 500, 1023, 2048, 3001, 9000]))
        if group == 2:
            n = min(n, 2048)
        sc = cref.random_fr(n, int(rnd.integers(1 << 30)))
        mode = int(rnd.integers(5))
        if mode == 1:  # few distinct scalars -> very long bucket runs
            sc[:] = sc[rnd.integers(0, 3, size=n)]
        elif mode == 2:  # small scalars: upper windows empty
            sc[:, 1:] = 0
        elif mode == 3:  # boolean-heavy witness
            k = rnd.integers(0, 3, size=n)
            sc[k == 0] = 0
            sc[k == 1] = cref.ints_to_arr([1], 4)[0]
        elif mode == 4:  # values near q and powers of two
            sc[::3] = cref.ints_to_arr([Q - 1], 4)[0]
            sc[1::3] = cref.ints_to_arr([1 << int(rnd.integers(1, 254))], 4)[0]
        dens = None
        nb, skip = n, 0
        if rnd.random() < 0.5:
            dens = rnd.random(n) < rnd.choice([0.1, 0.5, 0.9])
            skip = int(rnd.integers(0, 5))
            nb = int(dens.sum()) + skip
        bases = cref.gen_bases(group, max(nb, 1), a=int(rnd.integers(1, 1000)), b=int(rnd.integers(1, 1000)))[:nb]
        if nb > 10 and rnd.random() < 0.3:
            bases[5] = bases[4]
        c = int(rnd.choice([0, 2, 3, 4, 7, 8, 11, 13, 16]))
        K = int(rnd.choice([0, 1, 2, 5, 8, 16, 33, 1000]))
        # 4 = classic W-window plan instead of the window table; 16 / 32 = G2 kernel bundle; 8 = no host path
        flags = int(rnd.choice([0, 4, 8, 12])) | (int(rnd.choice([0, 16, 32])) if group == 2 else 0)
        hb = bellman_amd.Bases(worker, group, bases)
        dm = bellman_amd.FullDensity() if dens is None else bellman_amd.DensityTracker(dens)
        got = bellman_amd.multiexp(worker, hb, dm, sc, skip=skip, window_bits=c, chunk=K, flags=flags).wait()
        rc, want = cref.multiexp(group, bases, skip, None if dens is None else cref.density_bitmap(dens), sc)
        assert rc == 0
        assert np.array_equal(got, want), (group, n, mode, c, K, skip, dens is not None, flags)


@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("n,cbits", [(1, 0), (2, 3), (37, 0), (1000, 5), (1000, 0), (1000, 8), (5000, 11), (1 << 14, 0), (1 << 14, 16)])
def test_multiexp_with_window_table(worker, group, n, cbits):
    """bh_bases_precompute: all windows into one bucket set via the stored multiples 2^(c*j) P -
    the same group element as without the table and as the oracle, with skip and a density map."""
    import bellman_amd

    bases = cref.gen_bases(group, n + 3, a=5, b=11)
    sc = cref.random_fr(n, 77 + n)
    if n > 4:
        sc[0] = 0
        sc[1] = 0
        sc[1, 0] = 1                                       # scalar 1
        sc[2] = cref.ints_to_arr([bls.Q - 1], 4)[0]         # q - 1: top window + carries
    hb = bellman_amd.Bases(worker, group, bases)
    plain = bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc, skip=2).wait()
    hb.precompute(cbits)
    c_used, rows, nbytes = hb.table_info()
    assert rows == (256 + c_used - 1) // c_used and nbytes == rows * (n + 3) * (96 if group == 1 else 192)
    rc, want = cref.multiexp(group, bases, 2, None, sc)
    assert rc == 0
    got = bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc, skip=2).wait()
    assert np.array_equal(got, want) and np.array_equal(plain, want)
    rnd = random.Random(n)
    bits = [rnd.random() < 0.6 for _ in range(n)]
    rc, want = cref.multiexp(group, bases, 1, cref.density_bitmap(bits), sc)
    assert rc == 0
    assert np.array_equal(bellman_amd.multiexp(worker, hb, bellman_amd.DensityTracker(bits), sc, skip=1).wait(), want)


def test_window_table_error_semantics(worker):
    """identity bases and short base vectors behave the same with a table (Appendix A item 6)"""
    import bellman_amd

    n = 300
    bases = cref.gen_bases(1, n, a=3, b=7)
    bases[17] = 0
    sc = cref.random_fr(n, 9)
    hb = bellman_amd.Bases(worker, 1, bases).precompute()
    with pytest.raises(bellman_amd.UnexpectedIdentity):
        bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc).wait()
    sc[17] = 0   # an identity under a zero scalar is skipped silently
    rc, want = cref.multiexp(1, bases, 0, None, sc)
    assert rc == 0 and np.array_equal(bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc).wait(), want)
    with pytest.raises(bellman_amd.UnexpectedEof):
        bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc, skip=1).wait()


def test_worker_compute_and_scope_host_helpers(worker):
    """multicore.rs:33-91 mirrors: compute -> Waiter; scope chunking rule and join-on-return"""
    w = worker.compute(lambda: sum(range(1000)))
    assert w.wait() == 499500
    import os

    n_thr = os.cpu_count() or 1
    out = [0] * 1000

    def body(scope, chunk):
        assert chunk == (1 if len(out) < n_thr else len(out) // n_thr)
        for lo in range(0, len(out), chunk):
            def task(_scope, lo=lo):
                for i in range(lo, min(len(out), lo + chunk)):
                    out[i] = i * i
            scope.spawn(task)
        return chunk

    assert worker.scope(len(out), body) >= 1
    assert out == [i * i for i in range(1000)]      # every spawned task finished before scope returned
    assert worker.scope(3, lambda s, chunk: chunk) == (1 if 3 < n_thr else 3 // n_thr)
