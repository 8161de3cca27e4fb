"""Validates the C oracle (oracle/c) against the KAT-pinned big-int oracle
(oracle/pyref): field ops, curve ops, multiexp semantics (Appendix A items 1-9 of
SURVEY.md) and FFT/domain ops (items 10-15), all over BLS12-381."""

import random

import numpy as np
import pytest

from oracle import cref
from oracle.pyref import bls12_381 as bls
from oracle.pyref import multiexp as pm
from oracle.pyref.domain import EvaluationDomain, parallel_fft, serial_fft
from oracle.pyref.engines import Bls12
from oracle.pyref.errors import IoErrorUnexpectedEof, UnexpectedIdentity
from oracle.pyref.multicore import Worker

Q, P = bls.Q, bls.P


def test_constants():
    assert cref.limbs_to_int(cref.fr_to_mont(cref.ints_to_arr([1], 4))[0]) == (1 << 256) % Q
    assert cref.limbs_to_int(cref.fp_to_mont(cref.ints_to_arr([1], 6))[0]) == (1 << 384) % P
    # compressed generator KAT (Zcash encoding), SURVEY §8c (4)
    assert bls.g1_compress(bls.G1_GEN).hex() == (
        "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac58"
        "6c55e83ff97a1aeffb3af00adb22c6bb"
    )
    assert bls.FR_ROOT_OF_UNITY == 0x16A2A19EDFE81F20D09B681922C813B4B63683508C2280B93829971F439F0D2B


def test_field_mul_matches_bigint():
    rnd = random.Random(1)
    xs = [rnd.randrange(Q) for _ in range(50)] + [0, 1, Q - 1]
    ys = [rnd.randrange(Q) for _ in range(50)] + [Q - 1, Q - 1, Q - 1]
    xm, ym = cref.fr_to_mont(cref.ints_to_arr(xs, 4)), cref.fr_to_mont(cref.ints_to_arr(ys, 4))
    prod = cref.mul_assign(xm, ym)
    assert cref.arr_to_ints(cref.fr_from_mont(prod)) == [(x * y) % Q for x, y in zip(xs, ys)]
    assert cref.arr_to_ints(cref.fr_from_mont(cref.sub_assign(xm, ym))) == [(x - y) % Q for x, y in zip(xs, ys)]


def _rand_points(curve, n, rnd):
    return [curve.mul(curve.gen, rnd.randrange(1, Q)) for _ in range(n)]


@pytest.mark.parametrize("group", [1, 2])
def test_curve_ops_match_affine_bigint(group):
    rnd = random.Random(2 + group)
    curve = bls.G1 if group == 1 else bls.G2
    conv_to = cref.g1_from_py if group == 1 else cref.g2_from_py
    conv_from = cref.g1_to_py if group == 1 else cref.g2_to_py
    pts = _rand_points(curve, 6, rnd)
    arr = conv_to(pts + [None])
    assert cref.on_curve(group, arr[0])
    # add: generic, doubling, inverse, identity operands (complete formulas)
    cases = [(0, 1), (2, 2), (3, 6), (6, 4), (6, 6)]
    for i, j in cases:
        got = conv_from(cref.point_add(group, arr[i], arr[j]))[0]
        a = pts[i] if i < 6 else None
        b = pts[j] if j < 6 else None
        assert got == curve.add(a, b)
    neg = conv_to([curve.neg(pts[0])])[0]
    assert conv_from(cref.point_add(group, arr[0], neg))[0] is None
    for k in [0, 1, 2, Q - 1, rnd.randrange(Q)]:
        assert conv_from(cref.point_mul(group, arr[1], k))[0] == curve.mul(pts[1], k)
    # [q]G = identity
    k_q = np.array(cref.int_to_limbs(Q, 4), dtype=np.uint64)
    out = np.zeros(12 if group == 1 else 24, dtype=np.uint64)
    fn = cref.lib().orc_g1_mul if group == 1 else cref.lib().orc_g2_mul
    fn(cref._p(out), cref._p(np.ascontiguousarray(arr[0])), cref._p(k_q))
    assert not out.any()
    # synthetic base generator
    gb = conv_from(cref.gen_bases(group, 5, a=3, b=2))
    assert gb == [curve.mul(curve.gen, 3 + 2 * i) for i in range(5)]


def _py_multiexp(curve, bases, offset, density, scalars):
    d = pm.FullDensity()
    if density is not None:
        d = pm.DensityTracker()
        d.bv = list(density)
    exps = [pm.exponent_from(s) for s in scalars]
    return pm.multiexp(Worker(), curve, Bls12.Fr, bases, offset, d, exps).wait()


@pytest.mark.parametrize("group", [1, 2])
def test_multiexp_matches_pyref_with_density_and_trivial_scalars(group):
    rnd = random.Random(10 + group)
    curve = bls.G1 if group == 1 else bls.G2
    conv_to = cref.g1_from_py if group == 1 else cref.g2_from_py
    conv_from = cref.g1_to_py if group == 1 else cref.g2_to_py
    n = 40 if group == 1 else 33
    scalars = [rnd.randrange(Q) for _ in range(n)]
    scalars[3] = 0
    scalars[5] = 1
    scalars[7] = Q - 1
    scalars[9] = 1 << 200
    density = [rnd.random() < 0.6 for _ in range(n)]
    nb = sum(density) + 2
    bases = _rand_points(curve, nb, rnd)
    bases[4] = bases[1]  # duplicate base
    arr = conv_to(bases)
    sc = cref.ints_to_arr(scalars, 4)
    # density + skip offset 2
    want = _py_multiexp(curve, bases, 2, density, scalars)
    rc, got = cref.multiexp(group, arr, 2, cref.density_bitmap(density), sc)
    assert rc == 0 and conv_from(got)[0] == want
    # full density, offset 0
    full_bases = _rand_points(curve, n, rnd)
    want = _py_multiexp(curve, full_bases, 0, None, scalars)
    rc, got = cref.multiexp(group, conv_to(full_bases), 0, None, sc)
    assert rc == 0 and conv_from(got)[0] == want
    assert want == pm.naive_multiexp(curve, full_bases, scalars)
    assert conv_from(cref.naive_multiexp(group, conv_to(full_bases), sc))[0] == want


def test_multiexp_small_n_and_empty():
    rnd = random.Random(5)
    for n in [0, 1, 2, 31, 32, 33]:
        scalars = [rnd.randrange(Q) for _ in range(n)]
        bases = _rand_points(bls.G1, n, rnd)
        want = _py_multiexp(bls.G1, bases, 0, None, scalars)
        rc, got = cref.multiexp(1, cref.g1_from_py(bases) if n else np.zeros((0, 12), np.uint64), 0, None,
                                cref.ints_to_arr(scalars, 4) if n else np.zeros((0, 4), np.uint64))
        assert rc == 0 and cref.g1_to_py(got)[0] == want
    assert cref.window_size(31) == 3 and cref.window_size(32) == 4
    assert cref.window_size(1 << 20) == 14 and cref.window_size(1 << 26) == 19


def test_multiexp_error_semantics():
    """Appendix A item 6."""
    rnd = random.Random(6)
    n = 40
    scalars = [rnd.randrange(2, Q) for _ in range(n)]
    bases = _rand_points(bls.G1, n, rnd)
    sc = cref.ints_to_arr(scalars, 4)

    def both(bases_py, offset, density, scalars_py):
        try:
            _py_multiexp(bls.G1, bases_py, offset, density, scalars_py)
            want = 0
        except UnexpectedIdentity:
            want = 1
        except IoErrorUnexpectedEof:
            want = 2
        rc, _ = cref.multiexp(1, cref.g1_from_py(bases_py), offset, None if density is None else cref.density_bitmap(density),
                              cref.ints_to_arr(scalars_py, 4))
        assert rc == want
        return rc

    assert both(bases[:-1], 0, None, scalars) == 2  # one base short
    assert both(bases, 1, None, scalars) == 2  # skip eats one
    b2 = list(bases)
    b2[7] = None
    assert both(b2, 0, None, scalars) == 1  # identity consumed
    s2 = list(scalars)
    s2[7] = 0
    assert both(b2, 0, None, s2) == 0  # identity under a zero scalar is skipped
    # identity + EOF: which error wins depends on the top window digit of the identity's scalar
    s3 = list(scalars)
    s3[7] = 5  # top-window digit 0 -> EOF reported
    assert both(b2[:-1], 0, None, s3) == 2
    s3[7] = Q - 1  # top-window digit non-zero -> identity reported (met first in index order)
    assert both(b2[:-1], 0, None, s3) == 1
    # no dense entries: nothing consumed, no EOF even with no bases
    assert both([], 0, [False] * n, scalars) == 0


def test_fft_variants_match_pyref():
    rnd = random.Random(7)
    F = Bls12.Fr
    for log_n in [0, 1, 2, 3, 5, 8]:
        n = 1 << log_n
        vals = [rnd.randrange(Q) for _ in range(n)]
        mont = cref.fr_to_mont(cref.ints_to_arr(vals, 4))
        for mode, name in [(0, "fft"), (1, "ifft"), (2, "coset_fft"), (3, "icoset_fft")]:
            for threads in (1, 8):
                d = EvaluationDomain.from_coeffs(F, vals)
                getattr(d, name)(Worker(threads))
                got = cref.arr_to_ints(cref.fr_from_mont(cref.fft(mont, mode, threads=threads)))
                assert got == d.coeffs, (log_n, name, threads)


def test_serial_and_parallel_fft_consistency():
    """domain.rs:465-498 re-expressed."""
    rnd = random.Random(8)
    for log_d in range(0, 8):
        d = 1 << log_d
        vals = [rnd.randrange(Q) for _ in range(d)]
        mont = cref.fr_to_mont(cref.ints_to_arr(vals, 4))
        omega = EvaluationDomain.from_coeffs(Bls12.Fr, vals).omega
        v1 = list(vals)
        serial_fft(v1, Q, omega, log_d)
        assert cref.arr_to_ints(cref.fr_from_mont(cref.serial_fft(mont))) == v1
        for log_cpus in range(0, log_d + 1):
            v2 = list(vals)
            parallel_fft(v2, Q, omega, log_d, log_cpus)
            assert v2 == v1
            assert cref.arr_to_ints(cref.fr_from_mont(cref.parallel_fft(mont, log_cpus))) == v1


def test_h_coeffs_match_pyref():
    from oracle.pyref.prover import compute_h_coeffs

    rnd = random.Random(9)
    n = 13  # padded to 16
    a = [rnd.randrange(Q) for _ in range(n)]
    b = [rnd.randrange(Q) for _ in range(n)]
    c = [(x * y) % Q for x, y in zip(a, b)]  # satisfiable -> exact quotient
    want = compute_h_coeffs(Bls12.Fr, a, b, c, Worker(8))
    pad = lambda v: cref.fr_to_mont(cref.ints_to_arr(v + [0] * (16 - n), 4))
    got = cref.arr_to_ints(cref.fr_from_mont(cref.h_coeffs(pad(a), pad(b), pad(c))))
    assert got == want
