"""The reference's OWN property tests, run through the product API with nothing of the builder's oracle in the loop
(run with `pytest -m gpu` on a MI355X).  The reference holds no BLS12-381 golden vectors (every BLS test draws from
`thread_rng`), so these are what pins the BLS layer independently of `oracle/c` and of `oracle/pyref`'s restated
algorithms: the checker here is Python-integer arithmetic only (`oracle.pyref.bls12_381` is imported for the published
curve constants and its textbook affine group law - integers, no multiexp, no FFT).

  * polynomial_arith          src/domain.rs:376-425   all 70 x 70 length pairs: fft, fft, mul_assign, ifft on the device
                                                      == the schoolbook product
  * fft_composition           src/domain.rs:427-463   the four round trips, at every size 2^0 .. 2^17 (the reference: 2^0 .. 2^9)
  * parallel_fft_consistency  src/domain.rs:465-498   the device transform == a plain serial radix-2 transform written here
                                                      in Python integers (the reference compares its two CPU variants)
  * test_with_bls12           src/multiexp.rs:334-378 multiexp == the naive sum of scalar multiples, 2^10 terms (2^14 there)
Integer work: every limb equal, no tolerances."""

import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.test_gpu_groth16 import worker  # noqa: E402,F401

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R = (1 << 256) % Q
R_INV = pow(R, -1, Q)
MASK64 = (1 << 64) - 1


def to_mont(vals):
    """Python integers -> the in-memory form of `bls12_381::Scalar` (4 x u64 Montgomery limbs)"""
    a = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        m = v * R % Q
        a[i] = [(m >> (64 * k)) & MASK64 for k in range(4)]
    return a


def from_mont(a):
    out = []
    for row in a:
        m = 0
        for k in range(4):
            m |= int(row[k]) << (64 * k)
        assert m < Q, "non-canonical limbs left the device"
        out.append(m * R_INV % Q)
    return out


def test_polynomial_arith(worker):
    """src/domain.rs:376-425, verbatim: for every (coeffs_a, coeffs_b) in 0..70 x 0..70, random polynomials a, b resized
    to coeffs_a + coeffs_b; fft(a), fft(b), a.mul_assign(b), a.ifft() equals the naive product on every coefficient."""
    import bellman_amd

    rnd = random.Random(0x706F6C79)
    for coeffs_a in range(70):
        for coeffs_b in range(70):
            a = [rnd.randrange(Q) for _ in range(coeffs_a)]
            b = [rnd.randrange(Q) for _ in range(coeffs_b)]
            naive = [0] * (coeffs_a + coeffs_b)
            for i1, x in enumerate(a):
                for i2, y in enumerate(b):
                    naive[i1 + i2] = (naive[i1 + i2] + x * y) % Q
            n = coeffs_a + coeffs_b
            da = bellman_amd.EvaluationDomain.from_coeffs(worker, to_mont(a + [0] * (n - coeffs_a)))
            db = bellman_amd.EvaluationDomain.from_coeffs(worker, to_mont(b + [0] * (n - coeffs_b)))
            da.fft(worker)
            db.fft(worker)
            da.mul_assign(worker, db)
            da.ifft(worker)
            got = from_mont(da.into_coeffs())
            db.into_coeffs()
            assert got[:n] == naive, (coeffs_a, coeffs_b)
            assert all(v == 0 for v in got[n:]), (coeffs_a, coeffs_b)   # degree < n: the padding stays zero


@pytest.mark.parametrize("exp", list(range(18)))
def test_fft_composition(worker, exp):
    """src/domain.rs:427-463: ifft.fft, fft.ifft, icoset_fft.coset_fft, coset_fft.icoset_fft are the identity (the
    reference runs 2^0 .. 2^9; here to 2^17: one-, two- and, with the one-level tables, both table kinds)."""
    import bellman_amd

    rng = np.random.default_rng(1000 + exp)
    n = 1 << exp
    # uniform field elements as Montgomery limbs: any canonical limbs are the Montgomery form of SOME element
    v = np.zeros((n, 4), dtype=np.uint64)
    raw = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    for i in range(n):
        x = 0
        for k in range(4):
            x |= int(raw[i, k]) << (64 * k)
        x %= Q
        v[i] = [(x >> (64 * k)) & MASK64 for k in range(4)]
    d = bellman_amd.EvaluationDomain.from_coeffs(worker, v)
    for first, second in ((d.ifft, d.fft), (d.fft, d.ifft), (d.icoset_fft, d.coset_fft), (d.coset_fft, d.icoset_fft)):
        first(worker)
        second(worker)
        assert np.array_equal(d.as_ref(), v), exp
    d.into_coeffs()


def _serial_fft(a, omega, log_n):
    """src/domain.rs:272-314 in Python integers: bit-reversal, then log_n rounds of butterflies"""
    n = len(a)
    a = list(a)
    for k in range(n):
        rk = int(bin(k)[2:].zfill(log_n)[::-1], 2) if log_n else 0
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    m = 1
    for _ in range(log_n):
        w_m = pow(omega, n // (2 * m), Q)
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = a[k + j + m] * w % Q
                a[k + j + m] = (a[k + j] - t) % Q
                a[k + j] = (a[k + j] + t) % Q
                w = w * w_m % Q
        m *= 2
    return a


@pytest.mark.parametrize("log_d", list(range(13)))
def test_device_fft_is_the_serial_fft(worker, log_d):
    """src/domain.rs:465-498 (parallel_fft_consistency): however the transform is decomposed, it equals serial_fft - here
    the device's pass / tile / radix-8 decomposition against the reference's serial loop in Python integers, and the
    coset variant against distribute_powers + the same loop (src/domain.rs:101-125)."""
    import bellman_amd

    rnd = random.Random(9000 + log_d)
    n = 1 << log_d
    v = [rnd.randrange(Q) for _ in range(n)]
    omega = pow(pow(7, (Q - 1) >> 32, Q), 1 << (32 - log_d), Q)   # domain.rs:62-66
    d = bellman_amd.EvaluationDomain.from_coeffs(worker, to_mont(v))
    d.fft(worker)
    assert from_mont(d.as_ref()) == _serial_fft(v, omega, log_d)
    d.ifft(worker)
    assert from_mont(d.as_ref()) == v
    d.coset_fft(worker)
    assert from_mont(d.into_coeffs()) == _serial_fft([x * pow(7, i, Q) % Q for i, x in enumerate(v)], omega, log_d)


def _jac_dbl(p1, P):
    """2 (X : Y : Z) on y^2 = x^3 + b in Jacobian coordinates (textbook: "dbl-2009-l", a = 0); None = the identity"""
    if p1 is None or p1[1] == 0:
        return None
    x, y, z = p1
    a = x * x % P
    b = y * y % P
    c = b * b % P
    d = 2 * ((x + b) * (x + b) - a - c) % P
    e = 3 * a % P
    x3 = (e * e - 2 * d) % P
    return x3, (e * (d - x3) - 8 * c) % P, 2 * y * z % P


def _jac_add(p1, p2, P):
    """textbook "add-2007-bl" with the exceptional cases"""
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1, z1 = p1
    x2, y2, z2 = p2
    z1z1, z2z2 = z1 * z1 % P, z2 * z2 % P
    u1, u2 = x1 * z2z2 % P, x2 * z1z1 % P
    s1, s2 = y1 * z2 * z2z2 % P, y2 * z1 * z1z1 % P
    if u1 == u2:
        return _jac_dbl(p1, P) if s1 == s2 else None
    h = (u2 - u1) % P
    i = 4 * h * h % P
    j = h * i % P
    r = 2 * (s2 - s1) % P
    v = u1 * i % P
    x3 = (r * r - j - 2 * v) % P
    return x3, (r * (v - x3) - 2 * s1 * j) % P, ((z1 + z2) * (z1 + z2) - z1z1 - z2z2) * h % P


def _jac_mul(pt_affine, k, P):
    acc, base = None, (pt_affine[0], pt_affine[1], 1)
    for bit in bin(k)[2:]:
        acc = _jac_dbl(acc, P)
        if bit == "1":
            acc = _jac_add(acc, base, P)
    return acc


def _jac_to_affine(p1, P):
    if p1 is None:
        return None
    zi = pow(p1[2], -1, P)
    return p1[0] * zi * zi % P, p1[1] * zi * zi * zi % P


def test_with_bls12(worker):
    """src/multiexp.rs:334-378: multiexp(bases, FullDensity, scalars) == the naive sum of [s_i] P_i, with the scalar
    multiples and the sum computed by textbook formulas on Python integers - no oracle algorithm, no product code.
    [r6] The 1 024 double-and-add ladders run in Jacobian coordinates written out above (one inversion at the end instead
    of one per group operation: 60 s -> 6 s of the suite); the helper itself is checked against the affine textbook
    formulas of oracle/pyref/bls12_381.py on the first points."""
    import bellman_amd
    from oracle.pyref import bls12_381 as bls

    samples = 1 << 10
    rnd = random.Random(0x626C73)
    scalars = [rnd.randrange(Q) for _ in range(samples)]
    step = bls.G1.mul(bls.G1.gen, rnd.randrange(1, Q))
    pts, cur = [], bls.G1.mul(bls.G1.gen, rnd.randrange(1, Q))
    for _ in range(samples):
        pts.append(cur)
        cur = bls.G1.add(cur, step)
    for p, s in list(zip(pts, scalars))[:3]:
        assert _jac_to_affine(_jac_mul(p, s, bls.P), bls.P) == bls.G1.mul(p, s)
    total = None
    for p, s in zip(pts, scalars):
        total = _jac_add(total, _jac_mul(p, s, bls.P), bls.P)
    naive = _jac_to_affine(total, bls.P)
    # in-memory forms: canonical scalars (what `Exponent::from(&Scalar)` hands multiexp), Montgomery affine coordinates
    fp_r = (1 << 384) % bls.P
    bases = np.zeros((samples, 12), dtype=np.uint64)
    for i, (x, y) in enumerate(pts):
        for c, v in enumerate((x * fp_r % bls.P, y * fp_r % bls.P)):
            bases[i, 6 * c: 6 * c + 6] = [(v >> (64 * k)) & MASK64 for k in range(6)]
    sc = np.zeros((samples, 4), dtype=np.uint64)
    for i, s in enumerate(scalars):
        sc[i] = [(s >> (64 * k)) & MASK64 for k in range(4)]
    hb = bellman_amd.Bases(worker, 1, bases)
    got = bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc).wait()
    got = np.asarray(got, dtype=np.uint64).reshape(12)
    fp_rinv = pow(fp_r, -1, bls.P)
    coords = []
    for c in range(2):
        v = 0
        for k in range(6):
            v |= int(got[6 * c + k]) << (64 * k)
        coords.append(v * fp_rinv % bls.P)
    assert naive is not None and tuple(coords) == naive
