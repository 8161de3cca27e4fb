"""Test helpers: points that are on the curve but outside the prime-order subgroup, off-curve points,
and small serialized Parameters blobs (oracle side only)."""

from oracle.pyref import bls12_381 as bls

P = bls.P


def _fp_sqrt(a):
    r = pow(a, (P + 1) // 4, P)   # p = 3 mod 4
    return r if r * r % P == a % P else None


def _fp2_sqrt(a):
    a0, a1 = a
    if a1 == 0:
        r = _fp_sqrt(a0)
        if r is not None:
            return (r, 0)
        r = _fp_sqrt((-a0) % P)
        return None if r is None else (0, r)
    s = _fp_sqrt((a0 * a0 + a1 * a1) % P)
    if s is None:
        return None
    inv2 = pow(2, -1, P)
    for t in ((a0 + s) * inv2 % P, (a0 - s) * inv2 % P):
        x0 = _fp_sqrt(t)
        if x0:
            x1 = a1 * pow(2 * x0, -1, P) % P
            if bls.fp2_mul((x0, x1), (x0, x1)) == (a0 % P, a1 % P):
                return (x0, x1)
    return None


def g1_on_curve_not_in_subgroup(seed):
    x = seed
    while True:
        y = _fp_sqrt((x * x * x + 4) % P)
        if y is not None and bls.G1.mul((x, y), bls.Q) is not None:
            return (x, y)
        x += 1


def g2_on_curve_not_in_subgroup(seed):
    x = (seed, seed + 1)
    while True:
        rhs = bls.fp2_add(bls.fp2_mul(bls.fp2_mul(x, x), x), bls.G2_B)
        y = _fp2_sqrt(rhs)
        if y is not None and bls.G2.mul((x, y), bls.Q) is not None:
            return (x, y)
        x = (x[0] + 1, x[1])


def small_parameters(n_h=3, n_l=4, n_a=5, n_b=3, n_ic=2, k0=11):
    """a structurally valid Parameters blob's ingredients (distinct multiples of the generators)"""
    G1, G2 = bls.G1, bls.G2
    k = [k0]

    def g1():
        k[0] += 7
        return G1.mul(G1.gen, k[0])

    def g2():
        k[0] += 5
        return G2.mul(G2.gen, k[0])

    vk = dict(alpha_g1=g1(), beta_g1=g1(), beta_g2=g2(), gamma_g2=g2(), delta_g1=g1(), delta_g2=g2(), ic=[g1() for _ in range(n_ic)])
    return vk, [g1() for _ in range(n_h)], [g1() for _ in range(n_l)], [g1() for _ in range(n_a)], [g1() for _ in range(n_b)], [g2() for _ in range(n_b)]
