"""BLS12-381 optimal-ate pairing check on Python integers (TEST INFRASTRUCTURE).

Used only to restate the reference's `verify_proof` (groth16/src/verifier.rs:23-58), so that proofs made
by the HIP path can be checked the way the reference's own tests check theirs
(groth16/tests/mimc.rs:86-95 asserts `verify_proof(..).is_ok()`).  The pairing itself lives in the
third-party crate `bls12_381 0.8.0` (Cargo.lock:105-108, absent); what is restated here is the published
construction: Fp12 = Fp2[w]/(w^6 - (u+1)), the M-type twist E': y^2 = x^3 + 4(u+1) untwisted by
(x', y') -> (x'/w^2, y'/w^3), a Miller loop over |x| = 0xd201000000010000, and the final exponentiation
as a plain power by (p^12 - 1)/q.  Only products of pairings are compared with 1, so neither the sign of
x (a conjugation) nor subfield factors of the line functions matter - both vanish under the final power.
Deliberately simple and slow (about two seconds per check).
"""

from . import bls12_381 as bls

P, Q = bls.P, bls.Q
XI = (1, 1)   # u + 1
ATE_LOOP = 0xD201000000010000
_FINAL_EXP = (P ** 12 - 1) // Q

f2_add, f2_sub, f2_mul, f2_neg, f2_inv = bls.fp2_add, bls.fp2_sub, bls.fp2_mul, bls.fp2_neg, bls.fp2_inv
F12_ONE = ((1, 0),) + ((0, 0),) * 5


def f12_mul(a, b):
    """polynomials in w of degree < 6 over Fp2, reduced with w^6 = u + 1"""
    acc = [(0, 0)] * 11
    for i, ai in enumerate(a):
        if ai == (0, 0):
            continue
        for j, bj in enumerate(b):
            if bj == (0, 0):
                continue
            acc[i + j] = f2_add(acc[i + j], f2_mul(ai, bj))
    out = list(acc[:6])
    for k in range(6, 11):
        out[k - 6] = f2_add(out[k - 6], f2_mul(acc[k], XI))
    return tuple(out)


def f12_pow(a, e):
    r = F12_ONE
    for bit in bin(e)[2:]:
        r = f12_mul(r, r)
        if bit == "1":
            r = f12_mul(r, a)
    return r


def _line(lam, t, p):
    """the line of slope `lam` (on the twist) through T, evaluated at P in E(Fp) and scaled by w^3:
    (lam*xT - yT) - lam*xP * w^2 + yP * w^3"""
    xp, yp = p
    c0 = f2_sub(f2_mul(lam, t[0]), t[1])
    c2 = f2_neg((lam[0] * xp % P, lam[1] * xp % P))
    return (c0, (0, 0), c2, (yp % P, 0), (0, 0), (0, 0))


def miller_loop(p, q):
    """f_{|x|,Q}(P) for P in G1 (affine ints), Q in G2 (affine Fp2 pairs); identity arguments give 1"""
    if p is None or q is None:
        return F12_ONE
    f = F12_ONE
    t = q
    for bit in bin(ATE_LOOP)[3:]:
        lam = f2_mul(f2_mul((3, 0), f2_mul(t[0], t[0])), f2_inv(f2_mul((2, 0), t[1])))
        f = f12_mul(f12_mul(f, f), _line(lam, t, p))
        t = bls.G2.double(t)
        if bit == "1":
            lam = f2_mul(f2_sub(q[1], t[1]), f2_inv(f2_sub(q[0], t[0])))
            f = f12_mul(f, _line(lam, t, p))
            t = bls.G2.add(t, q)
    return f


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 with a single final exponentiation (verifier.rs:40-52)"""
    f = F12_ONE
    for p, q in pairs:
        f = f12_mul(f, miller_loop(p, q))
    return f12_pow(f, _FINAL_EXP) == F12_ONE


def verify_proof(vk, proof, public_inputs):
    """groth16/src/verifier.rs:23-58.  vk: dict alpha_g1, beta_g2, gamma_g2, delta_g2, ic (affine Python
    points); proof: (a, b, c); public_inputs: ints.  Returns True / False; raises ValueError for a key of
    the wrong size (VerificationError::InvalidVerifyingKey)."""
    if len(public_inputs) + 1 != len(vk["ic"]):
        raise ValueError("InvalidVerifyingKey")
    acc = vk["ic"][0]
    for x, base in zip(public_inputs, vk["ic"][1:]):
        acc = bls.G1.add(acc, bls.G1.mul(base, x % Q))
    a, b, c = proof
    # A*B + inputs*(-gamma) + C*(-delta) = alpha*beta   <=>   the product below is one
    return pairing_product_is_one([
        (a, b),
        (acc, bls.G2.neg(vk["gamma_g2"])),
        (c, bls.G2.neg(vk["delta_g2"])),
        (bls.G1.neg(vk["alpha_g1"]), vk["beta_g2"]),
    ])
