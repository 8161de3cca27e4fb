"""BLS12-381 field and curve arithmetic on Python big ints (AFFINE formulas).

Restates the published parameters of the third-party crate `bls12_381 0.8.0`
(pinned in /root/reference/Cargo.lock:105-108; source absent).  Deliberately
uses textbook affine chord-and-tangent formulas so that it is independent of
the projective formulas used by oracle/c and by the HIP kernels.

Call sites in the reference that reach this arithmetic (through traits):
  src/multiexp.rs:39,63,230,236,271-274,299   point add / double / identity
  src/multiexp.rs:174,179                     scalar zero test, to_le_bits
  src/domain.rs:63-77,105,130,140,250-258     Fr mul/add/sub/pow/invert/constants
  groth16/src/prover.rs:326-360               scalar mul, to_affine
"""

# ---- scalar field Fr ------------------------------------------------------
Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
FR_NUM_BITS = 255
FR_S = 32
FR_GENERATOR = 7
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (Q - 1) >> FR_S, Q)
FR_R = (1 << 256) % Q  # Montgomery radix (4x64 limbs)

# ---- base field Fp --------------------------------------------------------
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
FP_R = (1 << 384) % P

BLS_X = -0xD201000000010000  # curve parameter

G1_B = 4
G1_GEN = (
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
)
# Fp2 elements are (c0, c1) = c0 + c1*u, u^2 = -1
G2_B = (4, 4)
G2_GEN = (
    (
        0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
        0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
    ),
    (
        0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
        0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE,
    ),
)


# ---- Fp2 ------------------------------------------------------------------
def fp2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def fp2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def fp2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def fp2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def fp2_inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % P, P - 2, P)
    return ((a[0] * n) % P, (-a[1] * n) % P)


class _FpOps:
    """Field-op bundle for G1 (coordinates in Fp)."""

    zero = 0
    one = 1
    b = G1_B

    add = staticmethod(lambda a, b: (a + b) % P)
    sub = staticmethod(lambda a, b: (a - b) % P)
    mul = staticmethod(lambda a, b: (a * b) % P)
    neg = staticmethod(lambda a: (-a) % P)
    inv = staticmethod(lambda a: pow(a, P - 2, P))
    small = staticmethod(lambda k, a: (k * a) % P)


class _Fp2Ops:
    """Field-op bundle for G2 (coordinates in Fp2)."""

    zero = (0, 0)
    one = (1, 0)
    b = G2_B

    add = staticmethod(fp2_add)
    sub = staticmethod(fp2_sub)
    mul = staticmethod(fp2_mul)
    neg = staticmethod(fp2_neg)
    inv = staticmethod(fp2_inv)
    small = staticmethod(lambda k, a: ((k * a[0]) % P, (k * a[1]) % P))


class AffineCurve:
    """y^2 = x^3 + b, short Weierstrass a = 0.  Points: None = identity or (x, y)."""

    def __init__(self, ops, gen, name):
        self.F = ops
        self.gen = gen
        self.name = name

    def identity(self):
        return None

    def is_identity(self, pt):
        return pt is None

    def on_curve(self, pt):
        if pt is None:
            return True
        F = self.F
        x, y = pt
        return F.mul(y, y) == F.add(F.mul(F.mul(x, x), x), F.b)

    def neg(self, pt):
        if pt is None:
            return None
        return (pt[0], self.F.neg(pt[1]))

    def double(self, pt):
        if pt is None:
            return None
        F = self.F
        x, y = pt
        if y == F.zero:
            return None
        lam = F.mul(F.small(3, F.mul(x, x)), F.inv(F.small(2, y)))
        x3 = F.sub(F.mul(lam, lam), F.small(2, x))
        y3 = F.sub(F.mul(lam, F.sub(x, x3)), y)
        return (x3, y3)

    def add(self, p1, p2):
        if p1 is None:
            return p2
        if p2 is None:
            return p1
        F = self.F
        x1, y1 = p1
        x2, y2 = p2
        if x1 == x2:
            if y1 == y2:
                return self.double(p1)
            return None
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
        x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
        y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
        return (x3, y3)

    def mul(self, pt, k):
        """[k]pt for an integer k >= 0 (scalars are reduced mod Q by callers)."""
        acc = None
        for bit in bin(k)[2:] if k else "":
            acc = self.double(acc)
            if bit == "1":
                acc = self.add(acc, pt)
        return acc

    def eq(self, p1, p2):
        return p1 == p2


G1 = AffineCurve(_FpOps, G1_GEN, "G1")
G2 = AffineCurve(_Fp2Ops, G2_GEN, "G2")


# ---- Zcash serialisation (groth16/src/lib.rs:38-46, 143-156, 258-287) -----
def _fp_lex_largest(y):
    return y > (P - 1) // 2


def g1_compress(pt):
    if pt is None:
        out = bytearray(48)
        out[0] = 0xC0
        return bytes(out)
    x, y = pt
    out = bytearray(x.to_bytes(48, "big"))
    out[0] |= 0x80
    if _fp_lex_largest(y):
        out[0] |= 0x20
    return bytes(out)


def g1_uncompressed(pt):
    if pt is None:
        out = bytearray(96)
        out[0] = 0x40
        return bytes(out)
    return pt[0].to_bytes(48, "big") + pt[1].to_bytes(48, "big")


def _fp2_lex_largest(y):
    # compare c1 first, then c0
    if y[1] != 0:
        return _fp_lex_largest(y[1])
    return _fp_lex_largest(y[0])


def g2_compress(pt):
    if pt is None:
        out = bytearray(96)
        out[0] = 0xC0
        return bytes(out)
    x, y = pt
    out = bytearray(x[1].to_bytes(48, "big") + x[0].to_bytes(48, "big"))
    out[0] |= 0x80
    if _fp2_lex_largest(y):
        out[0] |= 0x20
    return bytes(out)


def g2_uncompressed(pt):
    if pt is None:
        out = bytearray(192)
        out[0] = 0x40
        return bytes(out)
    x, y = pt
    return (
        x[1].to_bytes(48, "big")
        + x[0].to_bytes(48, "big")
        + y[1].to_bytes(48, "big")
        + y[0].to_bytes(48, "big")
    )
