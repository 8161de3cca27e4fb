"""Engines: the pairing-engine abstraction the restated code is generic over.

An engine bundles a scalar field (modulus r, 2-adicity S, multiplicative
generator, 2^S-th root of unity, NUM_BITS) and two groups G1, G2 exposing
identity/is_identity/add/double/mul/eq.  Scalars are Python ints in [0, r).

DummyEngine restates /root/reference/groth16/src/tests/dummy_engine.rs:
  :15        MODULUS_R = 64513
  :294-296   NUM_BITS = 16, S = 10
  :317-319   MULTIPLICATIVE_GENERATOR = 5, ROOT_OF_UNITY = 57751
  :335-378   G1 = G2 = Gt = Fr, group law = field addition, scalar mul = field mul
"""

from . import bls12_381 as bls


class ScalarField:
    def __init__(self, modulus, num_bits, s, generator, root_of_unity):
        self.r = modulus
        self.NUM_BITS = num_bits
        self.S = s
        self.MULTIPLICATIVE_GENERATOR = generator
        self.ROOT_OF_UNITY = root_of_unity

    def inv(self, a):
        assert a % self.r != 0
        return pow(a, self.r - 2, self.r)


class _AdditiveFieldGroup:
    """DummyEngine group: elements of F_r under addition (identity = 0)."""

    def __init__(self, r):
        self.r = r
        self.gen = 1

    def identity(self):
        return 0

    def is_identity(self, a):
        return a == 0

    def add(self, a, b):
        return (a + b) % self.r

    def double(self, a):
        return (2 * a) % self.r

    def neg(self, a):
        return (-a) % self.r

    def mul(self, a, k):
        return (a * k) % self.r

    def eq(self, a, b):
        return a == b


class Engine:
    def __init__(self, name, fr, g1, g2):
        self.name = name
        self.Fr = fr
        self.G1 = g1
        self.G2 = g2


DummyEngine = Engine(
    "dummy",
    ScalarField(64513, 16, 10, 5, 57751),
    _AdditiveFieldGroup(64513),
    _AdditiveFieldGroup(64513),
)

Bls12 = Engine(
    "bls12_381",
    ScalarField(bls.Q, bls.FR_NUM_BITS, bls.FR_S, bls.FR_GENERATOR, bls.FR_ROOT_OF_UNITY),
    bls.G1,
    bls.G2,
)
