"""TEST INFRASTRUCTURE: a BLS12-381 engine for the generic oracle/pyref code whose group law is
executed by the C oracle (oracle/c) instead of Python big ints, so that the restated
generator/prover run at realistic sizes.  Points are immutable `bytes` records in the library
format (96/192-byte Montgomery affine, all-zero = identity)."""

import numpy as np

from . import cref
from .pyref import bls12_381 as bls
from .pyref.engines import Engine, ScalarField


class CGroup:
    def __init__(self, group):
        self.group = group
        self.words = 12 if group == 1 else 24
        self._zero = bytes(self.words * 8)
        gen = cref.g1_generator() if group == 1 else cref.g2_generator()
        self.gen = np.ascontiguousarray(gen).tobytes()

    def _arr(self, p):
        return np.frombuffer(p, dtype=np.uint64)

    def identity(self):
        return self._zero

    def is_identity(self, p):
        return p == self._zero

    def add(self, a, b):
        return cref.point_add(self.group, self._arr(a), self._arr(b)).tobytes()

    def double(self, a):
        return self.add(a, a)

    def mul(self, a, k):
        return cref.point_mul(self.group, self._arr(a), k % bls.Q).tobytes()

    def eq(self, a, b):
        return a == b

    def to_array(self, pts):
        return np.frombuffer(b"".join(pts), dtype=np.uint64).reshape(len(pts), self.words).copy() if pts else np.zeros((0, self.words), dtype=np.uint64)


CBls12 = Engine(
    "bls12_381-c",
    ScalarField(bls.Q, bls.FR_NUM_BITS, bls.FR_S, bls.FR_GENERATOR, bls.FR_ROOT_OF_UNITY),
    CGroup(1),
    CGroup(2),
)
