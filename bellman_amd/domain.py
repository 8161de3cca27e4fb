"""`EvaluationDomain` mirror (reference: src/domain.rs:21-190) for the `Scalar<Fr>` instantiation.

Coefficients are numpy uint64 [m,4] Montgomery-form Fr (the bytes of Rust `bls12_381::Scalar`s).
Every transform runs on the GPU through the C ABI; the vector lives in HBM between calls."""

import ctypes

import numpy as np

from . import _lib
from .errors import PolynomialDegreeTooLarge, check

FR_S = 32  # 2-adicity of BLS12-381 Fr (ff::PrimeField::S)


class EvaluationDomain:
    def __init__(self, worker, dev, m, exp):
        self.worker, self._dev, self.m, self.exp = worker, dev, m, exp
        self._lib = _lib.load()

    @classmethod
    def from_coeffs(cls, worker, coeffs):
        """domain.rs:47-79: pad to m = next power of two >= len with zeros; exp >= S errors."""
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        m, exp = 1, 0
        while m < coeffs.shape[0]:
            m *= 2
            exp += 1
            if exp >= FR_S:
                raise PolynomialDegreeTooLarge()
        if coeffs.shape[0] == m:
            padded = coeffs   # already a full domain: no second host copy (8 GiB at 2^28)
        else:
            padded = np.zeros((m, 4), dtype=np.uint64)
            padded[: coeffs.shape[0]] = coeffs
        dev = worker.alloc(m * 32)
        worker.upload(dev, padded)
        return cls(worker, dev, m, exp)

    def __len__(self):
        return self.m

    def into_coeffs(self):
        out = np.empty((self.m, 4), dtype=np.uint64)
        self.worker.download(out, self._dev)
        self.worker.free(self._dev)
        self._dev = None
        return out

    def as_ref(self):
        out = np.empty((self.m, 4), dtype=np.uint64)
        self.worker.download(out, self._dev)
        return out

    def _fft(self, mode):
        check(self._lib.bh_fft_fr_dev(self.worker.ctx, self._dev, self.exp, mode, None), "fft")
        self.worker.synchronize()

    def fft(self, worker=None):
        self._fft(0)

    def ifft(self, worker=None):
        self._fft(1)

    def coset_fft(self, worker=None):
        self._fft(2)

    def icoset_fft(self, worker=None):
        self._fft(3)

    def distribute_powers(self, worker, g_mont):
        g = np.ascontiguousarray(g_mont, dtype=np.uint64).reshape(4)
        check(self._lib.bh_fr_distribute_powers_dev(self.worker.ctx, self._dev, self.m,
                                                    g.ctypes.data_as(ctypes.c_void_p), None))
        self.worker.synchronize()

    def divide_by_z_on_coset(self, worker=None):
        check(self._lib.bh_fr_divide_by_z_on_coset_dev(self.worker.ctx, self._dev, self.exp, None))
        self.worker.synchronize()

    def mul_assign(self, worker, other):
        assert self.m == other.m  # domain.rs:155
        check(self._lib.bh_fr_mul_assign_dev(self.worker.ctx, self._dev, other._dev, self.m, None))
        self.worker.synchronize()

    def sub_assign(self, worker, other):
        assert self.m == other.m  # domain.rs:174
        check(self._lib.bh_fr_sub_assign_dev(self.worker.ctx, self._dev, other._dev, self.m, None))
        self.worker.synchronize()
