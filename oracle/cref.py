"""ctypes front-end of the C oracle (oracle/c -> oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg - never from bellman_amd/.

Data conventions (identical to the product's C ABI, include/bellman_hip.h):
  Fr array   : numpy uint64 [n,4]  little-endian limbs (Montgomery for FFT data,
               canonical for MSM scalars)
  G1 affine  : numpy uint64 [n,12] = x[6] | y[6]   Montgomery; identity = all zero
  G2 affine  : numpy uint64 [n,24] = x.c0 | x.c1 | y.c0 | y.c1
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
MASK64 = (1 << 64) - 1


def build(force=False):
    if force or not os.path.exists(_LIB):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "c")])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB)
        _lib.orc_window_size.restype = ctypes.c_uint
        _lib.orc_window_size.argtypes = [ctypes.c_size_t]
        _lib.orc_max_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


# ---- int <-> limb helpers -------------------------------------------------
def int_to_limbs(x, n):
    return [(x >> (64 * i)) & MASK64 for i in range(n)]


def limbs_to_int(l):
    v = 0
    for i, w in enumerate(l):
        v |= int(w) << (64 * i)
    return v


def ints_to_arr(vals, nl):
    a = np.zeros((len(vals), nl), dtype=np.uint64)
    for i, v in enumerate(vals):
        a[i] = int_to_limbs(v, nl)
    return a


def arr_to_ints(a):
    return [limbs_to_int(row) for row in a]


def fr_to_mont(a):
    out = np.empty_like(a)
    lib().orc_fr_to_mont(_p(out), _p(np.ascontiguousarray(a)), ctypes.c_size_t(a.shape[0]))
    return out


def fr_from_mont(a):
    out = np.empty_like(a)
    lib().orc_fr_from_mont(_p(out), _p(np.ascontiguousarray(a)), ctypes.c_size_t(a.shape[0]))
    return out


def fp_to_mont(a):
    a = np.ascontiguousarray(a).reshape(-1, 6)
    out = np.empty_like(a)
    lib().orc_fp_to_mont(_p(out), _p(a), ctypes.c_size_t(a.shape[0]))
    return out


def fp_from_mont(a):
    a = np.ascontiguousarray(a).reshape(-1, 6)
    out = np.empty_like(a)
    lib().orc_fp_from_mont(_p(out), _p(a), ctypes.c_size_t(a.shape[0]))
    return out


# ---- points: python-int affine <-> Montgomery records ---------------------
def g1_from_py(pts):
    """list of None|(x,y) ints -> [n,12] Montgomery records."""
    canon = np.zeros((len(pts), 12), dtype=np.uint64)
    for i, pt in enumerate(pts):
        if pt is not None:
            canon[i, :6] = int_to_limbs(pt[0], 6)
            canon[i, 6:] = int_to_limbs(pt[1], 6)
    out = fp_to_mont(canon).reshape(-1, 12)
    return out


def g1_to_py(arr):
    arr = np.ascontiguousarray(arr).reshape(-1, 12)
    canon = fp_from_mont(arr).reshape(-1, 12)
    out = []
    for row in canon:
        x, y = limbs_to_int(row[:6]), limbs_to_int(row[6:])
        out.append(None if (x == 0 and y == 0) else (x, y))
    return out


def g2_from_py(pts):
    canon = np.zeros((len(pts), 24), dtype=np.uint64)
    for i, pt in enumerate(pts):
        if pt is not None:
            (x0, x1), (y0, y1) = pt
            canon[i, 0:6] = int_to_limbs(x0, 6)
            canon[i, 6:12] = int_to_limbs(x1, 6)
            canon[i, 12:18] = int_to_limbs(y0, 6)
            canon[i, 18:24] = int_to_limbs(y1, 6)
    return fp_to_mont(canon).reshape(-1, 24)


def g2_to_py(arr):
    arr = np.ascontiguousarray(arr).reshape(-1, 24)
    canon = fp_from_mont(arr).reshape(-1, 24)
    out = []
    for row in canon:
        v = [limbs_to_int(row[6 * k : 6 * k + 6]) for k in range(4)]
        out.append(None if not any(v) else ((v[0], v[1]), (v[2], v[3])))
    return out


def g1_generator():
    from .pyref import bls12_381 as b

    return g1_from_py([b.G1_GEN])[0]


def g2_generator():
    from .pyref import bls12_381 as b

    return g2_from_py([b.G2_GEN])[0]


def gen_bases(group, n, a=1, b=1):
    """P_i = [a + i*b]G (SURVEY §8d synthetic bases)."""
    w = 12 if group == 1 else 24
    out = np.zeros((n, w), dtype=np.uint64)
    gen = g1_generator() if group == 1 else g2_generator()
    a_ = np.array(int_to_limbs(a % Q, 4), dtype=np.uint64)
    b_ = np.array(int_to_limbs(b % Q, 4), dtype=np.uint64)
    fn = lib().orc_g1_gen_bases if group == 1 else lib().orc_g2_gen_bases
    fn(_p(out), ctypes.c_size_t(n), _p(gen), _p(a_), _p(b_))
    return out


def point_add(group, a, b):
    w = 12 if group == 1 else 24
    out = np.zeros(w, dtype=np.uint64)
    fn = lib().orc_g1_add if group == 1 else lib().orc_g2_add
    fn(_p(out), _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)))
    return out


def point_mul(group, a, k):
    w = 12 if group == 1 else 24
    out = np.zeros(w, dtype=np.uint64)
    k_ = np.array(int_to_limbs(k % Q, 4), dtype=np.uint64)
    fn = lib().orc_g1_mul if group == 1 else lib().orc_g2_mul
    fn(_p(out), _p(np.ascontiguousarray(a)), _p(k_))
    return out


def on_curve(group, a):
    fn = lib().orc_g1_on_curve if group == 1 else lib().orc_g2_on_curve
    return bool(fn(_p(np.ascontiguousarray(a))))


def window_size(n):
    return lib().orc_window_size(n)


def density_bitmap(bits):
    """list/array of bools -> LSB0 uint64 words."""
    bits = np.asarray(bits, dtype=np.uint8)
    nwords = (len(bits) + 63) // 64
    padded = np.zeros(nwords * 64, dtype=np.uint8)
    padded[: len(bits)] = bits
    return np.packbits(padded, bitorder="little").view(np.uint64).copy()


def multiexp(group, bases, offset, density, scalars, c=0, threads=0):
    """Restated multiexp (multiexp.rs:305-332).  Returns (rc, affine record).
    density: None (FullDensity) or LSB0 uint64 bitmap."""
    w = 12 if group == 1 else 24
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, w)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(w, dtype=np.uint64)
    fn = lib().orc_multiexp_g1 if group == 1 else lib().orc_multiexp_g2
    fn.restype = ctypes.c_int
    rc = fn(
        _p(bases),
        ctypes.c_size_t(bases.shape[0]),
        ctypes.c_size_t(offset),
        _p(density),
        _p(scalars),
        ctypes.c_size_t(scalars.shape[0]),
        ctypes.c_uint(c),
        ctypes.c_int(threads),
        _p(out),
    )
    return rc, out


def naive_multiexp(group, bases, scalars, threads=0):
    w = 12 if group == 1 else 24
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, w)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    assert bases.shape[0] == scalars.shape[0]
    out = np.zeros(w, dtype=np.uint64)
    fn = lib().orc_naive_multiexp_g1 if group == 1 else lib().orc_naive_multiexp_g2
    fn(_p(bases), _p(scalars), ctypes.c_size_t(scalars.shape[0]), ctypes.c_int(threads), _p(out))
    return out


FFT, IFFT, COSET_FFT, ICOSET_FFT = 0, 1, 2, 3


def fft(data, mode, threads=8):
    """In-place on a copy; data [2^k,4] Montgomery."""
    a = np.ascontiguousarray(data, dtype=np.uint64).copy()
    n = a.shape[0]
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    lib().orc_fft(_p(a), ctypes.c_uint32(log_n), ctypes.c_int(mode), ctypes.c_int(threads))
    return a


def serial_fft(data):
    a = np.ascontiguousarray(data, dtype=np.uint64).copy()
    log_n = a.shape[0].bit_length() - 1
    lib().orc_serial_fft(_p(a), ctypes.c_uint32(log_n))
    return a


def parallel_fft(data, log_cpus, threads=1):
    a = np.ascontiguousarray(data, dtype=np.uint64).copy()
    log_n = a.shape[0].bit_length() - 1
    lib().orc_parallel_fft(_p(a), ctypes.c_uint32(log_n), ctypes.c_uint32(log_cpus), ctypes.c_int(threads))
    return a


def mul_assign(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    lib().orc_mul_assign(_p(a), _p(np.ascontiguousarray(b)), ctypes.c_size_t(a.shape[0]))
    return a


def sub_assign(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    lib().orc_sub_assign(_p(a), _p(np.ascontiguousarray(b)), ctypes.c_size_t(a.shape[0]))
    return a


def divide_by_z_on_coset(a, threads=8):
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    log_n = a.shape[0].bit_length() - 1
    lib().orc_divide_by_z_on_coset(_p(a), ctypes.c_uint32(log_n), ctypes.c_int(threads))
    return a


def h_coeffs(a, b, c, threads=8):
    """prover.rs:221-240 on padded Montgomery evaluation vectors; returns m-1 coeffs."""
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    b = np.ascontiguousarray(b, dtype=np.uint64).copy()
    c = np.ascontiguousarray(c, dtype=np.uint64).copy()
    log_n = a.shape[0].bit_length() - 1
    lib().orc_h_coeffs(_p(a), _p(b), _p(c), ctypes.c_uint32(log_n), ctypes.c_int(threads))
    return a[:-1]


def fr_dot(a, b, threads=0):
    """sum_i a[i]*b[i] mod q (canonical [n,4] inputs) as a Python int - test helper for the MSM identity
    sum_i s_i [t_i]G = [sum_i s_i t_i]G."""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    assert a.shape == b.shape
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_fr_dot(_p(out), _p(a), _p(b), ctypes.c_size_t(a.shape[0]), ctypes.c_int(threads))
    return limbs_to_int(out)


def random_fr(n, seed, canonical_lt_q=True):
    """n pseudo-random values < q as [n,4] uint64 (top limb clamped below q's top limb)."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(
        0, 2, size=(n, 4), dtype=np.uint64
    )
    a[:, 3] %= np.uint64(0x73EDA753299D7D48)
    return a
