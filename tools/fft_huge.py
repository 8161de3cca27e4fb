"""EvaluationDomain at 2^29 ... 2^31 points (src/domain.rs:57-59 allows up to 2^31), everything generated and checked on
the device side so that no 64 GiB host array is needed:

  python tools/fft_huge.py <log_n> [<log_n> ...]

  * a SPARSE polynomial (coefficients at low, middle and top indices, uploaded 32 bytes at a time into a zeroed vector):
    fft and coset_fft outputs at sampled positions == sum_j a_j s^(i_j) w^(i_j k) with Python integers - wrong index
    arithmetic above 2^28 (32-bit products, the three-pass digit reversal) cannot survive this;
  * dense data x_i = 3 * 5^i (bh_fr_powers_dev): ifft(fft(x)) - x and icoset_fft(coset_fft(x)) - x are downloaded in
    256 MiB pieces and must be all zero.
The method is tests/test_gpu_scale.py::test_fft_above_2_25's; memory: three vectors of 2^log_n x 32 B (192 GiB at 2^31)."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bellman_amd  # noqa: E402
from bellman_amd import _lib  # noqa: E402
from bellman_amd.groth16 import fr_to_mont_array  # noqa: E402

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R = 1 << 256


def from_mont(row):
    v = sum(int(x) << (64 * i) for i, x in enumerate(row))
    return v * pow(R, -1, Q) % Q


def at(ptr, byte_off):
    return ctypes.c_void_p(ptr.value + byte_off)


def run(log_n):
    lib = _lib.load()
    w = bellman_amd.Worker(0)
    n = 1 << log_n
    nbytes = n * 32
    rnd = np.random.default_rng(log_n)
    omega = pow(pow(7, (Q - 1) >> 32, Q), 1 << (32 - log_n), Q)
    pos = sorted(set([0, 1, 2047, 2048, (1 << 22) + 5, (1 << 28) + 77, n // 2 - 1, n // 2, n - 2049, n - 1] +
                     [int(x) for x in rnd.integers(0, n, 7)]))
    coef = [int(x) for x in rnd.integers(1, 1 << 62, len(pos))]
    cm = fr_to_mont_array(coef)
    ks = [0, 1, n - 1, n // 2, (1 << 25) + 3, (1 << 29) - 5] + [int(x) for x in rnd.integers(0, n, 40)]
    ks = [k % n for k in ks]
    d = w.alloc(nbytes)
    t0 = time.time()
    for mode, name, shift in ((0, "fft", 1), (2, "coset_fft", 7)):
        assert lib.bh_dev_zero(w.ctx, d, nbytes) == 0
        for i, p in enumerate(pos):
            w.upload(at(d, p * 32), cm[i:i + 1])
        t1 = time.time()
        assert lib.bh_fft_fr_dev(w.ctx, d, log_n, mode, None) == 0
        w.synchronize()
        ms = (time.time() - t1) * 1e3
        for k in ks:
            o = np.zeros((1, 4), dtype=np.uint64)
            w.download(o, at(d, k * 32))
            want = sum(c * pow(shift, i, Q) % Q * pow(omega, (i * k) % n, Q) for c, i in zip(coef, pos)) % Q
            assert from_mont(o[0]) == want, (log_n, name, k)
        print("2^%d %-9s sparse polynomial: %d sampled outputs == Python integers (%.0f ms incl. scratch allocation)" %
              (log_n, name, len(ks), ms), flush=True)
    # dense round trips
    copy = w.alloc(nbytes)
    g, sc = fr_to_mont_array([5]), fr_to_mont_array([3])
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    for fwd, inv, name in ((0, 1, "ifft(fft(x))"), (2, 3, "icoset_fft(coset_fft(x))")):
        assert lib.bh_fr_powers_dev(w.ctx, d, n, p(g), p(sc), None) == 0
        assert lib.bh_fr_powers_dev(w.ctx, copy, n, p(g), p(sc), None) == 0
        w.synchronize()
        t1 = time.time()
        assert lib.bh_fft_fr_dev(w.ctx, d, log_n, fwd, None) == 0
        assert lib.bh_fft_fr_dev(w.ctx, d, log_n, inv, None) == 0
        w.synchronize()
        ms = (time.time() - t1) * 1e3
        assert lib.bh_fr_sub_assign_dev(w.ctx, d, copy, n, None) == 0
        w.synchronize()
        piece = np.zeros((1 << 23, 4), dtype=np.uint64)   # 256 MiB
        for off in range(0, nbytes, piece.nbytes):
            m = min(piece.nbytes, nbytes - off) // 32
            w.download(piece[:m], at(d, off))
            assert not piece[:m].any(), (log_n, name, off)
        print("2^%d %s == x on all %d elements (two transforms %.0f ms)" % (log_n, name, n, ms), flush=True)
    w.free(copy)
    w.free(d)
    w.trim()
    w.close()
    print("2^%d done in %.0f s" % (log_n, time.time() - t0), flush=True)


if __name__ == "__main__":
    for a in sys.argv[1:]:
        run(int(a))
