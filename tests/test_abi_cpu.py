"""CPU-only checks of the product library: it builds, loads, exports every symbol declared in
include/bellman_hip.h, refuses to run without a GPU (no CPU fallback), and its arithmetic
headers (compiled for the host) agree with the oracle.  No device compute here."""

import ctypes
import os
import random
import re

import numpy as np
import pytest

from bellman_amd import _lib
from oracle import cref
from oracle.pyref import bls12_381 as bls

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _exported_c_symbols(path):
    """unmangled function symbols a shared library defines and exports"""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[2] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] == "T" and not ln.split()[2].startswith("_")}


def _all_dynamic_symbols(path):
    """every symbol a shared library DEFINES in its dynamic symbol table, whatever its type (functions, objects, weak C++)"""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if ln.split()}


def test_header_symbols_all_exported(lib):
    """[r4] libbellman_hip.so exports EXACTLY the C symbols include/bellman_hip.h declares; the test hooks and the
    built-in demo circuits (include/bellman_hip_test.h) live in libbellman_hip_test.so, which links against it.
    [r6] ... and NOTHING else: no mangled C++ internal is in any library's dynamic symbol table (csrc/libbellman_hip.map);
    the test library gets the C++ host API from the static archive libbellman_groth16.a and everything else through the C ABI."""
    hdr = open(os.path.join(ROOT, "include", "bellman_hip.h")).read()
    declared = set(re.findall(r"\b(bh_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    # the product boundary carries no test hooks and no demo circuits: they live in their own header and library
    assert not [d for d in declared if d.startswith("bh_test_") or "demo" in d]
    test_hdr = open(os.path.join(ROOT, "include", "bellman_hip_test.h")).read()
    hooks = set(re.findall(r"\b(bh_[a-z0-9_]+)\s*\(", test_hdr))
    assert hooks and all(h.startswith("bh_test_") or "demo" in h for h in hooks)
    assert declared == set(_lib.EXPORTS) and hooks == set(_lib.TEST_EXPORTS)
    assert _exported_c_symbols(_lib.LIB_PATH) == declared
    assert _exported_c_symbols(_lib.TEST_LIB_PATH) == hooks
    assert _all_dynamic_symbols(_lib.LIB_PATH) == declared, sorted(_all_dynamic_symbols(_lib.LIB_PATH) - declared)[:10]
    assert _all_dynamic_symbols(_lib.TEST_LIB_PATH) == hooks
    assert _all_dynamic_symbols(_lib.DEMO_LIB_PATH) == set(_lib.DEMO_EXPORTS)
    # the test and demo libraries import nothing from the product but its C ABI
    import subprocess
    for path in (_lib.TEST_LIB_PATH, _lib.DEMO_LIB_PATH):
        und = subprocess.run(["nm", "-D", "--undefined-only", "-C", path], capture_output=True, text=True, check=True).stdout
        internals = [ln for ln in und.splitlines() if " bh::" in ln or " groth16::" in ln or " bellman::" in ln]
        assert not internals, internals[:5]
    for sym in declared:
        assert hasattr(lib.product, sym), sym
    for sym in hooks:
        assert hasattr(lib.test, sym) and not hasattr(lib.product, sym), sym
    assert b"gfx950" in lib.bh_version()


def test_library_override_needs_two_variables():
    """BELLMAN_HIP_LIB alone does not swap the library under the tests or the bench (bellman_amd/_lib.py)."""
    import subprocess
    import sys

    code = "from bellman_amd import _lib; print(_lib.LIB_PATH)"
    env = dict(os.environ, BELLMAN_HIP_LIB="/nonexistent/libbellman_hip.so")
    env.pop("BELLMAN_HIP_ALLOW_LIB_OVERRIDE", None)
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, check=True).stdout
    assert out.strip() == os.path.join(ROOT, "bellman_amd", "lib", "libbellman_hip.so")
    env["BELLMAN_HIP_ALLOW_LIB_OVERRIDE"] = "1"
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, check=True).stdout
    assert out.strip() == "/nonexistent/libbellman_hip.so"


def test_no_cpu_fallback_without_gpu(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = ctypes.c_void_p()
    rc = lib.bh_ctx_create(0, ctypes.byref(ctx))
    assert rc == -3  # BH_ERR_NO_DEVICE
    from bellman_amd import BellmanHipError, Worker

    with pytest.raises(BellmanHipError):
        Worker()


def test_product_never_imports_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bellman_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|#include\s+\"[^\"]*oracle", txt):
                    bad.append(f)
    assert not bad, bad


def test_host_field_mul_matches_oracle(lib):
    a, b = cref.random_fr(500, 1), cref.random_fr(500, 2)
    r = np.zeros_like(a)
    lib.bh_test_fr_mul_host(_p(r), _p(a), _p(b), 500)
    assert np.array_equal(r, cref.mul_assign(a, b))
    a[0], b[1] = 0, 0
    a[2] = b[2] = cref.ints_to_arr([bls.Q - 1], 4)[0]
    r2 = np.zeros_like(a)
    lib.bh_test_fr_mul_bform_host(_p(r2), _p(a), _p(b), 500)   # the FFT's multiplier: b pre-sliced (B form)
    assert np.array_equal(r2, cref.mul_assign(a, b))
    rnd = random.Random(3)
    xs = [rnd.randrange(bls.P) for _ in range(100)] + [0, 1, bls.P - 1]
    ys = [rnd.randrange(bls.P) for _ in range(100)] + [bls.P - 1, bls.P - 1, bls.P - 1]
    xa, ya = cref.ints_to_arr(xs, 6), cref.ints_to_arr(ys, 6)
    ra = np.zeros_like(xa)
    lib.bh_test_fp_mul_host(_p(ra), _p(xa), _p(ya), len(xs))
    rinv = pow(pow(2, 384, bls.P), -1, bls.P)
    assert cref.arr_to_ints(ra) == [x * y * rinv % bls.P for x, y in zip(xs, ys)]


@pytest.mark.parametrize("group", [1, 2])
def test_host_group_law_matches_oracle(lib, group):
    n = 10
    w = 12 if group == 1 else 24
    A = cref.gen_bases(group, n, a=5, b=3)
    B = cref.gen_bases(group, n, a=7, b=11)
    B[0] = A[0]  # doubling through the addition path
    B[1] = 0  # + identity
    A[2] = 0  # identity +
    B[3] = cref.point_mul(group, A[3], bls.Q - 1)  # P + (-P)
    out = np.zeros((n, w), dtype=np.uint64)
    lib.bh_test_point_add_host(group, _p(out), _p(A), _p(B), n)
    want = np.stack([cref.point_add(group, A[i], B[i]) for i in range(n)])
    assert np.array_equal(out, want)
    assert not out[3].any()
    for k in (0, 1, 2, bls.Q - 1, random.Random(4).randrange(bls.Q)):
        ka = np.array(cref.int_to_limbs(k, 4), dtype=np.uint64)
        o = np.zeros(w, dtype=np.uint64)
        lib.bh_test_point_mul_host(group, _p(o), _p(A[4]), _p(ka))
        assert np.array_equal(o, cref.point_mul(group, A[4], k))


def test_host_fr_inverse_matches_oracle(lib):
    """Host-side domain constants (minv, zinv, geninv: domain.rs:75-78,139-140) come from fe_inv.
    Regression for the first GPU run: q-2 must borrow across 32-bit limbs (q's low limb is 1)."""
    a = cref.random_fr(20, 9)
    a[0] = cref.fr_to_mont(cref.ints_to_arr([2], 4))[0]
    r = np.zeros_like(a)
    lib.bh_test_fr_inv_host(_p(r), _p(a), 20)
    want = np.zeros_like(a)
    for i in range(20):
        cref.lib().orc_fr_inv(_p(want[i : i + 1]), _p(np.ascontiguousarray(a[i : i + 1])))
    assert np.array_equal(r, want)


@pytest.mark.parametrize("group", [1, 2])
def test_fast_host_group_ops_match_oracle(lib, group):
    """bh_point_add / bh_point_mul: the 64-bit-limb host arithmetic used for serial tails."""
    n = 8
    w = 12 if group == 1 else 24
    A = cref.gen_bases(group, n, a=15, b=4)
    B = cref.gen_bases(group, n, a=2, b=9)
    B[0] = A[0]
    B[1] = 0
    A[2] = 0
    B[3] = cref.point_mul(group, A[3], bls.Q - 1)
    out = np.zeros((n, w), dtype=np.uint64)
    lib.bh_point_add(group, _p(out), _p(A), _p(B), n)
    assert np.array_equal(out, np.stack([cref.point_add(group, A[i], B[i]) for i in range(n)]))
    for k in (0, 1, 5, bls.Q - 1, random.Random(8).randrange(bls.Q)):
        ka = np.array(cref.int_to_limbs(k, 4), dtype=np.uint64)
        o = np.zeros(w, dtype=np.uint64)
        lib.bh_point_mul(group, _p(o), _p(A[5]), _p(ka))
        assert np.array_equal(o, cref.point_mul(group, A[5], k))


@pytest.mark.parametrize("group", [1, 2])
def test_host_linear_combination_matches_oracle(lib, group):
    """bh_point_lincomb (the tail of create_proof, prover.rs:339-354, in one shared doubling chain): every kind of term -
    scalar 0, 1, small, q - 1, random; an identity point; P and -P cancelling; all-ones (scalars = NULL); empty."""
    w = 12 if group == 1 else 24
    rnd = random.Random(77 + group)
    P = cref.gen_bases(group, 9, a=21, b=6)
    P[6] = 0                                            # the identity as a term
    P[8] = cref.point_mul(group, P[7], bls.Q - 1)       # -P[7]
    ks = [0, 1, 5, bls.Q - 1, rnd.randrange(bls.Q), rnd.randrange(bls.Q), rnd.randrange(bls.Q), 9, 9]
    ka = np.array([cref.int_to_limbs(k, 4) for k in ks], dtype=np.uint64)
    want = np.zeros(w, dtype=np.uint64)
    for i, k in enumerate(ks):
        want = cref.point_add(group, want, cref.point_mul(group, P[i], k))
    out = np.ones(w, dtype=np.uint64)
    lib.bh_point_lincomb(group, _p(out), _p(P), _p(ka), len(ks))
    assert np.array_equal(out, want)
    plain = np.zeros(w, dtype=np.uint64)
    for i in range(len(ks)):
        plain = cref.point_add(group, plain, P[i])
    lib.bh_point_lincomb(group, _p(out), _p(P), None, len(ks))
    assert np.array_equal(out, plain)
    lib.bh_point_lincomb(group, _p(out), _p(P), _p(ka), 0)
    assert not out.any()
    lib.bh_point_lincomb(group, _p(out), _p(P[7:9]), _p(ka[7:9]), 2)   # 9 P - 9 P
    assert not out.any()


def test_host_group_ops_accept_unaligned_buffers(lib):
    """Regression (first GPU prover run): caller records are only 8-byte aligned in general."""
    A = cref.gen_bases(1, 2, a=3, b=1)
    raw = np.zeros(12 * 3 + 1, dtype=np.uint64)
    base = raw.ctypes.data
    off = 1 if (base % 16 == 0) else 0           # force an address that is 8 mod 16
    a = raw[off : off + 12]
    b = raw[off + 12 : off + 24]
    r = raw[off + 24 : off + 36]
    a[:] = A[0]
    b[:] = A[1]
    assert a.ctypes.data % 16 == 8
    lib.bh_point_add(1, _p(r), _p(a), _p(b), 1)
    assert np.array_equal(r, cref.point_add(1, A[0], A[1]))
    k = np.array(cref.int_to_limbs(12345, 4), dtype=np.uint64)
    lib.bh_point_mul(1, _p(r), _p(a), _p(k))
    assert np.array_equal(r, cref.point_mul(1, A[0], 12345))


def test_fr_wide_reduction_host():
    """Fr::from_u512 of the C++ mirror (create_random_proof's sampling, prover.rs:176-177): host code"""
    import ctypes
    import random

    import numpy as np

    from bellman_amd import _lib

    lib = _lib.load()
    q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    rnd = random.Random(3)
    for v in [0, 1, q - 1, q, q + 1, (1 << 256) - 1, 1 << 256, (1 << 512) - 1] + [rnd.getrandbits(512) for _ in range(50)]:
        limbs = np.array([(v >> (64 * i)) & ((1 << 64) - 1) for i in range(8)], dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        lib.bh_test_fr_from_u512_host(out.ctypes.data_as(ctypes.c_void_p), limbs.ctypes.data_as(ctypes.c_void_p))
        got = sum(int(x) << (64 * i) for i, x in enumerate(out))
        assert got == (v % q) * (1 << 256) % q


def test_mirror_scalar_field_arithmetic_host():
    """bellman::Fr of the C++ mirror (csrc/groth16.hpp): the arithmetic circuits and the linear-combination evaluation use
    during synthesis - branch-free add / sub, the mulx / adcx / adox Montgomery product (first operand below q, second any
    256-bit value), Fr::from_u64 by Barrett reduction (quotient estimates short by 0 ... 3 must all be corrected),
    to_canonical, inversion - against Python integers.  Host code, no device."""
    import ctypes
    import random

    import numpy as np

    from bellman_amd import _lib

    lib = _lib.load()
    q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    R = 1 << 256
    Rinv = pow(R, -1, q)
    M64 = (1 << 64) - 1

    def arr(vals):
        return np.array([[(v >> (64 * i)) & M64 for i in range(4)] for v in vals], dtype=np.uint64)

    def ints(a):
        return [sum(int(x) << (64 * i) for i, x in enumerate(row)) for row in a]

    def run(op, a, b=None):
        n = len(a)
        out = np.zeros((n, 4), dtype=np.uint64)
        aa = arr(a)
        bb = arr(b) if b is not None else None
        lib.bh_test_fr_ops_host(op, out.ctypes.data_as(ctypes.c_void_p), aa.ctypes.data_as(ctypes.c_void_p),
                                bb.ctypes.data_as(ctypes.c_void_p) if bb is not None else None, n)
        return ints(out)

    rnd = random.Random(11)
    edge = [0, 1, 2, q - 1, q - 2, R % q, (R % q) - 1, (q + 1) // 2, (q - 1) // 2, M64, 1 << 64, (1 << 255) % q, q >> 1, q >> 64]
    xs = edge + [rnd.randrange(q) for _ in range(3000)]
    # every edge value against every edge value, then random pairs
    a = [x for x in edge for _ in edge] + xs
    b = [y for _ in edge for y in edge] + [rnd.randrange(q) for _ in xs]
    assert run(0, a, b) == [(x + y) % q for x, y in zip(a, b)]
    assert run(1, a, b) == [(x - y) % q for x, y in zip(a, b)]
    assert run(2, a, b) == [x * y * Rinv % q for x, y in zip(a, b)]
    assert run(3, a) == [(-x) % q for x in a]
    # the second operand of a product need not be reduced (Fr::from_u512 multiplies R^2 by raw 256-bit words)
    wide = [R - 1, R - 2, q, q + 1, 2 * q - 1, 2 * q, 2 * q + 1, (1 << 255), (1 << 255) - 1] + [rnd.getrandbits(256) for _ in range(2000)]
    lhs = [rnd.choice(edge + [rnd.randrange(q)]) for _ in wide]
    assert run(2, lhs, wide) == [x * y * Rinv % q for x, y in zip(lhs, wide)]
    # from_u64: v * R mod q, incl. the values around the multiples of q / R where the quotient estimate is tightest
    vs = [0, 1, 2, M64, M64 - 1, 1 << 63, (1 << 63) - 1, (1 << 32), (1 << 32) - 1, 0xFFFFFFFF00000001, 0x73EDA753299D7D48]
    for k in range(1, 200):   # v with v * (R mod q) just below / above a multiple of q
        t = k * q // (R % q)
        vs += [v for v in (t - 1, t, t + 1, t + 2) if 0 <= v <= M64]
    vs += [rnd.getrandbits(64) for _ in range(20000)] + [rnd.getrandbits(rnd.randrange(1, 65)) for _ in range(5000)]
    assert run(4, vs) == [v * R % q for v in vs]
    mont = [x * R % q for x in xs]
    assert run(5, mont) == xs
    inv_in = [x for x in mont if x][:300]
    assert run(6, inv_in) == [pow(x * Rinv % q, -1, q) * R % q for x in inv_in]


def test_lazily_reduced_fp_helpers_host():
    """The curve kernels keep Fp values in [0, 2p) (ff.cuh fpl_*): every helper must preserve the range,
    agree with the integers mod p, and treat both representatives of zero (0 and p) as zero."""
    import ctypes
    import random

    import numpy as np

    from bellman_amd import _lib

    lib = _lib.load()
    p = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    R = 1 << 384

    def arr(v):
        return np.array([(v >> (64 * i)) & ((1 << 64) - 1) for i in range(6)], dtype=np.uint64)

    def val(a):
        return sum(int(x) << (64 * i) for i, x in enumerate(a))

    def call(op, a, b=None):
        out = np.zeros(6, dtype=np.uint64)
        xa, xb = arr(a), (arr(b) if b is not None else None)
        flag = lib.bh_test_fp_lazy_host(op, out.ctypes.data_as(ctypes.c_void_p), xa.ctypes.data_as(ctypes.c_void_p),
                                        xb.ctypes.data_as(ctypes.c_void_p) if xb is not None else None)
        return val(out), flag

    rnd = random.Random(11)
    edge = [0, 1, p - 1, p, p + 1, 2 * p - 1]
    vals = edge + [rnd.randrange(2 * p) for _ in range(40)]
    for a in vals:
        r, _ = call(2, a)
        assert r < 2 * p and (r + a) % p == 0 and (a != 0 or r == 0)
        r, _ = call(3, a)
        assert r == a % p
        assert call(4, a)[1] == (1 if a % p == 0 else 0)
        r, _ = call(6, a)
        assert r < 2 * p and (r * R - a * a) % p == 0          # Montgomery: a*a/R
        for b in edge + [rnd.randrange(2 * p) for _ in range(6)]:
            r, _ = call(0, a, b)
            assert r < 2 * p and (r - a - b) % p == 0
            r, _ = call(1, a, b)
            assert r < 2 * p and (r - a + b) % p == 0
            r, _ = call(5, a, b)
            assert r < 2 * p and (r * R - a * b) % p == 0
            assert call(7, a, b)[1] == (1 if (a - b) % p == 0 else 0)


def test_msm_plan_invariants_host():
    """make_plan (msm_stages.hip), host only: the windows cover 256 bits, signed digits need 2^(c-1) buckets,
    the 2-D reduction splits the bucket index exactly, a typical bucket never spans many chunks."""
    import ctypes

    import numpy as np

    from bellman_amd import _lib

    lib = _lib.load()
    out = np.zeros(9, dtype=np.uint32)
    for group in (1, 2):
        for lg in range(0, 27):
            for n in {1 << lg, (1 << lg) + 1, max(1, (1 << lg) - 1)}:
                for forced in (0, 2, 7, 13, 16, 24):
                    assert lib.bh_msm_plan_info(n, group, forced, out.ctypes.data_as(ctypes.c_void_p)) == 0
                    c_, W, nb, K, cpw, passes, lo, hi, pairs = (int(x) for x in out)
                    assert 2 <= c_ <= 24 and (forced == 0 or c_ == forced)
                    assert W == -(-256 // c_) and W * c_ >= 256 and nb == 1 << (c_ - 1)
                    assert lo + hi == c_ - 1 and passes == -(-c_ // 8)
                    assert K >= 1 and cpw == -(-n // K) and K >= (n >> (c_ - 1))
                    assert pairs == (W * n) & 0xFFFFFFFF


def test_proof_slices_partition_the_scalar_range_host():
    """slice_of (groth16_prover.cpp), host only: the parts of a sharded proof tile [0, n) without gaps or
    overlap and every interior cut is a multiple of 64 (so a density bitmap slice starts on a word)."""
    import ctypes
    import random

    from bellman_amd import _lib

    lib = _lib.load()
    rnd = random.Random(5)
    for n in [0, 1, 2, 63, 64, 65, 1000, (1 << 20) - 3, (1 << 24) + 17] + [rnd.randrange(1 << 26) for _ in range(20)]:
        for parts in (1, 2, 3, 4, 7, 8, 64, 1000):
            prev = 0
            for part in range(parts):
                lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
                lib.bh_test_proof_slice(n, part, parts, ctypes.byref(lo), ctypes.byref(hi))
                assert lo.value == prev and lo.value <= hi.value <= n
                assert hi.value % 64 == 0 or hi.value == n
                prev = hi.value
            assert prev == n


def test_montgomery_reduction_whole_columns_with_maximal_limbs(lib):
    """ff.cuh adds the product column c[k] WHOLE wherever Radix30::NOSPLIT proves that the column sum stays below 2^64
    (all 18 columns of Fr, 21 of 26 of Fp).  Operands whose 30-bit limbs are all ones drive every column to the
    bound the compile-time proof uses; results against big-integer arithmetic (multiexp.rs / domain.rs rely on exact
    field products everywhere)."""
    p, q = bls.P, bls.Q
    R = 1 << 384

    def arr(v, n):
        return np.array([(v >> (64 * i)) & ((1 << 64) - 1) for i in range(n)], dtype=np.uint64)

    def val(a):
        return sum(int(x) << (64 * i) for i, x in enumerate(a))

    def lazy(op, a, b=None):
        out = np.zeros(6, dtype=np.uint64)
        xa, xb = arr(a, 6), (arr(b, 6) if b is not None else None)
        lib.bh_test_fp_lazy_host(op, _p(out), _p(xa), _p(xb) if xb is not None else None)
        return val(out)

    ones = (1 << 384) - 1
    rnd = random.Random(77)
    big = [ones, ones >> 1, (1 << 383) - 1, (1 << 382) - 1, 2 * p - 1, 2 * p - 2, p - 1, p, p + 1,
           int("3fffffff" * 13, 16) & ones, (1 << 381) | ((1 << 381) - 1)]
    # the result of a Montgomery product is < a*b*2^6 / 2^390 + p: pairs whose result fits 384 bits
    for a in big + [rnd.randrange(ones) for _ in range(20)]:
        for b in big + [rnd.randrange(ones) for _ in range(5)]:
            if a * b // R + p >= R:
                continue
            r = lazy(5, a, b)
            assert (r * R - a * b) % p == 0 and r < a * b // R + p + 1, (hex(a), hex(b))
        if a * a // R + p < R:
            r = lazy(6, a)
            assert (r * R - a * a) % p == 0 and r < a * a // R + p + 1, hex(a)
    # Fr: any 256-bit words (the reduction is exact for every input; results below 2^256 are compared)
    Rq = 1 << 256
    ones_q = (1 << 256) - 1
    avals = [ones_q, ones_q >> 1, q - 1, q, 2 * q - 1, int("3fffffff" * 9, 16) & ones_q] + [rnd.randrange(ones_q) for _ in range(30)]
    bvals = [ones_q >> 2, q - 1, (1 << 254) - 1, int("3fffffff" * 9, 16) & (ones_q >> 2)] + [rnd.randrange(q) for _ in range(10)]
    pairs = [(a, b) for a in avals for b in bvals if a * b // Rq + q < Rq]
    A = np.stack([arr(a, 4) for a, _ in pairs])
    B = np.stack([arr(b, 4) for _, b in pairs])
    out = np.zeros_like(A)
    lib.bh_test_fr_mul_host(_p(out), _p(A), _p(B), len(pairs))
    out2 = np.zeros_like(A)
    lib.bh_test_fr_mul_bform_host(_p(out2), _p(A), _p(B), len(pairs))
    for (a, b), r, r2 in zip(pairs, out, out2):
        want = a * b * pow(Rq, -1, q) % q
        assert val(r) == want and val(r2) == want, (hex(a), hex(b))



def test_plain_c_caller_builds_and_passes_host_checks(tmp_path):
    """tests/c/abi_smoke.c: a C11 translation unit (gcc, no C++) against include/bellman_hip.h and libbellman_hip.so -
    the version string, the plan query, the host-side group operations and BH_ERR_NO_DEVICE without a GPU
    (its `gpu` mode runs in tests/test_gpu_c_abi.py)."""
    import subprocess

    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe, "-L", libdir, "-lbellman_hip",
                    "-Wl,-rpath," + libdir], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "host checks passed" in r.stdout, r.stdout + r.stderr
