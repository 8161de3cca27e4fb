"""Host-side logic added in round 3, checked without a GPU:
  * the workspace pool's size classes (1/8 octave: a request is rounded up by at most 12.5 %);
  * where a multi-context multiexp cuts its exponents (bh_msm_sharded_async; the reference semantics are those of
    ONE multiexp over the concatenated bases, src/multiexp.rs:45-86,210-332) against a numpy model;
  * the ProvingAssignment the C++ demo circuits synthesise (prover.rs:182-215) against the Python restatement;
  * the Rust patch: groth16/src/prover.rs is patched, every bellman_hip:: call it makes exists in the shim crate, and
    every ffi:: call of the crate exists in the generated ffi.rs."""

import ctypes
import os
import re

import numpy as np
import pytest

from bellman_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pool_size_classes_are_an_eighth_octave():
    lib = _lib.load()
    rnd = np.random.default_rng(1)
    for bytes_ in [1, 255, 256, 257, 4096, 600 << 20, (1 << 30) + 1, 30 << 30] + [int(x) for x in rnd.integers(1, 1 << 36, 200)]:
        c = lib.bh_test_pool_size_class(bytes_)
        assert c >= bytes_ and c >= 256
        assert c <= max(256, bytes_ + bytes_ // 8 + 1), (bytes_, c)      # at most 12.5 % waste
        assert lib.bh_test_pool_size_class(c) == c                        # classes are fixed points
    # a 0.6 GB multiexp workspace used to be rounded to 1 GiB
    assert lib.bh_test_pool_size_class(600 << 20) <= 640 << 20


def _model_cuts(lens, skip, density, n):
    """scalar i goes to the shard that holds base skip + rank_i (dense entries); non-dense entries go with the
    preceding dense entry's shard... any assignment of non-dense entries is valid, so only dense entries are checked."""
    off = np.concatenate([[0], np.cumsum(lens)])
    dense = np.ones(n, bool) if density is None else density
    rank = np.cumsum(dense) - dense
    base = skip + rank
    shard_of = np.searchsorted(off[1:], base, side="right")
    shard_of = np.minimum(shard_of, len(lens) - 1)     # beyond the end: the last shard reports EOF
    return dense, shard_of


def test_shard_cuts_match_the_base_index_model():
    lib = _lib.load()
    rnd = np.random.default_rng(7)
    for trial in range(200):
        k = int(rnd.integers(1, 6))
        lens = rnd.integers(0, 40, k).astype(np.uint64)
        n = int(rnd.integers(0, 150))
        skip = int(rnd.integers(0, 30))
        use_density = trial % 2 == 1
        density = rnd.random(n) < rnd.random() if use_density else None
        words = None
        if use_density:
            padded = np.zeros(((n + 63) // 64 + 1) * 64, dtype=np.uint8)
            padded[:n] = density
            words = np.packbits(padded, bitorder="little").view(np.uint64).copy()
        cuts = np.zeros(k + 1, dtype=np.uint64)
        rc = lib.bh_test_shard_cuts(lens.ctypes.data_as(ctypes.c_void_p), k, skip,
                                    None if words is None else words.ctypes.data_as(ctypes.c_void_p), n,
                                    cuts.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        cuts = cuts.astype(np.int64)
        assert cuts[0] == 0 and cuts[-1] == n and np.all(np.diff(cuts) >= 0), (lens, skip, n, cuts)
        dense, shard_of = _model_cuts(lens.astype(np.int64), skip, density, n)
        got = np.searchsorted(cuts[1:], np.arange(n), side="right")
        got = np.minimum(got, k - 1)
        assert np.array_equal(got[dense], shard_of[dense]), (lens, skip, n, cuts)


def test_demo_assignment_matches_the_python_restatement():
    """bh_test_demo_assignment (C++ ChainCircuit through ProvingAssignment) == tests.circuits.chain_assignment_fast"""
    from bellman_amd import groth16 as pg
    from oracle import cref
    from tests import circuits

    rounds, seed, x0 = 37, 99, 123456789
    asg = pg.demo_assignment(1, rounds, seed, [x0])
    f = circuits.chain_assignment_fast(rounds, seed, x0)
    for key in ("a", "b", "c", "input_assignment", "aux_assignment"):
        assert cref.arr_to_ints(cref.fr_from_mont(asg[key])) == [v % circuits.Q for v in f[key]], key
    for key in ("a_aux_density", "b_input_density", "b_aux_density"):
        bits = np.unpackbits(asg[key].view(np.uint8), bitorder="little")[:len(f[key])].astype(bool)
        assert list(bits) == list(f[key]), key


@pytest.mark.parametrize("kind,rounds", [(2, 1), (2, 2), (2, 7), (2, 200), (3, 1), (3, 5), (3, 60), (3, 400), (3, 401), (3, 1500)])
def test_every_form_of_linear_combination_matches_the_oracle(kind, rounds):
    """The C++ mirror's ProvingAssignment on FormsCircuit (evaluating terms with a product, `-`, zero coefficients, stored
    combinations that spill out of the inline storage, the empty combination, public inputs in the middle) == the oracle's
    ProvingAssignment (oracle/pyref/prover.py, prover.rs:57-162) on tests.circuits.forms_circuit - and the same for
    RandomCircuit / random_circuit (kind 3), whose structure (variables per round, empty / chained / stored combinations
    of up to 9 terms, +, -, coefficients 0, 1, -1, small, random, any earlier variable) is drawn from the seed: evaluations, assignments
    and the three density maps; and the structure capture of the same circuit agrees with its own ProvingAssignment
    (bh_test_capture_check: the captured matrices times the assignment).  Host code only."""
    import ctypes

    from bellman_amd import _lib
    from bellman_amd import groth16 as pg
    from oracle import cref
    from oracle.pyref import prover as oprover
    from oracle.pyref.core import INPUT, Variable
    from tests import circuits

    seed, x0 = 7 + rounds, 0x1234567890ABCDEF
    asg = pg.demo_assignment(kind, rounds, seed, [x0])
    pa = oprover.ProvingAssignment(circuits.Q)
    pa.alloc_input(lambda: 1)
    (circuits.forms_circuit if kind == 2 else circuits.random_circuit)(rounds, seed, x0)(pa)
    for i in range(len(pa.input_assignment)):   # prover.rs:208-215
        pa.enforce(lambda lc: lc + Variable(INPUT, i), lambda lc: lc, lambda lc: lc)
    for key in ("a", "b", "c", "input_assignment", "aux_assignment"):
        assert cref.arr_to_ints(cref.fr_from_mont(asg[key])) == [v % circuits.Q for v in getattr(pa, key)], key
    for key in ("a_aux_density", "b_input_density", "b_aux_density"):
        want = getattr(pa, key).bv
        bits = np.unpackbits(asg[key].view(np.uint8), bitorder="little")[:len(want)].astype(bool)
        assert list(bits) == [bool(b) for b in want], key
    lib = _lib.load()
    lib.bh_test_capture_check.restype = ctypes.c_double
    out4 = (ctypes.c_size_t * 4)()
    ms = lib.bh_test_capture_check(kind, ctypes.c_size_t(rounds), ctypes.c_uint64(seed), out4)
    assert ms >= 0 and out4[0] == len(pa.a) and out4[3] == 0, list(out4)


def test_mirror_under_sanitizers(tmp_path):
    """tests/cpp/mirror_sanitized.cpp: synthesis of the fixture circuits through the mirror (inline-assembly field
    arithmetic, evaluating / stored combinations, capture) under ASan + UBSan; every captured matrix must reproduce its
    ProvingAssignment."""
    import shutil
    import subprocess

    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx):
        cxx = shutil.which("clang++") or shutil.which("g++")
    csrc = os.path.join(ROOT, "bellman_amd", "csrc")
    libdir = os.path.join(ROOT, "bellman_amd", "lib")
    exe = str(tmp_path / "mirror_sanitized.bin")
    cmd = [cxx, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-mbmi2", "-madx", "-std=c++17",
           os.path.join(ROOT, "tests", "cpp", "mirror_sanitized.cpp")] + \
          [os.path.join(csrc, f) for f in ("demo_circuits.cpp", "groth16_prover.cpp", "groth16_fr.cpp", "groth16_params.cpp")] + \
          ["-o", exe, "-L" + libdir, "-lbellman_hip", "-Wl,-rpath," + libdir]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and "sanitizer" in (build.stderr + build.stdout).lower():
        pytest.skip("no sanitizer runtime for this compiler")
    assert build.returncode == 0, build.stderr[-2000:]
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0 and "synthesis ms" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_prover_rs_patch_and_shim_are_consistent():
    patch = open(os.path.join(ROOT, "shim", "patches", "bellman-hip.patch")).read()
    assert "+++ b/groth16/src/prover.rs" in patch and "+++ b/groth16/Cargo.toml" in patch
    assert "issue_on_device" in patch and "finish_proof" in patch
    lib_rs = open(os.path.join(ROOT, "shim", "bellman-hip", "src", "lib.rs")).read()
    ffi_rs = open(os.path.join(ROOT, "shim", "bellman-hip", "src", "ffi.rs")).read()
    added = "\n".join(l[1:] for l in patch.splitlines() if l.startswith("+") and not l.startswith("+++"))
    # context methods the patched bellman calls on bellman_hip::Context
    for m in set(re.findall(r"\bctx\.(\w+)\(", added)):
        assert re.search(r"pub fn %s\b" % m, lib_rs), "bellman_hip::Context::%s missing" % m
    for name in set(re.findall(r"bellman_hip::(\w+)", added)):
        if name[0].islower():
            assert re.search(r"pub fn %s\b" % name, lib_rs), name
        else:
            assert re.search(r"pub (struct|enum) %s\b" % name, lib_rs), name
    # functions of bellman::hip the patched prover.rs uses exist in the patched src/hip.rs
    hip_rs = added[added.index("//! Routes the BLS12-381 instantiations"):]
    for f in set(re.findall(r"\bdev::(\w+)", added)):
        assert re.search(r"pub fn %s\b" % f, hip_rs), "bellman::hip::%s missing" % f
    for fn in set(re.findall(r"ffi::(bh_\w+)\(", lib_rs)):
        assert "pub fn %s(" % fn in ffi_rs, fn
    # an unwaited job is waited for when dropped (create_proof drops the remaining Waiters on an early `?`)
    assert "impl Drop for MsmJob" in lib_rs and "impl Drop for Scalars" in lib_rs
    # `#[cfg]` never sits directly on an `if` (attributes on if-expressions are rejected by rustc)
    assert not re.search(r"#\[cfg\([^\n]*\)\]\n\+?\s*if ", added)


def test_fused_y3_host_check(tmp_path):
    """The fused last line of the mixed addition (R*(Q - X3) - Y1*PPP as two products under one Montgomery reduction,
    ff.cuh fe_mul2; the default since round 4, profiles/archive/r4_call1_fused_y3.txt): the curve and field code is
    __host__ __device__, so the formula and the multiplier are also checked on the host (tests/cpp/fused_y3_check.hip) -
    against the separate products and against the general addition, incl. operands with every limb set, the doubling
    and the inverse case (src/multiexp.rs:39)."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "tests", "cpp", "fused_y3_check.hip")
    exe = str(tmp_path / "fused_y3_check.bin")
    subprocess.run([hipcc, "--cuda-host-only", "-O2", "-std=c++17", src, "-o", exe], check=True,
                   capture_output=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "fused Y3: ok" in out.stdout, out.stdout + out.stderr
    # ... and no build switch is left around it
    for f in ("ff.cuh", "ec.cuh", "Makefile"):
        assert "BH_FUSED" not in open(os.path.join(ROOT, "bellman_amd", "csrc", f)).read(), f


@pytest.mark.parametrize("kind,size,table", [(0, 322, 9), (0, 4000, 9), (1, 1000, 2001), (1, (1 << 16) - 3, 131067)])
def test_structure_capture_matches_proving_assignment(kind, size, table):
    """The constraint matrices a circuit is captured into (what stays in HBM for the R1CS-resident prover; the
    KeypairAssembly of generator.rs:43-131 plus the input rows of prover.rs:208-215) times the assignment give the a, b, c
    rows a ProvingAssignment evaluates for the same circuit (prover.rs:105-145) - on the host, no device.  The coefficient
    table shares repeated constants (MiMC: 7 distinct round constants here + one + the input rows' ones)."""
    lib = _lib.load()
    out = (ctypes.c_size_t * 4)()
    ms = lib.bh_test_capture_check(kind, size, 2020, out)
    assert ms >= 0
    rows, terms, coeffs, bad = list(out)
    assert bad == 0 and rows >= size and terms > rows
    assert coeffs == table


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("rounds", [0, 3, 64])
def test_closures_that_discard_terms_are_refused(variant, rounds):
    """[r5] The mirror's evaluating LinearCombination counts a term the moment it is added (value and density maps live in
    the sink, groth16.hpp), whereas the reference's `eval` walks only the combination a closure RETURNS
    (groth16/src/prover.rs:19-55).  A closure that adds a term to a copy of its argument and discards the copy, builds two
    combinations and returns one, or touches its argument and returns a stored combination, would therefore give other
    densities than bellman - MisuseCircuit (fixture kind 4) does each of these after `rounds` well-behaved constraints.  The
    test library's closures are compiled with BELLMAN_HIP_CHECK_CLOSURES (term count in the sink): ProvingAssignment::enforce
    and the structure capture refuse them (std::invalid_argument -> BH_ERR_INVALID_ARG); the well-behaved kinds pass
    through the same checking build in every other test of this file."""
    import ctypes

    from bellman_amd import _lib
    from bellman_amd import groth16 as pg
    with pytest.raises(AssertionError, match="invalid argument"):   # BH_ERR_INVALID_ARG (the reference would panic)
        pg.demo_assignment(4, rounds, variant, [0x55AA55AA])
    lib = _lib.load()
    lib.bh_test_capture_check.restype = ctypes.c_double
    out4 = (ctypes.c_size_t * 4)()
    assert lib.bh_test_capture_check(4, ctypes.c_size_t(rounds), ctypes.c_uint64(variant), out4) < 0
