"""The Rust side of the boundary (shim/) cannot be compiled in this image (no Rust toolchain), so what can be
checked mechanically is: ffi.rs is exactly what the generator makes from include/bellman_hip.h (names,
argument names, argument count and types per entry point), the test-hook header is not bound, the constants
agree with the header, and the patch still applies to the reference tree when that tree is present."""

import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_ffi_rs_is_generated_from_the_header():
    import gen_rust_ffi as g

    decls = g.parse_decls(open(g.HEADER).read())
    assert open(g.OUT).read() == g.render(decls), "run python tools/gen_rust_ffi.py"
    names = [d[0] for d in decls]
    assert len(names) == len(set(names)) and "bh_msm_async" in names and "bh_fft_fr" in names
    assert not [n for n in names if n.startswith("bh_test_")]
    # every entry point the library exports through the product header is bound, and nothing else
    from bellman_amd import _lib

    assert set(names) == {s for s in _lib.EXPORTS if not s.startswith("bh_test_")}


def test_ffi_constants_match_the_header():
    import gen_rust_ffi as g

    hdr = g.header_constants()
    rs = open(g.OUT).read()
    for name, val in re.findall(r"pub const (BH_\w+): \w+ = (-?\d+);", rs):
        assert name in hdr and int(hdr[name]) == int(val), name


def test_ffi_signatures_spot_checks():
    rs = open(os.path.join(ROOT, "shim", "bellman-hip", "src", "ffi.rs")).read()
    assert ("pub fn bh_bases_register(ctx: *mut BhCtx, group: c_int, host_points: *const c_void, n: usize, stride: usize, "
            "inf_offset: c_long, out: *mut *mut BhBases) -> c_int;") in rs
    assert "pub fn bh_msm_wait(job: *mut BhMsmJob, out_affine: *mut c_void) -> c_int;" in rs
    assert "pub fn bh_fft_fr(ctx: *mut BhCtx, data_host: *mut c_void, log_n: u32, mode: c_int) -> c_int;" in rs
    assert "pub fn bh_point_add(group: c_int, r: *mut c_void, a: *const c_void, b: *const c_void, n: usize);" in rs
    # the wrapper only calls functions that exist
    lib_rs = open(os.path.join(ROOT, "shim", "bellman-hip", "src", "lib.rs")).read()
    for fn in set(re.findall(r"ffi::(bh_\w+)\(", lib_rs)):
        assert "pub fn %s(" % fn in rs, fn


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference tree not present (GPU box)")
def test_patch_applies_to_the_reference_tree(tmp_path):
    dst = tmp_path / "ref"
    shutil.copytree("/root/reference", dst, ignore=shutil.ignore_patterns(".git", "target"))
    patch = os.path.join(ROOT, "shim", "patches", "bellman-hip.patch")
    r = subprocess.run(["patch", "-p1", "--dry-run", "-i", patch], cwd=dst, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "src/hip.rs" in r.stdout and "src/multiexp.rs" in r.stdout


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference tree not present (GPU box)")
def test_patched_rust_files_are_balanced(tmp_path):
    """No Rust toolchain here: at least the patched files must keep their brackets balanced outside strings and
    comments (a truncated hunk or a lost brace is the kind of damage a hand-maintained patch suffers)."""
    dst = tmp_path / "ref"
    shutil.copytree("/root/reference", dst, ignore=shutil.ignore_patterns(".git", "target"))
    patch = os.path.join(ROOT, "shim", "patches", "bellman-hip.patch")
    r = subprocess.run(["patch", "-p1", "-s", "-i", patch], cwd=dst, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    files = [dst / "src" / f for f in ("hip.rs", "multiexp.rs", "domain.rs", "multicore.rs", "lib.rs")] + \
            [dst / "groth16" / "src" / "prover.rs"] + \
            [os.path.join(ROOT, "shim", "bellman-hip", "src", f) for f in ("lib.rs", "ffi.rs", "layout.rs")]
    for f in files:
        text = open(f).read()
        text = re.sub(r"//[^\n]*", "", text)                       # line comments
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)           # block comments
        text = re.sub(r'"(?:\\.|[^"\\])*"', '""', text)             # string literals
        text = re.sub(r"'(?:\\.|[^'\\])'", "''", text)              # char literals (lifetimes stay, they carry no brackets)
        for o, c in ("{}", "()", "[]"):
            assert text.count(o) == text.count(c), (str(f), o, text.count(o), text.count(c))
