"""Generates shim/bellman-hip/src/ffi.rs - the Rust `extern "C"` block - from include/bellman_hip.h so the
two cannot drift (tests/test_shim_cpu.py regenerates and compares).  No Rust toolchain exists in this image:
the output is checked structurally here and compiled by whoever applies the shim.

Usage: python tools/gen_rust_ffi.py [--check]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bellman_hip.h")
OUT = os.path.join(ROOT, "shim", "bellman-hip", "src", "ffi.rs")

OPAQUE = {"bh_ctx": "BhCtx", "bh_bases": "BhBases", "bh_msm_job": "BhMsmJob", "bh_params": "BhParams", "bh_r1cs": "BhR1cs",
          "bh_scalars": "BhScalars", "bh_msm_sharded_job": "BhMsmShardedJob", "bh_proof_job": "BhProofJob"}
STRUCTS = {"bh_csr": "BhCsr", "bh_msm_opts": "BhMsmOpts", "bh_ctx_info_t": "BhCtxInfo"}
SCALAR = {
    "int": "c_int", "unsigned": "c_uint", "unsigned int": "c_uint", "long": "c_long", "uint32_t": "u32", "uint64_t": "u64",
    "int32_t": "i32", "size_t": "usize", "float": "f32", "double": "f64", "char": "c_char", "void": "c_void",
}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def c_type_to_rust(ctype):
    """'const uint64_t *' -> '*const u64' ; 'bh_bases **' -> '*mut *mut BhBases'"""
    t = ctype.strip()
    # base type (with its own leading const), then one `*` per pointer level, each optionally followed by `const`
    # ("bh_ctx *const *": pointer to const pointer to BhCtx -> *const *mut BhCtx)
    m = re.match(r"^(const\s+)?([A-Za-z_][\w ]*?)\s*((?:\*\s*(?:const\s*)?)*)$", t)
    if not m:
        raise ValueError("unknown C type %r" % ctype)
    const, base = bool(m.group(1)), " ".join(m.group(2).split())
    levels = re.findall(r"\*\s*(const)?", m.group(3))
    if base in OPAQUE:
        r = OPAQUE[base]
    elif base in STRUCTS:
        r = STRUCTS[base]
    elif base in SCALAR:
        r = SCALAR[base]
    else:
        raise ValueError("unknown C type %r" % ctype)
    if not levels:
        return r
    # the pointee of level k is const when the qualifier to its left says so: the base's `const` for the innermost
    # pointer, the `const` after the previous `*` for the others; an unqualified outer pointer is an out-parameter
    quals = [const] + [bool(q) for q in levels[:-1]]
    out = r
    for q in quals:
        out = ("*const " if q else "*mut ") + out
    return out


def parse_decls(text):
    text = strip_comments(text)
    text = re.sub(r"typedef struct \{.*?\} \w+;", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    text = re.sub(r'extern "C" \{', " ", text)
    decls = []
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(bh_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if ret.startswith("typedef"):
            continue
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                arr = re.match(r"(.*?)(\w+)\s*\[\d*\]$", a)
                if arr:   # array parameter decays to a pointer
                    ctype, pname = arr.group(1) + "*", arr.group(2)
                else:
                    mm = re.match(r"(.*?)(\w+)$", a)
                    ctype, pname = mm.group(1), mm.group(2)
                params.append((pname, c_type_to_rust(ctype)))
        rret = None if ret == "void" else c_type_to_rust(ret)
        decls.append((name, params, rret))
    return decls


RUST_KEYWORDS = {"type", "in", "ref", "box", "move", "fn", "as", "loop", "match", "where", "use", "mod"}


def render(decls):
    lines = [
        "//! `extern \"C\"` declarations of libbellman_hip - GENERATED from include/bellman_hip.h by",
        "//! tools/gen_rust_ffi.py; do not edit.  One item per C entry point, same order, same argument names.",
        "#![allow(non_camel_case_types, dead_code)]",
        "use std::os::raw::{c_char, c_int, c_long, c_uint, c_void};",
        "",
    ]
    for c, r in OPAQUE.items():
        lines += ["/// opaque `%s`" % c, "#[repr(C)]", "pub struct %s {" % r, "    _private: [u8; 0],", "}"]
    lines += [
        "/// `bh_csr`: one constraint matrix in CSR form (include/bellman_hip.h)",
        "#[repr(C)]", "#[derive(Clone, Copy)]",
        "pub struct BhCsr {", "    pub row_ptr: *const u32,", "    pub var: *const u32,", "    pub coeff: *const u32,", "}",
        "/// `bh_msm_opts`: per-job plan overrides; all-zero = tuned defaults",
        "#[repr(C)]", "#[derive(Clone, Copy, Default)]",
        "pub struct BhMsmOpts {", "    pub window_bits: u32,", "    pub chunk: u32,", "    pub flags: u32,", "}",
        "/// `bh_ctx_info_t`: what bh_ctx_info reports about a context",
        "#[repr(C)]", "#[derive(Clone, Copy, Default, Debug)]",
        "pub struct BhCtxInfo {", "    pub device: i32,", "    pub num_cus: u32,", "    pub hbm_bytes: u64,",
        "    pub hw_queues_requested: u32,", "    pub hw_queues_set_before_hip_init: u32,", "    pub max_jobs_in_flight: u32,",
        "    pub jobs_in_flight: u32,", "    pub pool_bytes_held: u64,", "    pub pool_bytes_idle: u64,", "    pub table_bytes: u64,",
        "    pub table_budget: u64,", "    pub fft_table_bytes: u64,", "    pub fft_table_budget: u64,", "}",
        "",
    ]
    consts = [("BH_OK", 0), ("BH_ERR_UNEXPECTED_IDENTITY", 1), ("BH_ERR_UNEXPECTED_EOF", 2), ("BH_ERR_DEGREE_TOO_LARGE", 3),
              ("BH_ERR_UNCONSTRAINED_VARIABLE", 5), ("BH_ERR_INVALID_POINT", 6), ("BH_ERR_POINT_AT_INFINITY", 7),
              ("BH_ERR_HIP", -1), ("BH_ERR_INVALID_ARG", -2), ("BH_ERR_NO_DEVICE", -3), ("BH_SCALARS_CANONICAL", 0),
              ("BH_SCALARS_MONT", 1), ("BH_G1", 1), ("BH_G2", 2), ("BH_FFT", 0), ("BH_IFFT", 1), ("BH_COSET_FFT", 2),
              ("BH_ICOSET_FFT", 3)]
    for k, v in consts:
        lines.append("pub const %s: c_int = %d;" % (k, v))
    lines += ["pub const BH_POINTS_CHECKED: c_uint = 1;", "pub const BH_POINTS_FORBID_IDENTITY: c_uint = 2;",
              "pub const BH_MSM_SUMS_BYTES: usize = 960;", "", "#[link(name = \"bellman_hip\")]", "extern \"C\" {"]
    for name, params, ret in decls:
        ps = ", ".join("%s: %s" % (("r#" + p) if p in RUST_KEYWORDS else p, t) for p, t in params)
        lines.append("    pub fn %s(%s)%s;" % (name, ps, (" -> " + ret) if ret else ""))
    lines += ["}", ""]
    return "\n".join(lines)


def header_constants():
    text = open(HEADER).read()
    return {m.group(1): m.group(2) for m in re.finditer(r"#define\s+(BH_\w+)\s+\(?(-?\d+)u?\)?", text)}


def main():
    decls = parse_decls(open(HEADER).read())
    out = render(decls)
    if "--check" in sys.argv:
        cur = open(OUT).read()
        if cur != out:
            print("shim/bellman-hip/src/ffi.rs is stale: run python tools/gen_rust_ffi.py")
            sys.exit(1)
        print("ffi.rs matches include/bellman_hip.h (%d entry points)" % len(decls))
        return
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(out)
    print("wrote %s (%d entry points)" % (OUT, len(decls)))


if __name__ == "__main__":
    main()
