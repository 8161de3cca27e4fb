"""Loader of the C-ABI shared library (include/bellman_hip.h).

There is NO CPU fallback: if libbellman_hip.so is missing, or no gfx950 device is visible
when a context is requested, this raises.  The library is built in-tree by
`make -C bellman_amd/csrc` (or `__graft_entry__.build()`).
"""

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# BELLMAN_HIP_LIB: another build of the SAME library (A/B runs of a kernel experiment against the shipped build, e.g.
# tools/build_variant.sh) - honoured only together with BELLMAN_HIP_ALLOW_LIB_OVERRIDE=1, so that a stray variable cannot
# swap the library under the tests or the bench (bench.py prints bh_version() and the library's path and hash into its
# JSON line).  Never a different implementation - there is no fallback of any kind.
_OVERRIDE = os.environ.get("BELLMAN_HIP_LIB") if os.environ.get("BELLMAN_HIP_ALLOW_LIB_OVERRIDE") == "1" else None
LIB_PATH = _OVERRIDE or os.path.join(_HERE, "lib", "libbellman_hip.so")

TEST_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libbellman_hip_test.so")
# the demo circuits once more, compiled WITHOUT the closure checks of the test library (csrc/Makefile): what bench.py times
DEMO_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libbellman_hip_demo.so")
DEMO_EXPORTS = ["bh_groth16_prove_demo", "bh_groth16_demo_r1cs", "bh_groth16_prove_demo_r1cs", "bh_groth16_prove_demo_async",
                "bh_groth16_prove_demo_r1cs_part", "bh_test_synthesis_ms"]

# every symbol include/bellman_hip.h declares (libbellman_hip.so exports exactly these) ...
EXPORTS = [
    "bh_version", "bh_ctx_create", "bh_ctx_destroy", "bh_ctx_log_num_cus", "bh_runtime_configure", "bh_ctx_set_limits", "bh_ctx_info",
    "bh_dev_alloc", "bh_dev_free", "bh_dev_upload", "bh_dev_download", "bh_dev_zero", "bh_stream_create", "bh_stream_create_priority", "bh_stream_destroy", "bh_stream_synchronize", "bh_dev_upload_on",
    "bh_dev_zero_on", "bh_ctx_synchronize", "bh_ctx_accumulations_after", "bh_ctx_trim",
    "bh_fft_fr", "bh_fft_fr_dev", "bh_fr_mul_assign_dev", "bh_fr_sub_assign_dev",
    "bh_fr_divide_by_z_on_coset_dev", "bh_fr_distribute_powers_dev", "bh_h_poly_fr", "bh_h_poly_fr_dev", "bh_h_poly_fr_dev_on",
    "bh_bases_register", "bh_bases_register_uncompressed", "bh_bases_read_uncompressed", "bh_bases_download", "bh_bases_write_uncompressed", "bh_bases_copy_dev", "bh_bases_precompute", "bh_bases_table_info", "bh_bases_wrap_dev", "bh_bases_release", "bh_bases_len",
    "bh_msm_async", "bh_msm_async_dev", "bh_msm_wait", "bh_msm_wait_timed", "bh_msm_wait_profile", "bh_msm_wait_stats", "bh_msm_plan_info", "bh_msm_debug_stages", "bh_point_add", "bh_point_mul", "bh_point_lincomb", "bh_msm_async_opts", "bh_msm_async_dev_opts",
    "bh_scalars_register", "bh_scalars_adopt_dev", "bh_scalars_release", "bh_scalars_len", "bh_scalars_dev_ptr", "bh_msm_async_scalars", "bh_h_poly_fr_scalars", "bh_msm_async_dev_after", "bh_msm_start",
    "bh_msm_sharded_async", "bh_msm_sharded_wait",
    "bh_fixed_base_mul_dev",
    "bh_groth16_params_create", "bh_groth16_params_read", "bh_groth16_generate", "bh_groth16_params_write", "bh_groth16_params_vk_ext", "bh_groth16_params_query", "bh_groth16_params_vk", "bh_proof_write", "bh_groth16_params_release", "bh_groth16_prove_assignment",
    "bh_r1cs_create", "bh_r1cs_release", "bh_r1cs_shape", "bh_r1cs_density", "bh_r1cs_eval_dev", "bh_r1cs_eval_transposed_dev", "bh_fr_powers_dev", "bh_fr_qap_ext_dev",
    "bh_groth16_prove_witness", "bh_groth16_prove_assignment_async", "bh_groth16_prove_witness_async", "bh_groth16_proof_wait",
    "bh_groth16_prove_witness_part", "bh_groth16_sums_add", "bh_groth16_assemble",
]
# ... and what include/bellman_hip_test.h declares: test hooks and the built-in demo circuits, in libbellman_hip_test.so
TEST_EXPORTS = [
    "bh_groth16_prove_demo", "bh_groth16_demo_r1cs", "bh_groth16_prove_demo_r1cs", "bh_groth16_prove_demo_async", "bh_groth16_prove_demo_r1cs_part",
    "bh_test_fr_mul_dev", "bh_test_fp_mul_dev", "bh_test_point_add_dev", "bh_test_g2_k3_dev", "bh_test_g2_pairs_dev", "bh_test_g2_k6_dev",
    "bh_test_fr_mul_host", "bh_test_fr_mul_bform_host", "bh_test_fp_mul_host", "bh_test_point_add_host", "bh_test_point_mul_host", "bh_test_fr_inv_host", "bh_test_fp_lazy_host", "bh_test_proof_slice", "bh_test_synthesis_ms", "bh_test_fr_from_u512_host", "bh_test_fr_ops_host",
    "bh_test_groth16_prove_via_call_sites", "bh_test_demo_assignment", "bh_test_shard_cuts", "bh_test_pool_size_class", "bh_test_capture_check",
]


def library_identity():
    """What was loaded: bh_version(), the path, the first 16 hex digits of the file's sha256 (bench.py's JSON line)."""
    import hashlib

    lib = load()
    with open(LIB_PATH, "rb") as f:
        digest = hashlib.sha256(f.read()).hexdigest()[:16]
    return {"version": lib.bh_version().decode(), "path": os.path.relpath(LIB_PATH, os.path.dirname(_HERE)), "sha256_16": digest,
            "override": bool(_OVERRIDE)}


def build(jobs=8):
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j%d" % jobs])
    return LIB_PATH


_lib = None


class _Libs:
    """The product library, with the test library behind it: an attribute that libbellman_hip.so does not export (a test
    hook, a demo-circuit entry point) resolves in libbellman_hip_test.so, which links against the product library."""

    def __init__(self, product, test):
        self.product, self.test = product, test
        self.demo_unchecked = None

    def use_unchecked_demo_circuits(self):
        """bench.py: the demo-circuit entry points (DEMO_EXPORTS) from libbellman_hip_demo.so - the same circuits whose
        closures do not count their terms (the checking build costs 1.5-2 % of a synthesis)."""
        if self.demo_unchecked is None:
            if not os.path.exists(DEMO_LIB_PATH):
                raise RuntimeError("libbellman_hip_demo.so not built (run `make -C bellman_amd/csrc`)")
            d = ctypes.CDLL(DEMO_LIB_PATH)
            for name in DEMO_EXPORTS:
                f = getattr(d, name)
                t = getattr(self.test, name)
                f.argtypes, f.restype = t.argtypes, t.restype
                self.__dict__[name] = f
            self.demo_unchecked = d
        return self

    def __getattr__(self, name):
        try:
            return getattr(self.product, name)
        except AttributeError:
            if self.test is None:
                raise
            return getattr(self.test, name)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libbellman_hip.so not built (run `make -C bellman_amd/csrc`); "
            "bellman_amd has no CPU fallback"
        )
    # more hardware queues than the runtime's 4: a proof runs 6-7 job streams at once (see api.hip); must be in the
    # environment before the process' first HIP call
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    product = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    lib = _Libs(product, ctypes.CDLL(TEST_LIB_PATH) if os.path.exists(TEST_LIB_PATH) else None)
    lib.bh_runtime_configure()   # records that the request was made before this library's first HIP call
    c = ctypes
    vp, sz, u32, i32 = c.c_void_p, c.c_size_t, c.c_uint32, c.c_int
    lib.bh_version.restype = c.c_char_p
    lib.bh_ctx_create.argtypes = [i32, c.POINTER(vp)]
    lib.bh_ctx_destroy.argtypes = [vp]
    lib.bh_ctx_destroy.restype = None
    lib.bh_ctx_log_num_cus.argtypes = [vp]
    lib.bh_ctx_log_num_cus.restype = u32
    lib.bh_dev_alloc.argtypes = [vp, sz, c.POINTER(vp)]
    lib.bh_dev_free.argtypes = [vp, vp]
    lib.bh_dev_upload.argtypes = [vp, vp, vp, sz]
    lib.bh_dev_download.argtypes = [vp, vp, vp, sz]
    lib.bh_dev_zero.argtypes = [vp, vp, sz]
    lib.bh_stream_create.argtypes = [vp, c.POINTER(vp)]
    lib.bh_stream_create_priority.argtypes = [vp, i32, c.POINTER(vp)]
    lib.bh_stream_destroy.argtypes = [vp, vp]
    lib.bh_stream_synchronize.argtypes = [vp, vp]
    lib.bh_dev_upload_on.argtypes = [vp, vp, vp, sz, vp]
    lib.bh_dev_zero_on.argtypes = [vp, vp, sz, vp]
    lib.bh_ctx_synchronize.argtypes = [vp]
    lib.bh_ctx_accumulations_after.argtypes = [vp, vp]
    lib.bh_ctx_trim.argtypes = [vp]
    lib.bh_test_groth16_prove_via_call_sites.argtypes = [vp, i32, vp, vp, vp, sz, vp, sz, vp, sz, vp, vp, vp, vp, vp, vp, vp]
    lib.bh_test_demo_assignment.argtypes = [i32, sz, c.c_uint64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.bh_test_shard_cuts.argtypes = [vp, sz, sz, vp, sz, vp]
    lib.bh_test_pool_size_class.argtypes = [sz]
    lib.bh_test_pool_size_class.restype = sz
    lib.bh_test_fp_lazy_host.argtypes = [i32, vp, vp, vp]
    lib.bh_msm_plan_info.argtypes = [sz, i32, c.c_uint, vp]
    lib.bh_test_proof_slice.argtypes = [sz, sz, sz, c.POINTER(sz), c.POINTER(sz)]
    lib.bh_test_proof_slice.restype = None
    lib.bh_test_synthesis_ms.argtypes = [i32, sz, c.c_uint64, i32]
    lib.bh_test_synthesis_ms.restype = c.c_double
    lib.bh_test_capture_check.argtypes = [i32, sz, c.c_uint64, c.POINTER(sz)]
    lib.bh_test_capture_check.restype = c.c_double
    lib.bh_fft_fr.argtypes = [vp, vp, u32, i32]
    lib.bh_fft_fr_dev.argtypes = [vp, vp, u32, i32, vp]
    lib.bh_fr_mul_assign_dev.argtypes = [vp, vp, vp, sz, vp]
    lib.bh_fr_sub_assign_dev.argtypes = [vp, vp, vp, sz, vp]
    lib.bh_fr_divide_by_z_on_coset_dev.argtypes = [vp, vp, u32, vp]
    lib.bh_fr_distribute_powers_dev.argtypes = [vp, vp, sz, vp, vp]
    lib.bh_h_poly_fr.argtypes = [vp, vp, vp, vp, sz, vp, c.POINTER(sz)]
    lib.bh_h_poly_fr_dev.argtypes = [vp, vp, vp, vp, u32, vp]
    lib.bh_h_poly_fr_dev_on.argtypes = [vp, vp, vp, vp, vp, u32, vp]
    lib.bh_bases_register.argtypes = [vp, i32, vp, sz, sz, c.c_long, c.POINTER(vp)]
    lib.bh_bases_register_uncompressed.argtypes = [vp, i32, vp, sz, c.POINTER(vp)]
    lib.bh_bases_wrap_dev.argtypes = [vp, i32, vp, sz, c.POINTER(vp)]
    lib.bh_bases_release.argtypes = [vp, vp]
    lib.bh_bases_release.restype = None
    lib.bh_bases_len.argtypes = [vp]
    lib.bh_bases_len.restype = sz
    lib.bh_msm_async.argtypes = [vp, vp, sz, vp, sz, i32, vp, sz, c.POINTER(vp)]
    lib.bh_msm_async_dev.argtypes = [vp, vp, sz, vp, sz, i32, vp, sz, c.POINTER(vp)]
    lib.bh_msm_wait.argtypes = [vp, vp]
    lib.bh_msm_wait_timed.argtypes = [vp, vp, c.POINTER(c.c_float)]
    lib.bh_msm_wait_profile.argtypes = [vp, vp, c.POINTER(c.c_float)]
    lib.bh_point_add.argtypes = [i32, vp, vp, vp, sz]
    lib.bh_point_add.restype = None
    lib.bh_point_mul.argtypes = [i32, vp, vp, vp]
    lib.bh_point_mul.restype = None
    lib.bh_point_lincomb.argtypes = [i32, vp, vp, vp, ctypes.c_size_t]
    lib.bh_point_lincomb.restype = None
    lib.bh_msm_async_opts.argtypes = [vp, vp, sz, vp, sz, i32, vp, sz, vp, c.POINTER(vp)]
    lib.bh_msm_async_dev_opts.argtypes = [vp, vp, sz, vp, sz, i32, vp, sz, vp, c.POINTER(vp)]
    lib.bh_fixed_base_mul_dev.argtypes = [vp, i32, vp, vp, sz, i32, vp, vp]
    lib.bh_runtime_configure.argtypes = []
    lib.bh_ctx_set_limits.argtypes = [vp, u32, sz, sz, sz]
    lib.bh_ctx_info.argtypes = [vp, vp]
    lib.bh_scalars_register.argtypes = [vp, vp, sz, i32, c.POINTER(vp)]
    lib.bh_scalars_adopt_dev.argtypes = [vp, vp, sz, i32, i32, c.POINTER(vp)]
    lib.bh_scalars_release.argtypes = [vp]
    lib.bh_scalars_release.restype = None
    lib.bh_scalars_len.argtypes = [vp]
    lib.bh_scalars_len.restype = sz
    lib.bh_scalars_dev_ptr.argtypes = [vp]
    lib.bh_scalars_dev_ptr.restype = vp
    lib.bh_msm_async_scalars.argtypes = [vp, vp, sz, vp, sz, sz, vp, sz, vp, c.POINTER(vp)]
    lib.bh_msm_async_dev_after.argtypes = [vp, vp, sz, vp, sz, i32, vp, sz, vp, vp, c.POINTER(vp)]
    lib.bh_msm_start.argtypes = [vp]
    lib.bh_h_poly_fr_scalars.argtypes = [vp, vp, vp, vp, sz, c.POINTER(vp)]
    lib.bh_msm_sharded_async.argtypes = [vp, vp, sz, sz, vp, sz, i32, vp, sz, c.POINTER(vp)]
    lib.bh_msm_sharded_wait.argtypes = [vp, vp]
    lib.bh_groth16_params_create.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz, vp, sz, vp, sz, vp, sz, vp, sz, c.POINTER(vp)]
    lib.bh_groth16_params_release.argtypes = [vp]
    lib.bh_groth16_params_read.argtypes = [vp, vp, sz, i32, c.POINTER(vp)]
    lib.bh_groth16_params_query.argtypes = [vp, i32, c.POINTER(vp), c.POINTER(sz)]
    lib.bh_groth16_params_vk.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.bh_proof_write.argtypes = [vp, vp]
    lib.bh_groth16_generate.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, c.POINTER(vp)]
    lib.bh_groth16_params_write.argtypes = [vp, vp, sz, c.POINTER(sz)]
    lib.bh_groth16_params_vk_ext.argtypes = [vp, vp, vp, sz, c.POINTER(sz)]
    lib.bh_bases_copy_dev.argtypes = [vp, i32, vp, sz, c.POINTER(vp)]
    lib.bh_bases_precompute.argtypes = [vp, vp, c.c_uint]
    lib.bh_bases_table_info.argtypes = [vp, c.POINTER(c.c_uint), c.POINTER(c.c_uint), c.POINTER(sz)]
    lib.bh_r1cs_eval_transposed_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.bh_fr_powers_dev.argtypes = [vp, vp, sz, vp, vp, vp]
    lib.bh_fr_qap_ext_dev.argtypes = [vp, vp, vp, vp, vp, sz, sz, vp, vp, vp, vp, vp]
    lib.bh_proof_write.restype = None
    lib.bh_bases_read_uncompressed.argtypes = [vp, i32, vp, sz, c.c_uint, c.POINTER(vp), c.POINTER(sz)]
    lib.bh_bases_download.argtypes = [vp, vp, sz, sz, vp]
    lib.bh_bases_write_uncompressed.argtypes = [vp, vp, sz, sz, vp]
    lib.bh_groth16_params_release.restype = None
    lib.bh_groth16_prove_assignment.argtypes = [vp, vp, vp, vp, sz, vp, sz, vp, sz, vp, vp, vp, vp, vp, vp, vp]
    lib.bh_groth16_prove_demo.argtypes = [vp, i32, sz, c.c_uint64, vp, vp, vp, vp, vp, vp]
    lib.bh_r1cs_create.argtypes = [vp, sz, sz, sz, vp, vp, sz, c.POINTER(vp)]
    lib.bh_r1cs_release.argtypes = [vp]
    lib.bh_r1cs_release.restype = None
    lib.bh_r1cs_shape.argtypes = [vp, c.POINTER(sz), c.POINTER(sz), c.POINTER(sz)]
    lib.bh_r1cs_density.argtypes = [vp, i32, c.POINTER(vp), c.POINTER(vp), c.POINTER(sz)]
    lib.bh_r1cs_eval_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, c.c_uint32, vp]
    lib.bh_groth16_prove_witness.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp, vp, vp]
    lib.bh_groth16_demo_r1cs.argtypes = [vp, i32, sz, c.c_uint64, vp, c.POINTER(vp)]
    lib.bh_groth16_prove_demo_r1cs.argtypes = [vp, vp, i32, sz, c.c_uint64, vp, vp, vp, vp, vp, vp]
    lib.bh_groth16_prove_demo_async.argtypes = [vp, vp, i32, sz, c.c_uint64, vp, vp, vp, vp, c.POINTER(vp)]
    lib.bh_groth16_proof_wait.argtypes = [vp, vp, vp]
    lib.bh_groth16_prove_assignment_async.argtypes = [vp, vp, vp, vp, sz, vp, sz, vp, sz, vp, vp, vp, vp, vp, c.POINTER(vp)]
    lib.bh_groth16_prove_witness_async.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp, c.POINTER(vp)]
    lib.bh_groth16_prove_witness_part.argtypes = [vp, vp, vp, sz, vp, sz, sz, sz, vp, vp]
    lib.bh_groth16_sums_add.argtypes = [vp, vp]
    lib.bh_groth16_sums_add.restype = None
    lib.bh_groth16_assemble.argtypes = [vp, vp, vp, vp, vp]
    lib.bh_groth16_prove_demo_r1cs_part.argtypes = [vp, vp, i32, sz, c.c_uint64, vp, vp, sz, sz, vp, vp]
    lib.bh_test_fr_mul_dev.argtypes = [vp, vp, vp, vp, sz]
    lib.bh_test_fp_mul_dev.argtypes = [vp, vp, vp, vp, sz]
    lib.bh_test_point_add_dev.argtypes = [vp, i32, vp, vp, vp, sz]
    lib.bh_msm_debug_stages.argtypes = [vp, vp, sz, i32, c.c_uint, vp, vp]
    lib.bh_msm_wait_stats.argtypes = [vp, vp, c.POINTER(c.c_float), c.POINTER(c.c_uint64)]
    lib.bh_test_g2_k3_dev.argtypes = [vp, vp, vp, vp, vp, vp, sz]
    lib.bh_test_g2_pairs_dev.argtypes = [vp, vp, vp, vp, vp, vp, sz]
    lib.bh_test_g2_k6_dev.argtypes = [vp, vp, vp, vp, sz]
    for name in ("bh_test_fr_mul_host", "bh_test_fp_mul_host", "bh_test_fr_mul_bform_host"):
        getattr(lib, name).argtypes = [vp, vp, vp, sz]
        getattr(lib, name).restype = None
    lib.bh_test_fr_inv_host.argtypes = [vp, vp, sz]
    lib.bh_test_fr_inv_host.restype = None
    lib.bh_test_point_add_host.argtypes = [i32, vp, vp, vp, sz]
    lib.bh_test_point_add_host.restype = None
    lib.bh_test_point_mul_host.argtypes = [i32, vp, vp, vp]
    lib.bh_test_point_mul_host.restype = None
    _lib = lib
    return lib
