//! `extern "C"` declarations of libbellman_hip - GENERATED from include/bellman_hip.h by
//! tools/gen_rust_ffi.py; do not edit.  One item per C entry point, same order, same argument names.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_long, c_uint, c_void};

/// opaque `bh_ctx`
#[repr(C)]
pub struct BhCtx {
    _private: [u8; 0],
}
/// opaque `bh_bases`
#[repr(C)]
pub struct BhBases {
    _private: [u8; 0],
}
/// opaque `bh_msm_job`
#[repr(C)]
pub struct BhMsmJob {
    _private: [u8; 0],
}
/// opaque `bh_params`
#[repr(C)]
pub struct BhParams {
    _private: [u8; 0],
}
/// opaque `bh_r1cs`
#[repr(C)]
pub struct BhR1cs {
    _private: [u8; 0],
}
/// opaque `bh_scalars`
#[repr(C)]
pub struct BhScalars {
    _private: [u8; 0],
}
/// opaque `bh_msm_sharded_job`
#[repr(C)]
pub struct BhMsmShardedJob {
    _private: [u8; 0],
}
/// opaque `bh_proof_job`
#[repr(C)]
pub struct BhProofJob {
    _private: [u8; 0],
}
/// `bh_csr`: one constraint matrix in CSR form (include/bellman_hip.h)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct BhCsr {
    pub row_ptr: *const u32,
    pub var: *const u32,
    pub coeff: *const u32,
}
/// `bh_msm_opts`: per-job plan overrides; all-zero = tuned defaults
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct BhMsmOpts {
    pub window_bits: u32,
    pub chunk: u32,
    pub flags: u32,
}
/// `bh_ctx_info_t`: what bh_ctx_info reports about a context
#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct BhCtxInfo {
    pub device: i32,
    pub num_cus: u32,
    pub hbm_bytes: u64,
    pub hw_queues_requested: u32,
    pub hw_queues_set_before_hip_init: u32,
    pub max_jobs_in_flight: u32,
    pub jobs_in_flight: u32,
    pub pool_bytes_held: u64,
    pub pool_bytes_idle: u64,
    pub table_bytes: u64,
    pub table_budget: u64,
    pub fft_table_bytes: u64,
    pub fft_table_budget: u64,
}

pub const BH_OK: c_int = 0;
pub const BH_ERR_UNEXPECTED_IDENTITY: c_int = 1;
pub const BH_ERR_UNEXPECTED_EOF: c_int = 2;
pub const BH_ERR_DEGREE_TOO_LARGE: c_int = 3;
pub const BH_ERR_UNCONSTRAINED_VARIABLE: c_int = 5;
pub const BH_ERR_INVALID_POINT: c_int = 6;
pub const BH_ERR_POINT_AT_INFINITY: c_int = 7;
pub const BH_ERR_HIP: c_int = -1;
pub const BH_ERR_INVALID_ARG: c_int = -2;
pub const BH_ERR_NO_DEVICE: c_int = -3;
pub const BH_SCALARS_CANONICAL: c_int = 0;
pub const BH_SCALARS_MONT: c_int = 1;
pub const BH_G1: c_int = 1;
pub const BH_G2: c_int = 2;
pub const BH_FFT: c_int = 0;
pub const BH_IFFT: c_int = 1;
pub const BH_COSET_FFT: c_int = 2;
pub const BH_ICOSET_FFT: c_int = 3;
pub const BH_POINTS_CHECKED: c_uint = 1;
pub const BH_POINTS_FORBID_IDENTITY: c_uint = 2;
pub const BH_MSM_SUMS_BYTES: usize = 960;

#[link(name = "bellman_hip")]
extern "C" {
    pub fn bh_ctx_create(device: c_int, out: *mut *mut BhCtx) -> c_int;
    pub fn bh_ctx_destroy(ctx: *mut BhCtx);
    pub fn bh_ctx_log_num_cus(ctx: *const BhCtx) -> u32;
    pub fn bh_version() -> *const c_char;
    pub fn bh_runtime_configure() -> c_int;
    pub fn bh_ctx_set_limits(ctx: *mut BhCtx, max_jobs_in_flight: u32, pool_cap_bytes: usize, table_budget_bytes: usize, fft_table_budget_bytes: usize) -> c_int;
    pub fn bh_ctx_info(ctx: *mut BhCtx, info: *mut BhCtxInfo) -> c_int;
    pub fn bh_dev_alloc(ctx: *mut BhCtx, bytes: usize, dev_ptr: *mut *mut c_void) -> c_int;
    pub fn bh_dev_free(ctx: *mut BhCtx, dev_ptr: *mut c_void) -> c_int;
    pub fn bh_dev_upload(ctx: *mut BhCtx, dev_dst: *mut c_void, host_src: *const c_void, bytes: usize) -> c_int;
    pub fn bh_dev_download(ctx: *mut BhCtx, host_dst: *mut c_void, dev_src: *const c_void, bytes: usize) -> c_int;
    pub fn bh_dev_zero(ctx: *mut BhCtx, dev_ptr: *mut c_void, bytes: usize) -> c_int;
    pub fn bh_stream_create(ctx: *mut BhCtx, stream: *mut *mut c_void) -> c_int;
    pub fn bh_stream_create_priority(ctx: *mut BhCtx, high: c_int, stream: *mut *mut c_void) -> c_int;
    pub fn bh_stream_destroy(ctx: *mut BhCtx, stream: *mut c_void) -> c_int;
    pub fn bh_stream_synchronize(ctx: *mut BhCtx, stream: *mut c_void) -> c_int;
    pub fn bh_dev_upload_on(ctx: *mut BhCtx, dev_dst: *mut c_void, host_src: *const c_void, bytes: usize, stream: *mut c_void) -> c_int;
    pub fn bh_dev_zero_on(ctx: *mut BhCtx, dev_ptr: *mut c_void, bytes: usize, stream: *mut c_void) -> c_int;
    pub fn bh_ctx_synchronize(ctx: *mut BhCtx) -> c_int;
    pub fn bh_ctx_accumulations_after(ctx: *mut BhCtx, stream: *mut c_void) -> c_int;
    pub fn bh_ctx_trim(ctx: *mut BhCtx) -> c_int;
    pub fn bh_fft_fr(ctx: *mut BhCtx, data_host: *mut c_void, log_n: u32, mode: c_int) -> c_int;
    pub fn bh_fft_fr_dev(ctx: *mut BhCtx, data_dev: *mut c_void, log_n: u32, mode: c_int, stream: *mut c_void) -> c_int;
    pub fn bh_fr_mul_assign_dev(ctx: *mut BhCtx, a_dev: *mut c_void, b_dev: *const c_void, n: usize, stream: *mut c_void) -> c_int;
    pub fn bh_fr_sub_assign_dev(ctx: *mut BhCtx, a_dev: *mut c_void, b_dev: *const c_void, n: usize, stream: *mut c_void) -> c_int;
    pub fn bh_fr_divide_by_z_on_coset_dev(ctx: *mut BhCtx, a_dev: *mut c_void, log_n: u32, stream: *mut c_void) -> c_int;
    pub fn bh_fr_distribute_powers_dev(ctx: *mut BhCtx, a_dev: *mut c_void, n: usize, g_host: *const c_void, stream: *mut c_void) -> c_int;
    pub fn bh_h_poly_fr(ctx: *mut BhCtx, a_host: *const c_void, b_host: *const c_void, c_host: *const c_void, n_evals: usize, h_out_host: *mut c_void, h_len: *mut usize) -> c_int;
    pub fn bh_h_poly_fr_dev(ctx: *mut BhCtx, a_dev: *mut c_void, b_dev: *mut c_void, c_dev: *mut c_void, log_n: u32, stream: *mut c_void) -> c_int;
    pub fn bh_h_poly_fr_dev_on(ctx: *mut BhCtx, a_dev: *mut c_void, b_dev: *mut c_void, c_dev: *mut c_void, scratch_dev: *mut c_void, log_n: u32, stream: *mut c_void) -> c_int;
    pub fn bh_bases_register(ctx: *mut BhCtx, group: c_int, host_points: *const c_void, n: usize, stride: usize, inf_offset: c_long, out: *mut *mut BhBases) -> c_int;
    pub fn bh_bases_register_uncompressed(ctx: *mut BhCtx, group: c_int, host_bytes: *const c_void, n: usize, out: *mut *mut BhBases) -> c_int;
    pub fn bh_bases_read_uncompressed(ctx: *mut BhCtx, group: c_int, host_bytes: *const c_void, n: usize, flags: c_uint, out: *mut *mut BhBases, bad_index: *mut usize) -> c_int;
    pub fn bh_bases_download(ctx: *mut BhCtx, b: *const BhBases, first: usize, count: usize, out_host: *mut c_void) -> c_int;
    pub fn bh_bases_write_uncompressed(ctx: *mut BhCtx, bases: *const BhBases, first: usize, count: usize, out_host_bytes: *mut c_void) -> c_int;
    pub fn bh_bases_precompute(ctx: *mut BhCtx, b: *mut BhBases, window_bits: c_uint) -> c_int;
    pub fn bh_bases_table_info(b: *const BhBases, window_bits: *mut c_uint, rows: *mut c_uint, bytes: *mut usize) -> c_int;
    pub fn bh_bases_copy_dev(ctx: *mut BhCtx, group: c_int, dev_points: *const c_void, n: usize, out: *mut *mut BhBases) -> c_int;
    pub fn bh_bases_wrap_dev(ctx: *mut BhCtx, group: c_int, dev_points: *const c_void, n: usize, out: *mut *mut BhBases) -> c_int;
    pub fn bh_bases_release(ctx: *mut BhCtx, b: *mut BhBases);
    pub fn bh_bases_len(b: *const BhBases) -> usize;
    pub fn bh_msm_async(ctx: *mut BhCtx, bases: *const BhBases, skip: usize, scalars_host: *const c_void, n_scalars: usize, scalar_fmt: c_int, density_words: *const u64, density_len: usize, job: *mut *mut BhMsmJob) -> c_int;
    pub fn bh_msm_async_dev(ctx: *mut BhCtx, bases: *const BhBases, skip: usize, scalars_dev: *const c_void, n_scalars: usize, scalar_fmt: c_int, density_words_dev: *const u64, density_len: usize, job: *mut *mut BhMsmJob) -> c_int;
    pub fn bh_msm_wait(job: *mut BhMsmJob, out_affine: *mut c_void) -> c_int;
    pub fn bh_msm_wait_timed(job: *mut BhMsmJob, out_affine: *mut c_void, device_ms: *mut f32) -> c_int;
    pub fn bh_msm_wait_profile(job: *mut BhMsmJob, out_affine: *mut c_void, stage_ms4: *mut f32) -> c_int;
    pub fn bh_msm_wait_stats(job: *mut BhMsmJob, out_affine: *mut c_void, stage_ms4: *mut f32, stats8: *mut u64) -> c_int;
    pub fn bh_msm_plan_info(n: usize, group: c_int, forced_c: c_uint, out9: *mut c_uint) -> c_int;
    pub fn bh_msm_debug_stages(ctx: *mut BhCtx, scalars_host: *const c_void, n: usize, scalar_fmt: c_int, c: c_uint, pairs_out_host: *mut u64, zstart_out_host: *mut u32) -> c_int;
    pub fn bh_point_add(group: c_int, r: *mut c_void, a: *const c_void, b: *const c_void, n: usize);
    pub fn bh_point_mul(group: c_int, r: *mut c_void, a: *const c_void, k_canonical: *const c_void);
    pub fn bh_point_lincomb(group: c_int, r: *mut c_void, points: *const c_void, scalars_canonical: *const c_void, n: usize);
    pub fn bh_msm_async_opts(ctx: *mut BhCtx, bases: *const BhBases, skip: usize, scalars_host: *const c_void, n_scalars: usize, scalar_fmt: c_int, density_words: *const u64, density_len: usize, opts: *const BhMsmOpts, job: *mut *mut BhMsmJob) -> c_int;
    pub fn bh_msm_async_dev_opts(ctx: *mut BhCtx, bases: *const BhBases, skip: usize, scalars_dev: *const c_void, n_scalars: usize, scalar_fmt: c_int, density_words_dev: *const u64, density_len: usize, opts: *const BhMsmOpts, job: *mut *mut BhMsmJob) -> c_int;
    pub fn bh_msm_start(job: *mut BhMsmJob) -> c_int;
    pub fn bh_msm_async_dev_after(ctx: *mut BhCtx, bases: *const BhBases, skip: usize, scalars_dev: *const c_void, n_scalars: usize, scalar_fmt: c_int, density_words_dev: *const u64, density_len: usize, opts: *const BhMsmOpts, after_stream: *mut c_void, job: *mut *mut BhMsmJob) -> c_int;
    pub fn bh_scalars_register(ctx: *mut BhCtx, scalars_host: *const c_void, n: usize, scalar_fmt: c_int, out: *mut *mut BhScalars) -> c_int;
    pub fn bh_scalars_adopt_dev(ctx: *mut BhCtx, scalars_dev: *mut c_void, n: usize, scalar_fmt: c_int, take_ownership: c_int, out: *mut *mut BhScalars) -> c_int;
    pub fn bh_scalars_release(s: *mut BhScalars);
    pub fn bh_scalars_len(s: *const BhScalars) -> usize;
    pub fn bh_scalars_dev_ptr(s: *const BhScalars) -> *const c_void;
    pub fn bh_msm_async_scalars(ctx: *mut BhCtx, bases: *const BhBases, skip: usize, scalars: *const BhScalars, first: usize, n: usize, density_words: *const u64, density_len: usize, opts: *const BhMsmOpts, job: *mut *mut BhMsmJob) -> c_int;
    pub fn bh_h_poly_fr_scalars(ctx: *mut BhCtx, a_host: *const c_void, b_host: *const c_void, c_host: *const c_void, n_evals: usize, h_out: *mut *mut BhScalars) -> c_int;
    pub fn bh_msm_sharded_async(ctxs: *const *mut BhCtx, shards: *const *const BhBases, n_shards: usize, skip: usize, scalars_host: *const c_void, n_scalars: usize, scalar_fmt: c_int, density_words: *const u64, density_len: usize, job: *mut *mut BhMsmShardedJob) -> c_int;
    pub fn bh_msm_sharded_wait(job: *mut BhMsmShardedJob, out_affine: *mut c_void) -> c_int;
    pub fn bh_fixed_base_mul_dev(ctx: *mut BhCtx, group: c_int, base_affine_host: *const c_void, scalars_dev: *const c_void, n: usize, scalar_fmt: c_int, out_dev: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bh_groth16_params_create(ctx: *mut BhCtx, alpha_g1: *const c_void, beta_g1: *const c_void, beta_g2: *const c_void, delta_g1: *const c_void, delta_g2: *const c_void, h: *const c_void, nh: usize, l: *const c_void, nl: usize, a: *const c_void, na: usize, b_g1: *const c_void, nb1: usize, b_g2: *const c_void, nb2: usize, out: *mut *mut BhParams) -> c_int;
    pub fn bh_groth16_params_read(ctx: *mut BhCtx, bytes: *const c_void, len: usize, checked: c_int, out: *mut *mut BhParams) -> c_int;
    pub fn bh_groth16_generate(ctx: *mut BhCtx, r1cs: *mut BhR1cs, g1: *const c_void, g2: *const c_void, alpha: *const c_void, beta: *const c_void, gamma: *const c_void, delta: *const c_void, tau: *const c_void, out: *mut *mut BhParams) -> c_int;
    pub fn bh_groth16_params_write(p: *const BhParams, buf: *mut c_void, cap: usize, len: *mut usize) -> c_int;
    pub fn bh_groth16_params_vk_ext(p: *const BhParams, gamma_g2: *mut c_void, ic_out: *mut c_void, ic_cap: usize, n_ic: *mut usize) -> c_int;
    pub fn bh_groth16_params_query(p: *const BhParams, which: c_int, bases: *mut *const BhBases, len: *mut usize) -> c_int;
    pub fn bh_groth16_params_vk(p: *const BhParams, alpha_g1: *mut c_void, beta_g1: *mut c_void, beta_g2: *mut c_void, delta_g1: *mut c_void, delta_g2: *mut c_void) -> c_int;
    pub fn bh_proof_write(proof_affine: *const c_void, out192: *mut c_void);
    pub fn bh_groth16_params_release(p: *mut BhParams);
    pub fn bh_groth16_prove_assignment(params: *mut BhParams, a_evals: *const c_void, b_evals: *const c_void, c_evals: *const c_void, n_constraints: usize, input_assignment: *const c_void, n_inputs: usize, aux_assignment: *const c_void, n_aux: usize, a_aux_density: *const u64, b_input_density: *const u64, b_aux_density: *const u64, r: *const c_void, s: *const c_void, proof_out: *mut c_void, timings4: *mut f32) -> c_int;
    pub fn bh_r1cs_create(ctx: *mut BhCtx, n_inputs: usize, n_aux: usize, n_constraints: usize, abc: *const BhCsr, coeffs: *const c_void, n_coeffs: usize, out: *mut *mut BhR1cs) -> c_int;
    pub fn bh_r1cs_release(r: *mut BhR1cs);
    pub fn bh_r1cs_shape(r: *const BhR1cs, n_inputs: *mut usize, n_aux: *mut usize, n_constraints: *mut usize) -> c_int;
    pub fn bh_r1cs_density(r: *const BhR1cs, which: c_int, dev_words: *mut *const u64, host_words: *mut *const u64, total: *mut usize) -> c_int;
    pub fn bh_r1cs_eval_dev(ctx: *mut BhCtx, r: *const BhR1cs, inputs_dev: *const c_void, aux_dev: *const c_void, a_dev: *mut c_void, b_dev: *mut c_void, c_dev: *mut c_void, log_m: u32, stream: *mut c_void) -> c_int;
    pub fn bh_fr_powers_dev(ctx: *mut BhCtx, out_dev: *mut c_void, n: usize, g_host: *const c_void, scale_host: *const c_void, stream: *mut c_void) -> c_int;
    pub fn bh_r1cs_eval_transposed_dev(ctx: *mut BhCtx, r: *mut BhR1cs, lagrange_dev: *const c_void, at_dev: *mut c_void, bt_dev: *mut c_void, ct_dev: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn bh_fr_qap_ext_dev(ctx: *mut BhCtx, e_dev: *mut c_void, at_dev: *const c_void, bt_dev: *const c_void, ct_dev: *const c_void, n_inputs: usize, n_vars: usize, alpha: *const c_void, beta: *const c_void, gamma_inv: *const c_void, delta_inv: *const c_void, stream: *mut c_void) -> c_int;
    pub fn bh_groth16_prove_witness(params: *mut BhParams, r1cs: *const BhR1cs, input_assignment: *const c_void, n_inputs: usize, aux_assignment: *const c_void, n_aux: usize, r: *const c_void, s: *const c_void, proof_out: *mut c_void, timings4: *mut f32) -> c_int;
    pub fn bh_groth16_prove_assignment_async(params: *mut BhParams, a_evals: *const c_void, b_evals: *const c_void, c_evals: *const c_void, n_constraints: usize, input_assignment: *const c_void, n_inputs: usize, aux_assignment: *const c_void, n_aux: usize, a_aux_density: *const u64, b_input_density: *const u64, b_aux_density: *const u64, r: *const c_void, s: *const c_void, job: *mut *mut BhProofJob) -> c_int;
    pub fn bh_groth16_prove_witness_async(params: *mut BhParams, r1cs: *const BhR1cs, input_assignment: *const c_void, n_inputs: usize, aux_assignment: *const c_void, n_aux: usize, r: *const c_void, s: *const c_void, job: *mut *mut BhProofJob) -> c_int;
    pub fn bh_groth16_proof_wait(job: *mut BhProofJob, proof_out: *mut c_void, timings4: *mut f32) -> c_int;
    pub fn bh_groth16_prove_witness_part(params: *mut BhParams, r1cs: *const BhR1cs, input_assignment: *const c_void, n_inputs: usize, aux_assignment: *const c_void, n_aux: usize, part: usize, parts: usize, sums_out: *mut c_void, timings4: *mut f32) -> c_int;
    pub fn bh_groth16_sums_add(acc: *mut c_void, other: *const c_void);
    pub fn bh_groth16_assemble(params: *mut BhParams, sums: *const c_void, r: *const c_void, s: *const c_void, proof_out: *mut c_void) -> c_int;
}
