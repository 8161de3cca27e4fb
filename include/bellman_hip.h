/* libbellman_hip - C ABI of the MI355X (gfx950) implementation of bellman's Groth16
 * proving hot path.  Plain pointers and sizes only; no exceptions cross this boundary.
 *
 * The reference (zkcrypto/bellman @ 2024-08-07) has no FFI of its own: the boundary it
 * exposes is a set of generic Rust signatures.  Each entry point below names the reference
 * interface it replaces (file:line under /root/reference); INTEGRATION.md shows the Rust
 * `extern "C"` block + shim a maintainer would add behind those signatures.
 *
 * DATA FORMATS (identical to `bls12_381 0.8.0` in-memory values on a little-endian host)
 *   Fr element   32 B  4 x u64 little-endian limbs.  FFT/polynomial data is in MONTGOMERY
 *                      form (R = 2^256), exactly the bytes of a Rust `bls12_381::Scalar`.
 *   MSM scalar   32 B  either canonical little-endian (BH_SCALARS_CANONICAL - the bits of
 *                      `Exponent::Bits`, src/multiexp.rs:179; a value >= q, which the reference
 *                      cannot produce, is taken mod q) or Montgomery (BH_SCALARS_MONT -
 *                      a Rust `Scalar` as is; converted on the device).
 *   G1 affine    96 B  x | y, each 6 x u64 LE limbs, Montgomery form (R = 2^384).
 *   G2 affine   192 B  x.c0 | x.c1 | y.c0 | y.c1 (Fp2 = c0 + c1*u).
 *                      The point at infinity is the ALL-ZERO record ((0,0) is not on either
 *                      curve).  bh_bases_register_* can translate `infinity` flag bytes.
 *
 * ERROR CODES map 1:1 onto bellman::SynthesisError (src/lib.rs:303-319):
 */
#ifndef BELLMAN_HIP_H
#define BELLMAN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BH_OK 0
#define BH_ERR_UNEXPECTED_IDENTITY 1 /* SynthesisError::UnexpectedIdentity   src/multiexp.rs:63-65   */
#define BH_ERR_UNEXPECTED_EOF 2      /* SynthesisError::IoError(UnexpectedEof) src/multiexp.rs:55-61,74-80 */
#define BH_ERR_DEGREE_TOO_LARGE 3    /* SynthesisError::PolynomialDegreeTooLarge src/domain.rs:57-59 */
#define BH_ERR_UNCONSTRAINED_VARIABLE 5 /* SynthesisError::UnconstrainedVariable  groth16/src/generator.rs:464-470 */
#define BH_ERR_INVALID_POINT 6       /* io::ErrorKind::InvalidData "invalid G1" / "invalid G2" groth16/src/lib.rs:300-304,326-330 */
#define BH_ERR_POINT_AT_INFINITY 7   /* io::ErrorKind::InvalidData "point at infinity"         groth16/src/lib.rs:306-315,332-341 */
#define BH_ERR_HIP (-1)              /* HIP runtime failure (message on stderr) */
#define BH_ERR_INVALID_ARG (-2)      /* the reference would panic (e.g. density length mismatch,
                                        src/multiexp.rs:324-329; length mismatch src/domain.rs:155,174) */
#define BH_ERR_NO_DEVICE (-3)        /* no gfx950 device / kernels unavailable: there is NO CPU fallback */

#define BH_SCALARS_CANONICAL 0
#define BH_SCALARS_MONT 1

#define BH_G1 1
#define BH_G2 2

typedef struct bh_ctx bh_ctx;         /* one per (process, GPU); thread-safe */
typedef struct bh_bases bh_bases;     /* device-resident, immutable base vector (the CRS queries) */
typedef struct bh_msm_job bh_msm_job; /* an MSM in flight == bellman's Waiter<Result<G,_>> */

/* ---- context: replaces multicore::Worker::new() (src/multicore.rs:24-27) ------------- */
int bh_ctx_create(int device, bh_ctx **out);
/* Waits for the device, then frees the context's pools, streams and tables.  Jobs must have been waited on.  Base
 * handles (bh_bases) may be released before or AFTER this call: a handle that outlives its context keeps its own
 * device memory until bh_bases_release and can no longer be used for a multiexp. */
void bh_ctx_destroy(bh_ctx *ctx);
/* Worker::log_num_threads analogue (src/multicore.rs:29-31): log2 of the CU count. */
uint32_t bh_ctx_log_num_cus(const bh_ctx *ctx);
const char *bh_version(void);
/* Process-level runtime settings, to be called BEFORE the process makes its first HIP call (the HIP runtime reads
 * them when it initialises): asks for 16 hardware queues (GPU_MAX_HW_QUEUES, unless the variable is already set) -
 * a proof keeps 6-7 job streams in flight and the runtime's default of 4 queues serialises them.  Returns 1 when no
 * HIP call had been made through this library yet, 0 when it is (probably) too late to take effect.  The library
 * never changes the environment on its own. */
int bh_runtime_configure(void);
/* Limits of a context (0 / (size_t)-1 = leave unchanged):
 *   max_jobs_in_flight  multiexps issued and not yet completed.  At the limit bh_msm_async* completes the OLDEST job
 *                       on the calling thread before issuing (its result is kept for its bh_msm_wait) - the analogue
 *                       of Worker::compute running a task inline once 4 x threads are pending
 *                       (src/multicore.rs:47-73).  Default: BELLMAN_HIP_MAX_JOBS, else from the device's memory.
 *   pool_cap_bytes      cap on the device memory the context's workspace pool holds (0 = none; BELLMAN_HIP_POOL_CAP_MB).
 *                       An allocation that does not fit first returns idle blocks to the driver, then completes jobs in
 *                       flight as above, and only fails (BH_ERR_HIP) when nothing is left to wait for.  The same
 *                       happens when hipMalloc itself fails.
 *   table_budget_bytes  total size of the window tables built AUTOMATICALLY at registration (bh_bases_register etc.;
 *                       default a quarter of the device's memory, BELLMAN_HIP_TABLE_BUDGET_MB); bh_ctx_trim drops them.
 *   fft_table_budget_bytes  total size of the cached per-size FFT tables (default an eighth of the device's memory,
 *                       BELLMAN_HIP_FFT_TABLE_BUDGET_MB).  From 2^12 to 2^24 points a domain size caches n-entry
 *                       twiddle / coset tables (32 bytes per entry; up to 4 tables at two passes, 6 at three: 0.5 GB at
 *                       2^22, 3.2 GB at 2^24); a size whose complete set exceeds the budget runs on its small two-level
 *                       tables instead (two products per factor, same results), and a table that does not fit beside
 *                       the others first drops the least recently used other sizes.  bh_ctx_trim drops them all. */
int bh_ctx_set_limits(bh_ctx *ctx, uint32_t max_jobs_in_flight, size_t pool_cap_bytes, size_t table_budget_bytes,
                      size_t fft_table_budget_bytes);
typedef struct {
  int32_t device;
  uint32_t num_cus;
  uint64_t hbm_bytes;
  uint32_t hw_queues_requested;            /* GPU_MAX_HW_QUEUES when the context was created (0 = unset: runtime default, 4) */
  uint32_t hw_queues_set_before_hip_init;  /* 1 = set before this library's first HIP call (bh_runtime_configure or the
                                              caller's environment): the request can have taken effect */
  uint32_t max_jobs_in_flight, jobs_in_flight;
  uint64_t pool_bytes_held, pool_bytes_idle, table_bytes, table_budget;
  uint64_t fft_table_bytes, fft_table_budget;   /* the cached per-size FFT tables (bh_ctx_set_limits) */
} bh_ctx_info_t;
int bh_ctx_info(bh_ctx *ctx, bh_ctx_info_t *info);

/* ---- raw device memory helpers (so callers can keep vectors resident in HBM) -------- */
int bh_dev_alloc(bh_ctx *ctx, size_t bytes, void **dev_ptr);
int bh_dev_free(bh_ctx *ctx, void *dev_ptr);
int bh_dev_upload(bh_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);
int bh_dev_download(bh_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);
int bh_dev_zero(bh_ctx *ctx, void *dev_ptr, size_t bytes);   /* hipMemsetAsync(0) on the context stream */
/* caller-owned streams (opaque hipStream_t) so that independent proofs issued from different host
 * threads do not serialise on the context stream; the *_on variants enqueue and return */
int bh_stream_create(bh_ctx *ctx, void **stream);
/* high != 0: a stream of the device's highest priority - for short work on a proof's critical path (the h block, whose
 * result the H multiexp waits for) that must not queue behind another job's long-running kernels */
int bh_stream_create_priority(bh_ctx *ctx, int high, void **stream);
int bh_stream_destroy(bh_ctx *ctx, void *stream);
int bh_stream_synchronize(bh_ctx *ctx, void *stream);
int bh_dev_upload_on(bh_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes, void *stream);
int bh_dev_zero_on(bh_ctx *ctx, void *dev_ptr, size_t bytes, void *stream);
int bh_ctx_synchronize(bh_ctx *ctx);
/* Scheduling hint.  Bucket accumulations that fill the chip run one after the other in issue order (two of them side by
 * side take twice as long each and every job's latency-bound tail ends up at the end).  This call makes the NEXT such
 * accumulation - and through that chain every later one - start after everything enqueued on `stream` so far:
 * create_proof puts its h block (a short chain of FFT passes that would otherwise crawl behind the accumulations' long
 * workgroups, with the H multiexp waiting for it) in front of its multiexps' accumulations this way, while their digit
 * and sort stages still run beside it.  Results never depend on it. */
int bh_ctx_accumulations_after(bh_ctx *ctx, void *stream);
/* returns the context's idle cached device memory (recycled job workspaces, FFT twiddle tables) to the
 * driver; waits for the device to be idle first.  Purely a memory-footprint control; call it while no
 * other thread is inside a call on this context. */
int bh_ctx_trim(bh_ctx *ctx);

/* ---- EvaluationDomain (src/domain.rs) ------------------------------------------------
 * mode: 0 fft (:81-83)  1 ifft (:85-99)  2 coset_fft (:115-118)  3 icoset_fft (:120-125).
 * `data` holds 2^log_n Montgomery Fr (the padded `coeffs` of from_coeffs, :47-79), natural
 * order in and out, transformed in place.  log_n >= 32 -> BH_ERR_DEGREE_TOO_LARGE (:57-59). */
#define BH_FFT 0
#define BH_IFFT 1
#define BH_COSET_FFT 2
#define BH_ICOSET_FFT 3
int bh_fft_fr(bh_ctx *ctx, void *data_host, uint32_t log_n, int mode);
/* same on a device pointer, asynchronous on `stream` (a hipStream_t; NULL = context stream) */
int bh_fft_fr_dev(bh_ctx *ctx, void *data_dev, uint32_t log_n, int mode, void *stream);
/* pointwise domain ops on device vectors of n Montgomery Fr:
 *   mul_assign (:154-170)  sub_assign (:173-189)  divide_by_z_on_coset (:139-151, n = 2^log_n)
 *   distribute_powers(g) (:101-113; g = 32-byte Montgomery Fr on the HOST) */
int bh_fr_mul_assign_dev(bh_ctx *ctx, void *a_dev, const void *b_dev, size_t n, void *stream);
int bh_fr_sub_assign_dev(bh_ctx *ctx, void *a_dev, const void *b_dev, size_t n, void *stream);
int bh_fr_divide_by_z_on_coset_dev(bh_ctx *ctx, void *a_dev, uint32_t log_n, void *stream);
int bh_fr_distribute_powers_dev(bh_ctx *ctx, void *a_dev, size_t n, const void *g_host, void *stream);
/* The whole h-polynomial block of create_proof (groth16/src/prover.rs:221-240), fused:
 * a,b,c = n_evals constraint evaluations each (Montgomery Fr, HOST); writes the m-1 quotient
 * coefficients (m = next pow2 >= n_evals, :238-239) to h_out_host (Montgomery) and returns
 * m-1 in *h_len.  Runs ifft/coset_fft x3, (a*b-c)/Z in one pass, icoset_fft. */
int bh_h_poly_fr(bh_ctx *ctx, const void *a_host, const void *b_host, const void *c_host,
                 size_t n_evals, void *h_out_host, size_t *h_len);
/* device-resident variant: a,b,c are 2^log_n-element device vectors (clobbered); result in a */
int bh_h_poly_fr_dev(bh_ctx *ctx, void *a_dev, void *b_dev, void *c_dev, uint32_t log_n, void *stream);
/* the same, enqueue only (no synchronisation): `scratch_dev` = 2^log_n Fr of caller-owned workspace (may be NULL
 * up to 2^11), which like a, b, c must stay valid until `stream` has drained */
int bh_h_poly_fr_dev_on(bh_ctx *ctx, void *a_dev, void *b_dev, void *c_dev, void *scratch_dev, uint32_t log_n,
                        void *stream);

/* ---- bases: the `(Arc<Vec<G::Affine>>, usize)` SourceBuilder (src/multiexp.rs:45-86) --
 * Uploads `n` affine points once (the CRS is immutable and shared by every proof,
 * groth16/src/lib.rs:443-473).  `stride` bytes between records; coordinates (Montgomery) at
 * byte offset 0; if inf_offset >= 0 the byte at that offset != 0 marks the identity
 * (the `infinity: Choice` of bls12_381's G1Affine/G2Affine), else identity == all-zero. */
int bh_bases_register(bh_ctx *ctx, int group, const void *host_points, size_t n, size_t stride,
                      long inf_offset, bh_bases **out);
/* Same, straight from bellman's serialized CRS (`Parameters::write`, groth16/src/lib.rs:258-287):
 * `n` uncompressed points in the Zcash encoding (`to_uncompressed()`: big-endian canonical
 * coordinates, 96 B for G1 = x|y, 192 B for G2 = x.c1|x.c0|y.c1|y.c0, flag bits in the top of byte
 * 0; the infinity flag yields the identity).  Byte order, flags and the conversion to Montgomery form
 * are handled on the device.  Like `from_uncompressed_unchecked` (lib.rs:296-300) no on-curve or
 * subgroup check is made; a compressed-form flag returns BH_ERR_INVALID_ARG. */
int bh_bases_register_uncompressed(bh_ctx *ctx, int group, const void *host_bytes, size_t n,
                                   bh_bases **out);
/* `from_uncompressed` / `from_uncompressed_unchecked` + the identity test of Parameters::read
 * (groth16/src/lib.rs:289-341) for `n` uncompressed points, all on the device.  Without flags only
 * the encoding rules are enforced (flag bits, coordinates < p, a clean infinity encoding);
 * BH_POINTS_CHECKED adds the on-curve and prime-order-subgroup tests, BH_POINTS_FORBID_IDENTITY the
 * "point at infinity" rule.  On BH_ERR_INVALID_POINT / BH_ERR_POINT_AT_INFINITY *bad_index (optional)
 * is the first offending point in stream order, which is the one the reference reports. */
#define BH_POINTS_CHECKED 1u
#define BH_POINTS_FORBID_IDENTITY 2u
int bh_bases_read_uncompressed(bh_ctx *ctx, int group, const void *host_bytes, size_t n, unsigned flags,
                               bh_bases **out, size_t *bad_index);
/* copies `count` device-resident affine records (Montgomery) starting at `first` back to the host */
int bh_bases_download(bh_ctx *ctx, const bh_bases *b, size_t first, size_t count, void *out_host);
/* the same range in the uncompressed encoding Parameters::write emits (groth16/src/lib.rs:258-287; 96 / 192 bytes per
 * point, big-endian coordinates, G2 c1 before c0, identity = flag byte 0x40): encoded on the device, count * 96|192 bytes
 * to out_host_bytes - the inverse of bh_bases_read_uncompressed */
int bh_bases_write_uncompressed(bh_ctx *ctx, const bh_bases *bases, size_t first, size_t count, void *out_host_bytes);
/* Window table for a registered base vector (optional; the CRS is fixed across proofs, groth16/src/lib.rs
 * :443-473 hands out the same Arc<Vec<Affine>> every time): stores 2^(c*j) P_i for every window j next to
 * the bases (W = ceil(256/c) rows, W x the memory).  Multiexps over such bases send every digit of a
 * scalar to ONE bucket set - one bucket reduction instead of W and no doubling chain over windows.
 * The result of a multiexp is the same group element either way.  window_bits = 0 picks the tuned
 * value for the vector length.  Registration (bh_bases_register / _uncompressed / bh_bases_copy_dev) builds this
 * table by itself for G1 vectors of up to 2^24 points and G2 vectors of up to 2^22 while the context's table
 * budget lasts (bh_ctx_set_limits): G1 vectors from 2^19 points take 20-bit rows - 13 rows, so a multiexp
 * performs 13 n additions instead of the 16 n of the classic 16-window plan - at 13 x 128 bytes per point (1.7 GB
 * per 2^20 points; built in ~0.13 s); G2 vectors keep 16-bit rows (20-bit ones on request: faster on uniform
 * scalars from 2^20 points, slower on boolean-heavy ones - their 2^19 buckets cost ~2 ms to reduce).  BH_ERR_HIP when the table does not fit in HBM (the handle stays usable
 * without it); BH_ERR_INVALID_ARG when W x n >= 2^31. */
int bh_bases_precompute(bh_ctx *ctx, bh_bases *b, unsigned window_bits);
int bh_bases_table_info(const bh_bases *b, unsigned *window_bits, unsigned *rows, size_t *bytes);
/* a new owned handle holding a device-to-device copy of `n` packed records */
int bh_bases_copy_dev(bh_ctx *ctx, int group, const void *dev_points, size_t n, bh_bases **out);
/* wrap an existing device array of packed 96/192-byte records (not owned): a LIVE view - nothing is copied or
 * precomputed from the buffer at wrap time (no host mirror for tiny multiexps, no automatic window table), every
 * multiexp reads the buffer as it is when the job runs; the caller orders its writes before issuing.
 * bh_bases_precompute on such a handle is an explicit SNAPSHOT of the buffer's contents at that call. */
int bh_bases_wrap_dev(bh_ctx *ctx, int group, const void *dev_points, size_t n, bh_bases **out);
void bh_bases_release(bh_ctx *ctx, bh_bases *b);
size_t bh_bases_len(const bh_bases *b);

/* ---- multiexp (src/multiexp.rs:305-332) ---------------------------------------------
 * Computes  sum_i [density_i] s_i * B[skip + rank_i]   (rank_i = #dense entries before i),
 * i.e. multiexp(pool, (bases, skip), density_map, exponents):
 *   scalars        n_scalars x 32 B (HOST unless *_dev), format per scalar_fmt
 *   density_words  NULL = FullDensity (:95-115); else LSB0 bitmap, ceil(n/64) u64 words
 *                  (DensityTracker's BitVec<usize,Lsb0>, :117-131); density_len must equal
 *                  n_scalars or BH_ERR_INVALID_ARG is returned (the reference panics, :324-329)
 * Returns immediately with a job (the Waiter); bh_msm_wait blocks (Waiter::wait,
 * src/multicore.rs:100-109), writes the AFFINE result record (96/192 B, Montgomery; all-zero =
 * identity) and returns BH_OK / BH_ERR_UNEXPECTED_IDENTITY / BH_ERR_UNEXPECTED_EOF with the
 * reference's precedence (highest failing window wins, :295-300).  Many jobs may be in
 * flight per context (create_proof issues 8, groth16/src/prover.rs:244-318). */
int bh_msm_async(bh_ctx *ctx, const bh_bases *bases, size_t skip, const void *scalars_host,
                 size_t n_scalars, int scalar_fmt, const uint64_t *density_words,
                 size_t density_len, bh_msm_job **job);
int bh_msm_async_dev(bh_ctx *ctx, const bh_bases *bases, size_t skip, const void *scalars_dev,
                     size_t n_scalars, int scalar_fmt, const uint64_t *density_words_dev,
                     size_t density_len, bh_msm_job **job);
int bh_msm_wait(bh_msm_job *job, void *out_affine);
/* device time of the job's kernels in milliseconds (hipEvents on the job's stream; needs BH_MSM_STAGE_TIMES) */
int bh_msm_wait_timed(bh_msm_job *job, void *out_affine, float *device_ms);
/* as above with per-stage device times (hipEvents on the job's stream), milliseconds:
 * [0] whole pipeline  [1] digits + radix sort + task list  [2] bucket accumulation  [3] reductions;
 * zeros unless the job was issued with BH_MSM_STAGE_TIMES in bh_msm_opts.flags */
int bh_msm_wait_profile(bh_msm_job *job, void *out_affine, float *stage_ms4);
/* ... and with what the job EXECUTED, counted on the device (every job; no flag needed):
 * [0] sorted (digit, base) entries = windows x terms   [1] zero digits among them (skipped after the sort)
 * [2] mixed additions the accumulate launch performed into a non-empty accumulator - the entries that open a bucket or a
 *     chunk partial are copies (src/multiexp.rs:256-258 adds into an identity bucket likewise)
 * [3] chunk lanes of the accumulate launch   [4] window bits c   [5] chunk length K   [6] bucket sets W (1 = window-table
 * plan)   [7] digit columns per scalar.  What bench.py's roofline.alu and its check of the PMC files are computed from. */
int bh_msm_wait_stats(bh_msm_job *job, void *out_affine, float *stage_ms4, uint64_t *stats8);
/* The plan a multiexp of n terms over unregistered bases would run (host only, no context): out9 = c, windows, buckets
 * per window, K, chunks per window, sort passes, lo bits, hi bits of the two-dimensional bucket reduction, low 32 bits of
 * windows x n.  forced_c = 0: the tuned window size. */
int bh_msm_plan_info(size_t n, int group, unsigned forced_c, unsigned *out9);
/* Verification aid: runs the digit and sort stages of a G1 multiexp for window size c on n host scalars (src/multiexp.rs:
 * 159-208, 281-286: Exponent, chunks) and copies the sorted (|digit| << 32 | sign << 31 | index) pairs [windows * n] and
 * the per-window number of zero digits [windows] to the host. */
int bh_msm_debug_stages(bh_ctx *ctx, const void *scalars_host, size_t n, int scalar_fmt, unsigned c,
                        uint64_t *pairs_out_host, uint32_t *zstart_out_host);
/* r[i] = a[i] + b[i] for affine records on the HOST - used to fold the per-GPU partial results of
 * a base-sharded MSM after the all-gather, and g_a/g_b/g_c in create_proof. */
void bh_point_add(int group, void *r, const void *a, const void *b, size_t n);
/* r = [k] a on the HOST, k = 32-byte canonical little-endian scalar: the five single scalar
 * multiplications of create_proof (groth16/src/prover.rs:326-338, 342, 351) */
void bh_point_mul(int group, void *r, const void *a, const void *k_canonical);
/* r = sum_i [k_i] points[i] on the HOST (n affine records, n x 32-byte canonical scalars; scalars = NULL: all ones) with
 * one shared doubling chain and one inversion: g_c = ... + [s] a_answer + [r] b1_answer + h + l of create_proof
 * (groth16/src/prover.rs:339-354) in one call */
void bh_point_lincomb(int group, void *r, const void *points, const void *scalars_canonical, size_t n);
/* Per-job plan overrides (experiments, tests, tuning sweeps): NULL or all-zero = the tuned defaults.
 * They travel with the job, so concurrent jobs on one context never see each other's settings.  The
 * result is the same group element whatever the plan. */
typedef struct {
  uint32_t window_bits; /* c, 2..24; 0 = automatic */
  uint32_t chunk;       /* K: sorted entries per accumulation lane; 0 = automatic */
  uint32_t flags;       /* BH_MSM_* below */
} bh_msm_opts;
#define BH_MSM_ACC_REGISTERS 1u /* keep the running bucket sum in registers */
#define BH_MSM_ACC_LDS 2u       /* ... in LDS */
#define BH_MSM_NO_TABLE 4u      /* ignore a window table attached to the bases */
#define BH_MSM_NO_SMALL_PATH 8u /* run the full pipeline even for a handful of terms */
#define BH_MSM_STAGE_TIMES 64u  /* record the per-stage HIP events bh_msm_wait_profile reports (4 extra API calls) */
#define BH_MSM_HOLD 128u        /* enqueue only the digit / sort stage; bh_msm_start (or the job's wait) enqueues the rest */
#define BH_MSM_G2_SINGLE_LANE 16u /* G2: force the one-lane-per-point kernels (default: merge + reduction above 2^17 buckets) */
#define BH_MSM_G2_LANE_TRIPLES 32u /* G2: force the lane-triple (Karatsuba) kernels (default below 2^15 terms) */
#define BH_MSM_G2_LANE_PAIRS 256u /* G2: force the lane-pair kernel for the bucket accumulation (default from 2^15 terms) */
int bh_msm_async_opts(bh_ctx *ctx, const bh_bases *bases, size_t skip, const void *scalars_host,
                      size_t n_scalars, int scalar_fmt, const uint64_t *density_words,
                      size_t density_len, const bh_msm_opts *opts, bh_msm_job **job);
int bh_msm_async_dev_opts(bh_ctx *ctx, const bh_bases *bases, size_t skip, const void *scalars_dev,
                          size_t n_scalars, int scalar_fmt, const uint64_t *density_words_dev,
                          size_t density_len, const bh_msm_opts *opts, bh_msm_job **job);
/* Two-phase issue (scheduling only; results never depend on it): a job issued with BH_MSM_HOLD enqueues its digit and
 * sort stages and stops; bh_msm_start enqueues bucket accumulation, reductions and the result copy.  Chip-filling
 * accumulations run in the order they are started, and the first one started waits for the sort stages of every held
 * job issued before it: create_proof issues its multiexps held and starts them longest first, so that no sort runs
 * beside an accumulation.  bh_msm_wait on a job that was never started starts it. */
int bh_msm_start(bh_msm_job *job);

/* as bh_msm_async_dev_opts, with the job ORDERED AFTER everything enqueued so far on `after_stream` (a stream of
 * bh_stream_create, or any hipStream_t of this device): the scalars may still be being produced there - create_proof's
 * H multiexp consumes the h block's coefficients (groth16/src/prover.rs:221-245) without the host waiting in between */
int bh_msm_async_dev_after(bh_ctx *ctx, const bh_bases *bases, size_t skip, const void *scalars_dev, size_t n_scalars,
                           int scalar_fmt, const uint64_t *density_words_dev, size_t density_len, const bh_msm_opts *opts,
                           void *after_stream, bh_msm_job **job);

/* ---- scalar vectors resident in HBM ------------------------------------------------------------------
 * create_proof hands the same `Arc<Vec<Exponent>>` to several multiexps (groth16/src/prover.rs:267,279,285,300,306,
 * 316,318: input_assignment to three, aux_assignment to four): registered once, uploaded once.  scalar_fmt as for
 * bh_msm_async - with BH_SCALARS_MONT a Rust `Vec<Scalar>` is handed over as is and the serial Fr -> Exponent pass of
 * prover.rs:241-261 disappears (the conversion happens on the device inside the digit kernel). */
typedef struct bh_scalars bh_scalars;
int bh_scalars_register(bh_ctx *ctx, const void *scalars_host, size_t n, int scalar_fmt, bh_scalars **out);
/* the same over an existing device vector; take_ownership != 0: the vector came from bh_dev_alloc of this context and
 * is returned to it by bh_scalars_release */
int bh_scalars_adopt_dev(bh_ctx *ctx, void *scalars_dev, size_t n, int scalar_fmt, int take_ownership, bh_scalars **out);
/* release after every multiexp issued over the handle has been waited on */
void bh_scalars_release(bh_scalars *s);
size_t bh_scalars_len(const bh_scalars *s);
const void *bh_scalars_dev_ptr(const bh_scalars *s);
/* multiexp over scalars [first, first + n) of a registered vector; the density map (NULL = FullDensity) is given on
 * the HOST as for bh_msm_async and describes exactly those n scalars; opts may be NULL */
int bh_msm_async_scalars(bh_ctx *ctx, const bh_bases *bases, size_t skip, const bh_scalars *scalars, size_t first,
                         size_t n, const uint64_t *density_words, size_t density_len, const bh_msm_opts *opts,
                         bh_msm_job **job);
/* The h block of create_proof (groth16/src/prover.rs:221-245) with the result LEFT IN HBM: a, b, c = n_evals
 * constraint evaluations (Montgomery Fr, host); *h_out = the m - 1 quotient coefficients (Montgomery) as a registered
 * scalar vector, ready for the H multiexp - no download, no Fr -> Exponent pass (prover.rs:241-242), no re-upload. */
int bh_h_poly_fr_scalars(bh_ctx *ctx, const void *a_host, const void *b_host, const void *c_host, size_t n_evals,
                         bh_scalars **h_out);

/* ---- one multiexp over several GPUs of ONE process (the reference is a single process,
 * src/multicore.rs:21-92) ------------------------------------------------------------------------
 * ctxs[k] = a context on GPU k (bh_ctx_create(k)), shards[k] = the k-th contiguous piece of the base vector,
 * registered on ctxs[k]; the concatenation of the shards is the `bases` of multiexp(pool, (bases, skip), density,
 * exponents).  The call cuts the exponents where the base index skip + rank_i crosses a shard boundary (re-basing the
 * density map), issues one job per shard - each GPU runs the whole single-GPU pipeline on its piece, there is no
 * device-to-device exchange - and bh_msm_sharded_wait folds the per-shard results (96 / 192 bytes each) on the
 * host.  Result and error semantics are those of ONE multiexp over the whole vector, including "EOF vs. identity:
 * the top window's first failure wins" (src/multiexp.rs:295-300) with the window size of the whole length.  Several
 * contexts on the same device are allowed (tests on a single-GPU box). */
typedef struct bh_msm_sharded_job bh_msm_sharded_job;
int bh_msm_sharded_async(bh_ctx *const *ctxs, const bh_bases *const *shards, size_t n_shards, size_t skip,
                         const void *scalars_host, size_t n_scalars, int scalar_fmt, const uint64_t *density_words,
                         size_t density_len, bh_msm_sharded_job **job);
int bh_msm_sharded_wait(bh_msm_sharded_job *job, void *out_affine);

/* ---- fixed-base scalar multiplication (fixture / CRS generation; SURVEY §8 f4,
 * groth16/src/generator.rs:271-296,398-421): out[i] = [s_i] base, affine records on device.  Windowed like the
 * reference's (a table of d * 2^(8j) * base is built on the device in front of the multiplications: at most 32 mixed
 * additions per scalar, no doublings); the call returns when the results are in out_dev (it waits for the stream). */
int bh_fixed_base_mul_dev(bh_ctx *ctx, int group, const void *base_affine_host,
                          const void *scalars_dev, size_t n, int scalar_fmt, void *out_dev,
                          void *stream);

/* ---- groth16::create_proof (groth16/src/prover.rs:182-361) --------------------------------------
 * bh_params = `&Parameters` as ParameterSource (groth16/src/lib.rs:435-473): the verifying-key
 * elements the prover uses plus the five query vectors (h, l, a, b_g1, b_g2), registered in HBM.
 * A proof is written as affine records a (96 B) | b (192 B) | c (96 B).
 * The C++ mirror of Circuit / ConstraintSystem / LinearCombination / ProvingAssignment lives in
 * bellman_amd/csrc/groth16.hpp; these entry points expose it to other host languages. */
typedef struct bh_params bh_params;
typedef struct bh_r1cs bh_r1cs; /* constraint matrices on the device, see the R1CS section below */
int bh_groth16_params_create(bh_ctx *ctx, const void *alpha_g1, const void *beta_g1, const void *beta_g2,
                             const void *delta_g1, const void *delta_g2, const void *h, size_t nh,
                             const void *l, size_t nl, const void *a, size_t na, const void *b_g1,
                             size_t nb1, const void *b_g2, size_t nb2, bh_params **out);
/* Parameters::read(reader, checked) (groth16/src/lib.rs:289-398, with VerifyingKey::read :159-215) on
 * the serialized CRS (`Parameters::write`, :258-287): decoding, and with checked != 0 the on-curve /
 * subgroup validation of the query points, run on the device; the verifying key is always validated.
 * The first failure in stream order is returned, as the reference's sequential reader would:
 * BH_ERR_INVALID_POINT, BH_ERR_POINT_AT_INFINITY (query and ic points must not be the identity) or
 * BH_ERR_UNEXPECTED_EOF (input ends inside the structure). Trailing bytes are ignored. */
int bh_groth16_params_read(bh_ctx *ctx, const void *bytes, size_t len, int checked, bh_params **out);
/* generate_parameters (groth16/src/generator.rs:163-510) for the circuit whose matrices are `r1cs`
 * (bh_r1cs_create; the matrices include the `input_i * 0 = 0` rows, generator.rs:195-202): g1 / g2 =
 * affine generators (96 / 192 B), alpha..tau = Montgomery Fr.  Everything per-variable or
 * per-constraint runs on the device.  BH_ERR_UNEXPECTED_IDENTITY for gamma = 0 or delta = 0 (:227-243),
 * BH_ERR_UNCONSTRAINED_VARIABLE (:464-470), BH_ERR_DEGREE_TOO_LARGE. */
int bh_groth16_generate(bh_ctx *ctx, bh_r1cs *r1cs, const void *g1, const void *g2, const void *alpha,
                        const void *beta, const void *gamma, const void *delta, const void *tau,
                        bh_params **out);
/* Parameters::write (groth16/src/lib.rs:258-287).  buf == NULL: *len = bytes needed.  Parameters made by
 * bh_groth16_params_create carry no gamma_g2 / ic and write them as the identity / an empty list. */
int bh_groth16_params_write(const bh_params *p, void *buf, size_t cap, size_t *len);
/* verifier-side key elements kept by params_read / generate: gamma_g2 (192 B), ic (n_ic x 96 B) */
int bh_groth16_params_vk_ext(const bh_params *p, void *gamma_g2, void *ic_out, size_t ic_cap, size_t *n_ic);
/* which: 0 h, 1 l, 2 a, 3 b_g1, 4 b_g2 - the device-resident query (owned by the params) and its length */
int bh_groth16_params_query(const bh_params *p, int which, const bh_bases **bases, size_t *len);
/* the prover-side verifying-key elements as affine Montgomery records (any pointer may be NULL) */
int bh_groth16_params_vk(const bh_params *p, void *alpha_g1, void *beta_g1, void *beta_g2, void *delta_g1,
                         void *delta_g2);
/* Proof::write (groth16/src/lib.rs:38-46): affine a | b | c (384 B) -> compressed A (48) | B (96) | C (48) */
void bh_proof_write(const void *proof_affine, void *out192);
void bh_groth16_params_release(bh_params *p);
/* prover.rs:217-360 on the fields of a synthesised ProvingAssignment (prover.rs:57-71): a/b/c
 * evaluations (n_constraints, input constraints of :208-215 already appended), input/aux
 * assignments (Montgomery Fr), the three density bitmaps (LSB0 words), r and s (Montgomery Fr).
 * timings4 (optional): [synthesis, h block, multiexps, total] host milliseconds. */
int bh_groth16_prove_assignment(bh_params *params, const void *a_evals, const void *b_evals,
                                const void *c_evals, size_t n_constraints, const void *input_assignment,
                                size_t n_inputs, const void *aux_assignment, size_t n_aux,
                                const uint64_t *a_aux_density, const uint64_t *b_input_density,
                                const uint64_t *b_aux_density, const void *r, const void *s,
                                void *proof_out, float *timings4);
/* ---- R1CS resident in HBM: constraint evaluation as sparse matrix x witness --------
 * The reference evaluates the A/B/C linear combinations of every constraint on one host thread
 * during synthesis (groth16/src/prover.rs:19-55 `eval`, :105-145 `enforce`).  The matrices are a
 * property of the circuit (like the CRS), so they are registered once; per proof only the witness
 * is uploaded and a = A.w, b = B.w, c = C.w are computed on the device, where the h block consumes
 * them.  Row i of a matrix holds terms [row_ptr[i], row_ptr[i+1]); a term is (var, coeff): var <
 * n_inputs addresses input_assignment[var], otherwise aux_assignment[var - n_inputs]; coeff indexes
 * `coeffs` (n_coeffs Montgomery Fr, coeffs[0] must be 1).  Terms with a zero coefficient contribute
 * neither to the value nor to the query densities (prover.rs:31).  n_constraints includes the
 * `input_i * 0 = 0` rows that create_proof appends (prover.rs:208-215). */
typedef struct {
  const uint32_t *row_ptr; /* n_constraints + 1 */
  const uint32_t *var;     /* nnz */
  const uint32_t *coeff;   /* nnz */
} bh_csr;
int bh_r1cs_create(bh_ctx *ctx, size_t n_inputs, size_t n_aux, size_t n_constraints, const bh_csr abc[3],
                   const void *coeffs, size_t n_coeffs, bh_r1cs **out);
void bh_r1cs_release(bh_r1cs *r);
int bh_r1cs_shape(const bh_r1cs *r, size_t *n_inputs, size_t *n_aux, size_t *n_constraints);
/* which: 0 a_aux_density, 1 b_input_density, 2 b_aux_density (prover.rs:59-61): LSB0 words on the
 * device / on the host, and get_total_density() (multiexp.rs:154-156).  Any out pointer may be NULL. */
int bh_r1cs_density(const bh_r1cs *r, int which, const uint64_t **dev_words, const uint64_t **host_words,
                    size_t *total);
/* a/b/c[0 .. 2^log_m) <- constraint evaluations (Montgomery), zero above n_constraints
 * (EvaluationDomain::from_coeffs padding, domain.rs:68).  Asynchronous on `stream`. */
int bh_r1cs_eval_dev(bh_ctx *ctx, const bh_r1cs *r, const void *inputs_dev, const void *aux_dev,
                     void *a_dev, void *b_dev, void *c_dev, uint32_t log_m, void *stream);
/* Parameter generation (groth16/src/generator.rs:247-462) - device building blocks:
 *   bh_fr_powers_dev             out[i] = scale * g^i           (:249-263 powers of tau; :266-296 with t(tau)/delta folded in)
 *   bh_r1cs_eval_transposed_dev  at/bt/ct[v] = QAP polynomial of variable v at tau (:369-387) from the
 *                                Lagrange coefficients (ifft of the powers of tau, :299-300); n_inputs + n_aux entries each
 *   bh_fr_qap_ext_dev            e[v] = (at*beta + bt*alpha + ct) * (v < n_inputs ? 1/gamma : 1/delta)   (:400-407)
 * followed by bh_fixed_base_mul_dev for the h, a, b_g1, b_g2, ic/l points.  Scalars are Montgomery Fr. */
int bh_fr_powers_dev(bh_ctx *ctx, void *out_dev, size_t n, const void *g_host, const void *scale_host, void *stream);
int bh_r1cs_eval_transposed_dev(bh_ctx *ctx, bh_r1cs *r, const void *lagrange_dev, void *at_dev, void *bt_dev,
                                void *ct_dev, void *stream);
int bh_fr_qap_ext_dev(bh_ctx *ctx, void *e_dev, const void *at_dev, const void *bt_dev, const void *ct_dev,
                      size_t n_inputs, size_t n_vars, const void *alpha, const void *beta, const void *gamma_inv,
                      const void *delta_inv, void *stream);
/* create_proof (prover.rs:217-360) from the witness alone: input_assignment (n_inputs, [0] = 1) and
 * aux_assignment (n_aux) as produced by the circuit's alloc closures; lengths must match the R1CS. */
int bh_groth16_prove_witness(bh_params *params, const bh_r1cs *r1cs, const void *input_assignment,
                             size_t n_inputs, const void *aux_assignment, size_t n_aux, const void *r,
                             const void *s, void *proof_out, float *timings4);
/* ---- one caller, proofs back to back: create_proof split at the synthesis / device boundary -----------------
 * The reference's create_proof is synthesis on the host (groth16/src/prover.rs:182-215) followed by the device part
 * (:217-360); a single caller that proves in a loop leaves the GPU idle during every synthesis and the host idle during
 * every device part.  The _async calls return once the device part has been handed to a helper thread;
 * bh_groth16_proof_wait blocks for it (the Waiter of the whole proof), writes the proof and frees the job.  Issuing
 * proof k+1 before waiting for proof k overlaps its synthesis with proof k's GPU work - what groth16::ProofPipeline
 * (csrc/groth16.hpp) does for C++ callers.  Proofs are identical to the synchronous calls'.
 *   bh_groth16_prove_assignment_async  arguments as bh_groth16_prove_assignment; the arrays are READ IN PLACE until
 *                                      bh_groth16_proof_wait returns (a Rust host keeps its ProvingAssignment alive);
 *   bh_groth16_prove_witness_async     arguments as bh_groth16_prove_witness; the two witness vectors are copied before
 *                                      the call returns, the r1cs handle must outlive the wait. */
typedef struct bh_proof_job bh_proof_job;
int bh_groth16_prove_assignment_async(bh_params *params, const void *a_evals, const void *b_evals,
                                      const void *c_evals, size_t n_constraints, const void *input_assignment,
                                      size_t n_inputs, const void *aux_assignment, size_t n_aux,
                                      const uint64_t *a_aux_density, const uint64_t *b_input_density,
                                      const uint64_t *b_aux_density, const void *r, const void *s, bh_proof_job **job);
int bh_groth16_prove_witness_async(bh_params *params, const bh_r1cs *r1cs, const void *input_assignment,
                                   size_t n_inputs, const void *aux_assignment, size_t n_aux, const void *r,
                                   const void *s, bh_proof_job **job);
int bh_groth16_proof_wait(bh_proof_job *job, void *proof_out, float *timings4);

/* ---- one proof over several GPUs: every rank holds the CRS and the matrices, builds the
 * witness and runs the (small) h block, but computes each of the eight multiexps of prover.rs:244-318
 * only over part `part` of `parts` of the scalar indices (contiguous, cut at multiples of 64).  The
 * result is BH_MSM_SUMS_BYTES = 6 x 96 + 2 x 192 bytes: affine a_inputs, a_aux, b_g1_inputs, b_g1_aux
 * (G1), b_g2_inputs, b_g2_aux (G2), h, l (G1) - the wait order of prover.rs:339-354.  The ranks
 * all-gather these records, add them slot-wise (bh_groth16_sums_add) and every rank assembles the same
 * proof (prover.rs:326-360, bh_groth16_assemble).  parts = 1 is bh_groth16_prove_witness. */
#define BH_MSM_SUMS_BYTES 960
int bh_groth16_prove_witness_part(bh_params *params, const bh_r1cs *r1cs, const void *input_assignment,
                                  size_t n_inputs, const void *aux_assignment, size_t n_aux, size_t part,
                                  size_t parts, void *sums_out, float *timings4);
void bh_groth16_sums_add(void *acc, const void *other);
int bh_groth16_assemble(bh_params *params, const void *sums, const void *r, const void *s, void *proof_out);
#ifdef __cplusplus
}
#endif
#endif
