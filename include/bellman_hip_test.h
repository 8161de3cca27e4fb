/* libbellman_hip_test.so - self-test hooks and built-in demo circuits used by tests/, tools/ and bench.py only.  NOT part
 * of the product boundary (include/bellman_hip.h): nothing here is needed by a caller of multiexp / EvaluationDomain /
 * create_proof, and none of it is in libbellman_hip.so.  The test library links against the shipped one, so everything
 * below still runs the shipped kernels and the shipped prover (csrc/test_hooks.hip, demo_circuits.cpp,
 * groth16_callsites.cpp). */
#ifndef BELLMAN_HIP_TEST_H
#define BELLMAN_HIP_TEST_H
#include "bellman_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- built-in demo circuits: create_proof on a circuit written in C++ against the mirror (like bellman user code;
 * Python cannot define one, so bench.py and the tests reach these through ctypes):
 * kind 0 = MiMCDemo (groth16/tests/common/mod.rs; witness = xl|xr, constants = `size` Fr),
 * kind 1 = synthetic multiplicative chain of `size` rounds (witness = x0),
 * kind 2 = every form of linear combination (`size` rounds, witness = x0; not satisfiable: a fixture for the host-side
 *          hooks bh_test_demo_assignment / bh_test_capture_check, bh_groth16_prove_demo rejects it),
 * kind 3 = a circuit whose structure is drawn from `seed` (`size` rounds, witness = x0; same use as kind 2),
 * kind 5 = boolean-heavy bit mixing in the shape of src/gadgets/boolean.rs (64 state bits from the low word of the witness
 *          x0, `size` AND / XOR steps, the state packed into a field element every 64 steps: > 98 % of the aux assignment
 *          is 0 or 1). */
int bh_groth16_prove_demo(bh_params *params, int circuit_kind, size_t size, uint64_t seed,
                          const void *witness, const void *constants, const void *r, const void *s,
                          void *proof_out, float *timings4);
/* the same split at the synthesis / device boundary (bh_groth16_prove_*_async of the product): synthesises the circuit
 * on the CALLING thread (prover.rs:182-215; with `r1cs` only the witness closures run, constraints are evaluated on the
 * device) and returns while the device part runs on a helper thread; bh_groth16_proof_wait collects it */
int bh_groth16_prove_demo_async(bh_params *params, const bh_r1cs *r1cs, int circuit_kind, size_t size, uint64_t seed,
                                const void *witness, const void *constants, const void *r, const void *s,
                                bh_proof_job **job);
int bh_groth16_prove_demo_r1cs_part(bh_params *params, const bh_r1cs *r1cs, int circuit_kind, size_t size,
                                    uint64_t seed, const void *witness, const void *constants, size_t part,
                                    size_t parts, void *sums_out, float *timings4);
/* The demo circuits of bh_groth16_prove_demo through that path: capture the matrices once ... */
int bh_groth16_demo_r1cs(bh_ctx *ctx, int circuit_kind, size_t size, uint64_t seed, const void *constants,
                         bh_r1cs **out);
/* ... then per proof run only the circuit's witness closures on the host (enforce is a no-op). */
int bh_groth16_prove_demo_r1cs(bh_params *params, const bh_r1cs *r1cs, int circuit_kind, size_t size,
                               uint64_t seed, const void *witness, const void *constants, const void *r,
                               const void *s, void *proof_out, float *timings4);

/* element-wise field / group ops on the device */
int bh_test_fr_mul_dev(bh_ctx *ctx, void *r_dev, const void *a_dev, const void *b_dev, size_t n);
int bh_test_fp_mul_dev(bh_ctx *ctx, void *r_dev, const void *a_dev, const void *b_dev, size_t n);
/* r[i] = a[i] + b[i] on the curve (affine in, affine out) */
int bh_test_point_add_dev(bh_ctx *ctx, int group, void *r_dev, const void *a_dev, const void *b_dev, size_t n);
/* the G2 group law in the lane-triple (K3) form of the MSM kernels (csrc/fp2k3.cuh): a[i] + b[i] by the general
 * addition, by the mixed addition (a[i] stays when b[i] is the identity), and 2 a[i]; n affine records each, HOST */
int bh_test_g2_k3_dev(bh_ctx *ctx, void *out_add_host, void *out_madd_host, void *out_dbl_host, const void *a_dev,
                      const void *b_dev, size_t n);
/* the general addition in the lane-SEXTET (K6) form of the merge kernels (csrc/msm_ec.cuh 5''): a[i] + b[i], affine out, HOST */
int bh_test_g2_k6_dev(bh_ctx *ctx, void *out_add_host, const void *a_dev, const void *b_dev, size_t n);
/* the same in the lane-pair form (csrc/fp2pair.cuh: schoolbook Fp2 products, one reduction per lane) */
int bh_test_g2_pairs_dev(bh_ctx *ctx, void *out_add_host, void *out_madd_host, void *out_dbl_host, const void *a_dev,
                         const void *b_dev, size_t n);
/* host-side (CPU) versions of the same arithmetic headers, for toolchain-only unit tests */
void bh_test_fr_mul_host(void *r, const void *a, const void *b, size_t n);
void bh_test_fp_mul_host(void *r, const void *a, const void *b, size_t n);
void bh_test_fr_mul_bform_host(void *r, const void *a, const void *b, size_t n); /* b pre-sliced as the FFT tables are */
void bh_test_point_add_host(int group, void *r, const void *a, const void *b, size_t n);
void bh_test_point_mul_host(int group, void *r, const void *a, const void *k_canonical);
void bh_test_fr_inv_host(void *r, const void *a, size_t n); /* Montgomery in/out */
/* host only: the scalar-index slice [lo, hi) that part `part` of `parts` of a sharded proof computes */
void bh_test_proof_slice(size_t n, size_t part, size_t parts, size_t *lo, size_t *hi);
/* lazily reduced Fp helpers of the curve kernels, host build: op 0 add, 1 sub, 2 neg, 3 canonicalise,
 * 4 is_zero (returned), 5 product, 6 square, 7 eq (returned); operands are 48-byte values in [0, 2p) */
int bh_test_fp_lazy_host(int op, void *r, const void *a, const void *b);
/* host-only: milliseconds to synthesise a demo circuit (kind/size/seed as bh_groth16_prove_demo) into a
 * ProvingAssignment (mode 0) or a WitnessAssignment (mode 1); modes 2 / 3: the same into a recycled (cleared, capacity
 * kept) assignment, as create_proof does from the second proof on; no device involved */
double bh_test_synthesis_ms(int circuit_kind, size_t size, uint64_t seed, int mode);
/* host only: captures the constraint matrices of a demo circuit (what R1cs / bh_groth16_demo_r1cs does before the upload)
 * and checks them against the ProvingAssignment of the same circuit; out4 = [constraints, terms, coefficient-table
 * entries, rows that differ]; returns the capture time in ms (negative on failure) */
double bh_test_capture_check(int circuit_kind, size_t size, uint64_t seed, size_t out4[4]);
/* host only: the ProvingAssignment create_proof synthesises for a demo circuit (arguments as bh_groth16_prove_demo; input
 * constraints of prover.rs:208-215 appended).  counts3 = [n_constraints, n_inputs, n_aux]; with a == NULL only the counts
 * are returned; otherwise a, b, c (n_constraints Fr), inputs, aux (Montgomery Fr) and the three LSB0 density bitmaps */
int bh_test_demo_assignment(int circuit_kind, size_t size, uint64_t seed, const void *witness, const void *constants,
                            size_t counts3[3], void *a, void *b, void *c, void *inputs, void *aux, uint64_t *a_aux_density,
                            uint64_t *b_input_density, uint64_t *b_aux_density);
void bh_test_fr_from_u512_host(void *r, const void *limbs8); /* 64 bytes LE -> Montgomery Fr (create_random_proof's sampling) */
/* host only: the scalar-field arithmetic of the C++ mirror (bellman::Fr, csrc/groth16.hpp - what circuits and the
 * linear-combination evaluation compute with during synthesis), n operations on arrays of 32-byte Montgomery elements:
 * op 0 r = a + b, 1 r = a - b, 2 r = a * b (b may be ANY 256-bit value), 3 r = -a, 4 r = Fr::from_u64(low limb of a),
 * 5 r = canonical limbs of a (to_canonical), 6 r = a^-1 (a != 0) */
void bh_test_fr_ops_host(int op, void *r, const void *a, const void *b, size_t n);

/* host only: where bh_msm_sharded_async cuts the exponents for shards of lens[k] bases (cuts_out[n_shards + 1]), and
 * the size class the workspace pool rounds a request up to */
int bh_test_shard_cuts(const size_t *lens, size_t n_shards, size_t skip, const uint64_t *density_words, size_t n_scalars,
                       size_t *cuts_out);
size_t bh_test_pool_size_class(size_t bytes);
/* create_proof's h block + eight multiexps issued as the reference's call sites would issue them through the Rust shim
 * (shim/patches/bellman-hip.patch), transcribed in C++ (csrc/groth16_callsites.cpp):
 *   mode 1  groth16/src/prover.rs patched: bh_scalars_register x2, bh_msm_async_scalars x8, bh_h_poly_fr_scalars
 *   mode 2  only multiexp.rs / domain.rs (/ hip.rs) patched: the EvaluationDomain's vector resident in HBM between its calls,
 *           Exponent::from deferred, every exponent vector registered once (bh_scalars_register), bh_msm_async_scalars x8
 *   mode 0  the round-4 form of that level: 7 x bh_fft_fr on host vectors, the pointwise passes and the
 *           Fr -> Exponent passes on the host, 8 x bh_msm_async with canonical host scalars
 * arguments as bh_groth16_prove_assignment; ms2 (optional): [issue + waits, total] host milliseconds */
int bh_test_groth16_prove_via_call_sites(bh_params *params, int mode, const void *a_evals, const void *b_evals,
                                         const void *c_evals, size_t n_constraints, const void *input_assignment,
                                         size_t n_inputs, const void *aux_assignment, size_t n_aux,
                                         const uint64_t *a_aux_density, const uint64_t *b_input_density,
                                         const uint64_t *b_aux_density, const void *r, const void *s, void *proof_out,
                                         float *ms2);

#ifdef __cplusplus
}
#endif
#endif
