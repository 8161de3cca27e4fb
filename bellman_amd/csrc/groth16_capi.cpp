// The extern "C" entry points of the groth16 layer (declared in include/bellman_hip.h).  The built-in demo circuits and
// their entry points live in the test library (demo_circuits.cpp, include/bellman_hip_test.h).
#include <string.h>

#include <memory>

#include <chrono>
#include <functional>
#include <vector>

#include "groth16_internal.hpp"


// ---------------------------------------------------------------------------------------------------
// C entry points (declared in include/bellman_hip.h)
// ---------------------------------------------------------------------------------------------------
extern "C" {

int bh_groth16_params_create(bh_ctx *ctx, const void *alpha_g1, const void *beta_g1, const void *beta_g2,
                             const void *delta_g1, const void *delta_g2, const void *h, size_t nh, const void *l,
                             size_t nl, const void *a, size_t na, const void *b_g1, size_t nb1, const void *b_g2,
                             size_t nb2, bh_params **out) {
  try {
    groth16::VerifyingKey vk;
    memcpy(&vk.alpha_g1, alpha_g1, 96); memcpy(&vk.beta_g1, beta_g1, 96); memcpy(&vk.beta_g2, beta_g2, 192);
    memcpy(&vk.delta_g1, delta_g1, 96); memcpy(&vk.delta_g2, delta_g2, 192);
    *out = new bh_params{new groth16::Parameters(ctx, vk, (const groth16::G1Affine *)h, nh, (const groth16::G1Affine *)l, nl,
                                                 (const groth16::G1Affine *)a, na, (const groth16::G1Affine *)b_g1, nb1,
                                                 (const groth16::G2Affine *)b_g2, nb2)};
    return BH_OK;
  } catch (const bellman::SynthesisError &e) { return e.code;
  } catch (...) { return BH_ERR_HIP; }
}
int bh_groth16_params_read(bh_ctx *ctx, const void *bytes, size_t len, int checked, bh_params **out) {
  if (!ctx || !out || (len && !bytes)) return BH_ERR_INVALID_ARG;
  try {
    *out = new bh_params{new groth16::Parameters(ctx, bytes, len, checked != 0)};
    return BH_OK;
  } catch (const bellman::IoError &e) { return e.code;
  } catch (const bellman::SynthesisError &e) { return e.code;
  } catch (...) { return BH_ERR_HIP; }
}
int bh_groth16_generate(bh_ctx *ctx, bh_r1cs *r1cs, const void *g1, const void *g2, const void *alpha, const void *beta,
                        const void *gamma, const void *delta, const void *tau, bh_params **out) {
  if (!ctx || !r1cs || !out) return BH_ERR_INVALID_ARG;
  using namespace groth16;
  try {
    R1csView view(r1cs);
    G1Affine p1; G2Affine p2;
    Fr f[5];
    memcpy(&p1, g1, 96); memcpy(&p2, g2, 192);
    const void *src[5] = {alpha, beta, gamma, delta, tau};
    for (int i = 0; i < 5; i++) memcpy(&f[i], src[i], 32);
    *out = new bh_params{new Parameters(ctx, view.r, p1, p2, f[0], f[1], f[2], f[3], f[4])};
    return BH_OK;
  } catch (const bellman::SynthesisError &e) { return e.code;
  } catch (...) { return BH_ERR_HIP; }
}
int bh_groth16_params_write(const bh_params *p, void *buf, size_t cap, size_t *len) {
  if (!p || !len) return BH_ERR_INVALID_ARG;
  try {
    const groth16::Parameters &P = *p->p;
    const size_t need = P.serialized_size();
    *len = need;
    if (!buf || cap < need) return buf ? BH_ERR_INVALID_ARG : BH_OK;   // size query when buf == NULL
    P.write_into((unsigned char *)buf, cap);
    return BH_OK;
  } catch (const bellman::SynthesisError &e) { return e.code;
  } catch (...) { return BH_ERR_HIP; }
}
int bh_groth16_params_vk_ext(const bh_params *p, void *gamma_g2, void *ic_out, size_t ic_cap, size_t *n_ic) {
  if (!p) return BH_ERR_INVALID_ARG;
  const groth16::VerifyingKey &vk = p->p->vk;
  if (gamma_g2) memcpy(gamma_g2, &vk.gamma_g2, 192);
  if (n_ic) *n_ic = vk.ic.size();
  if (ic_out) {
    if (ic_cap < vk.ic.size()) return BH_ERR_INVALID_ARG;
    if (!vk.ic.empty()) memcpy(ic_out, vk.ic.data(), vk.ic.size() * 96);
  }
  return BH_OK;
}
int bh_groth16_params_query(const bh_params *p, int which, const bh_bases **bases, size_t *len) {
  if (!p || which < 0 || which > 4) return BH_ERR_INVALID_ARG;
  const bh_bases *q[5] = {p->p->h, p->p->l, p->p->a, p->p->b_g1, p->p->b_g2};
  if (bases) *bases = q[which];
  if (len) *len = bh_bases_len(q[which]);
  return BH_OK;
}
int bh_groth16_params_vk(const bh_params *p, void *alpha_g1, void *beta_g1, void *beta_g2, void *delta_g1, void *delta_g2) {
  if (!p) return BH_ERR_INVALID_ARG;
  const groth16::VerifyingKey &vk = p->p->vk;
  if (alpha_g1) memcpy(alpha_g1, &vk.alpha_g1, 96);
  if (beta_g1) memcpy(beta_g1, &vk.beta_g1, 96);
  if (beta_g2) memcpy(beta_g2, &vk.beta_g2, 192);
  if (delta_g1) memcpy(delta_g1, &vk.delta_g1, 96);
  if (delta_g2) memcpy(delta_g2, &vk.delta_g2, 192);
  return BH_OK;
}
void bh_proof_write(const void *proof_affine, void *out192) {
  groth16::Proof p;
  memcpy(&p.a, proof_affine, 96);
  memcpy(&p.b, (const char *)proof_affine + 96, 192);
  memcpy(&p.c, (const char *)proof_affine + 288, 96);
  p.write((unsigned char *)out192);
}
void bh_groth16_params_release(bh_params *p) {
  if (!p) return;
  delete p->p;
  delete p;
}

int bh_groth16_prove_assignment(bh_params *params, const void *a_evals, const void *b_evals, const void *c_evals,
                                size_t n_constraints, const void *input_assignment, size_t n_inputs,
                                const void *aux_assignment, size_t n_aux, const uint64_t *a_aux_density,
                                const uint64_t *b_input_density, const uint64_t *b_aux_density, const void *r,
                                const void *s, void *proof_out, float *timings4) {
  using namespace groth16;
  if (!params || !r || !s || !proof_out) return BH_ERR_INVALID_ARG;
  if ((n_constraints && (!a_evals || !b_evals || !c_evals)) || (n_inputs && (!input_assignment || !b_input_density)) ||
      (n_aux && (!aux_assignment || !a_aux_density || !b_aux_density)))
    return BH_ERR_INVALID_ARG;
  ProveTimings tm = {0, 0, 0, 0};
  // everything that can allocate runs inside the guard: no C++ exception may cross the C boundary
  int rc = run_guarded([&] {
    // views of the caller's arrays: they are only read as bytes (uploads, the tiny input multiexps), never copied
    AssignmentView v;
    v.a = (const Fr *)a_evals; v.b = (const Fr *)b_evals; v.c = (const Fr *)c_evals; v.n_constraints = n_constraints;
    v.input_assignment = (const Fr *)input_assignment; v.n_inputs = n_inputs;
    v.aux_assignment = (const Fr *)aux_assignment; v.n_aux = n_aux;
    v.a_aux_density = a_aux_density; v.b_input_density = b_input_density; v.b_aux_density = b_aux_density;
    Fr rr, ss;
    memcpy(&rr, r, 32); memcpy(&ss, s, 32);
    return prove_assignment(v, *params->p, rr, ss, &tm);
  }, proof_out);
  if (timings4) { timings4[0] = tm.synthesis_ms; timings4[1] = tm.h_poly_ms; timings4[2] = tm.msm_ms; timings4[3] = tm.total_ms; }
  return rc;
}

int bh_groth16_prove_witness(bh_params *params, const bh_r1cs *r1cs, const void *input_assignment, size_t n_inputs,
                             const void *aux_assignment, size_t n_aux, const void *r, const void *s, void *proof_out,
                             float *timings4) {
  using namespace groth16;
  if (!params || !r1cs || !r || !s || !proof_out || (n_inputs && !input_assignment) || (n_aux && !aux_assignment))
    return BH_ERR_INVALID_ARG;
  ProveTimings tm = {0, 0, 0, 0};
  int rc = run_guarded([&] {
    R1csView view(r1cs);
    std::vector<Fr> in(n_inputs), aux(n_aux);   // caller records may be unaligned
    if (n_inputs) memcpy(in.data(), input_assignment, n_inputs * 32);
    if (n_aux) memcpy(aux.data(), aux_assignment, n_aux * 32);
    Fr rr, ss;
    memcpy(&rr, r, 32); memcpy(&ss, s, 32);
    return prove_witness(view.r, *params->p, in.data(), n_inputs, aux.data(), n_aux, rr, ss, &tm);
  }, proof_out);
  if (timings4) { timings4[0] = tm.synthesis_ms; timings4[1] = tm.h_poly_ms; timings4[2] = tm.msm_ms; timings4[3] = tm.total_ms; }
  return rc;
}

// ---- asynchronous proofs: the device part of a proof on a helper thread (groth16.hpp AsyncProof) ----------------------
int bh_groth16_prove_assignment_async(bh_params *params, const void *a_evals, const void *b_evals, const void *c_evals,
                                      size_t n_constraints, const void *input_assignment, size_t n_inputs,
                                      const void *aux_assignment, size_t n_aux, const uint64_t *a_aux_density,
                                      const uint64_t *b_input_density, const uint64_t *b_aux_density, const void *r,
                                      const void *s, bh_proof_job **out) {
  using namespace groth16;
  if (!params || !r || !s || !out) return BH_ERR_INVALID_ARG;
  if ((n_constraints && (!a_evals || !b_evals || !c_evals)) || (n_inputs && (!input_assignment || !b_input_density)) ||
      (n_aux && (!aux_assignment || !a_aux_density || !b_aux_density)))
    return BH_ERR_INVALID_ARG;
  try {
    AssignmentView v;
    v.a = (const Fr *)a_evals; v.b = (const Fr *)b_evals; v.c = (const Fr *)c_evals; v.n_constraints = n_constraints;
    v.input_assignment = (const Fr *)input_assignment; v.n_inputs = n_inputs;
    v.aux_assignment = (const Fr *)aux_assignment; v.n_aux = n_aux;
    v.a_aux_density = a_aux_density; v.b_input_density = b_input_density; v.b_aux_density = b_aux_density;
    Fr rr, ss;
    memcpy(&rr, r, 32); memcpy(&ss, s, 32);
    std::unique_ptr<bh_proof_job> pj(new bh_proof_job());
    pj->job = prove_assignment_async(v, *params->p, rr, ss);
    *out = pj.release();
    return BH_OK;
  } catch (const bellman::SynthesisError &e) { return e.code;
  } catch (const std::invalid_argument &) { return BH_ERR_INVALID_ARG;
  } catch (...) { return BH_ERR_HIP; }
}
int bh_groth16_prove_witness_async(bh_params *params, const bh_r1cs *r1cs, const void *input_assignment, size_t n_inputs,
                                   const void *aux_assignment, size_t n_aux, const void *r, const void *s, bh_proof_job **out) {
  using namespace groth16;
  if (!params || !r1cs || !r || !s || !out || (n_inputs && !input_assignment) || (n_aux && !aux_assignment))
    return BH_ERR_INVALID_ARG;
  try {
    Fr rr, ss;
    memcpy(&rr, r, 32); memcpy(&ss, s, 32);
    std::unique_ptr<bh_proof_job> pj(new bh_proof_job());
    pj->view.reset(new R1csView(r1cs));
    pj->job = prove_witness_async(pj->view->r, *params->p, input_assignment, n_inputs, aux_assignment, n_aux, rr, ss);
    *out = pj.release();
    return BH_OK;
  } catch (const bellman::SynthesisError &e) { return e.code;
  } catch (const std::invalid_argument &) { return BH_ERR_INVALID_ARG;
  } catch (...) { return BH_ERR_HIP; }
}
int bh_groth16_proof_wait(bh_proof_job *job, void *proof_out, float *timings4) {
  using namespace groth16;
  if (!job || !proof_out) return BH_ERR_INVALID_ARG;
  std::unique_ptr<bh_proof_job> pj(job);
  ProveTimings tm = {0, 0, 0, 0};
  int rc = run_guarded([&] { return pj->job->wait(&tm); }, proof_out);
  if (timings4) { timings4[0] = tm.synthesis_ms; timings4[1] = tm.h_poly_ms; timings4[2] = tm.msm_ms; timings4[3] = tm.total_ms; }
  return rc;
}

int bh_groth16_prove_witness_part(bh_params *params, const bh_r1cs *r1cs, const void *input_assignment, size_t n_inputs,
                                  const void *aux_assignment, size_t n_aux, size_t part, size_t parts, void *sums_out,
                                  float *timings4) {
  using namespace groth16;
  if (!params || !r1cs || !sums_out || (n_inputs && !input_assignment) || (n_aux && !aux_assignment)) return BH_ERR_INVALID_ARG;
  ProveTimings tm = {0, 0, 0, 0};
  int rc = run_guarded_sums([&] {
    R1csView view(r1cs);
    std::vector<Fr> in(n_inputs), aux(n_aux);
    if (n_inputs) memcpy(in.data(), input_assignment, n_inputs * 32);
    if (n_aux) memcpy(aux.data(), aux_assignment, n_aux * 32);
    return prove_witness_part(view.r, *params->p, in.data(), n_inputs, aux.data(), n_aux, part, parts, &tm);
  }, sums_out);
  if (timings4) { timings4[0] = tm.synthesis_ms; timings4[1] = tm.h_poly_ms; timings4[2] = tm.msm_ms; timings4[3] = tm.total_ms; }
  return rc;
}

void bh_groth16_sums_add(void *acc, const void *other) {
  groth16::MsmSums a, b;
  memcpy(&a, acc, sizeof a);
  memcpy(&b, other, sizeof b);
  a.add(b);
  memcpy(acc, &a, sizeof a);
}

int bh_groth16_assemble(bh_params *params, const void *sums, const void *r, const void *s, void *proof_out) {
  using namespace groth16;
  if (!params || !sums || !r || !s || !proof_out) return BH_ERR_INVALID_ARG;
  MsmSums m;
  memcpy(&m, sums, sizeof m);
  Fr rr, ss;
  memcpy(&rr, r, 32); memcpy(&ss, s, 32);
  return run_guarded([&] { return assemble_proof(*params->p, m, rr, ss); }, proof_out);
}

}  // extern "C"

