// The demo circuits (written against the mirror like bellman user code) and the extern "C" entry points of
// the groth16 layer (declared in include/bellman_hip.h).
#include <string.h>

#include <memory>

#include <chrono>
#include <functional>
#include <vector>

#include "../../include/bellman_hip_test.h"
#include "groth16_internal.hpp"

namespace groth16 {
using namespace bellman;
using namespace detail;

// ---------------------------------------------------------------------------------------------------
// demo circuits, written against the mirror exactly like bellman user code
// ---------------------------------------------------------------------------------------------------
// MiMCDemo: /root/reference/groth16/tests/common/mod.rs:37-129 (LongsightF322p3)
class MiMCDemo : public Circuit {
 public:
  Fr xl, xr;
  const Fr *constants;
  size_t rounds;
  void synthesize(ConstraintSystem &cs) override {
    Fr xl_value = xl, xr_value = xr;
    Variable xlv = cs.alloc([&] { return xl_value; });
    Variable xrv = cs.alloc([&] { return xr_value; });
    for (size_t i = 0; i < rounds; i++) {
      const Fr ci = constants[i];
      const Fr t0 = xl_value + ci;
      const Fr tmp_value = t0 * t0;
      Variable tmp = cs.alloc([&] { return tmp_value; });
      cs.enforce([&](LinearCombination lc) { return lc + xlv + std::make_pair(ci, ConstraintSystem::one()); },
                 [&](LinearCombination lc) { return lc + xlv + std::make_pair(ci, ConstraintSystem::one()); },
                 [&](LinearCombination lc) { return lc + tmp; });
      const Fr new_xl_value = t0 * tmp_value + xr_value;
      Variable new_xl = (i == rounds - 1) ? cs.alloc_input([&] { return new_xl_value; })
                                          : cs.alloc([&] { return new_xl_value; });
      cs.enforce([&](LinearCombination lc) { return lc + tmp; },
                 [&](LinearCombination lc) { return lc + xlv + std::make_pair(ci, ConstraintSystem::one()); },
                 [&](LinearCombination lc) { return lc + new_xl - xrv; });
      xrv = xlv; xr_value = xl_value;
      xlv = new_xl; xl_value = new_xl_value;
    }
  }
};

// Synthetic multiplicative chain: M rounds
//   even i: (x_i + k_i) * (x_i + k'_i) = x_{i+1}       (x_i in the A and B queries)
//   odd  i: (x_i + k_i + 0*x_0) * (k'_i)  = x_{i+1}    (x_i only in A; a zero-coefficient term, prover.rs:31)
// and finally x_M * 1 = out (public input).  Constants from SplitMix64(seed).
class ChainCircuit : public Circuit {
 public:
  uint64_t seed;
  size_t rounds;
  Fr x0;
  static uint64_t splitmix(uint64_t &st) {
    uint64_t z = (st += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
  }
  void synthesize(ConstraintSystem &cs) override {
    uint64_t st = seed;
    Fr x_value = x0;
    Variable x = cs.alloc([&] { return x_value; });
    const Variable first = x;
    for (size_t i = 0; i < rounds; i++) {
      const Fr k = Fr::from_u64(splitmix(st)), k2 = Fr::from_u64(splitmix(st) | 1);
      const Fr lhs = x_value + k;
      const Fr rhs = (i & 1) ? k2 : (x_value + k2);
      const Fr next_value = lhs * rhs;
      Variable next = cs.alloc([&] { return next_value; });
      if (i & 1) {
        cs.enforce([&](LinearCombination lc) { return lc + x + std::make_pair(k, ConstraintSystem::one()) + std::make_pair(Fr::zero(), first); },
                   [&](LinearCombination lc) { return lc + std::make_pair(k2, ConstraintSystem::one()); },
                   [&](LinearCombination lc) { return lc + next; });
      } else {
        cs.enforce([&](LinearCombination lc) { return lc + x + std::make_pair(k, ConstraintSystem::one()); },
                   [&](LinearCombination lc) { return lc + x + std::make_pair(k2, ConstraintSystem::one()); },
                   [&](LinearCombination lc) { return lc + next; });
      }
      x = next;
      x_value = next_value;
    }
    Variable out = cs.alloc_input([&] { return x_value; });
    cs.enforce([&](LinearCombination lc) { return lc + x; }, [&](LinearCombination lc) { return lc + ConstraintSystem::one(); },
               [&](LinearCombination lc) { return lc + out; });
  }
};

}  // namespace groth16

// ---------------------------------------------------------------------------------------------------
// C entry points (declared in include/bellman_hip.h)
// ---------------------------------------------------------------------------------------------------
struct bh_params {
  groth16::Parameters *p;
};

static int run_guarded_sums(const std::function<groth16::MsmSums()> &f, void *sums_out) {
  try {
    groth16::MsmSums m = f();
    memcpy(sums_out, &m, sizeof m);
    return BH_OK;
  } catch (const bellman::SynthesisError &e) { return e.code;
  } catch (const std::invalid_argument &) { return BH_ERR_INVALID_ARG;
  } catch (...) { return BH_ERR_HIP; }
}

// a non-owning groth16::R1cs over a handle that belongs to the C caller
struct R1csView {
  groth16::R1cs r;
  explicit R1csView(const bh_r1cs *h) : r(const_cast<bh_r1cs *>(h)) {}
  ~R1csView() { r.handle = nullptr; }
};

template <class F>
static int with_demo_circuit(int circuit_kind, size_t size, uint64_t seed, const void *witness, const void *constants, F &&f) {
  using namespace groth16;
  if (circuit_kind == 0) {   // MiMC: witness = xl | xr, constants = `size` round constants (Montgomery Fr)
    MiMCDemo c;
    c.xl = Fr::zero(); c.xr = Fr::zero();
    if (witness) { memcpy(&c.xl, witness, 32); memcpy(&c.xr, (const char *)witness + 32, 32); }
    c.constants = (const Fr *)constants;
    c.rounds = size;
    return f(c);
  }
  if (circuit_kind == 1) {   // chain: witness = x0, `size` rounds
    ChainCircuit c;
    c.seed = seed; c.rounds = size;
    c.x0 = Fr::zero();
    if (witness) memcpy(&c.x0, witness, 32);
    return f(c);
  }
  return BH_ERR_INVALID_ARG;
}

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
    size_t need = 864 + 4 + P.vk.ic.size() * 96 + 5 * 4;
    const bh_bases *qs[5] = {P.h, P.l, P.a, P.b_g1, P.b_g2};
    for (int q = 0; q < 5; q++) need += bh_bases_len(qs[q]) * (q < 4 ? 96 : 192);
    *len = need;
    if (!buf || cap < need) return buf ? BH_ERR_INVALID_ARG : BH_OK;   // size query when buf == NULL
    std::vector<unsigned char> o = P.write();
    memcpy(buf, o.data(), o.size());
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
void bh_test_proof_slice(size_t n, size_t part, size_t parts, size_t *lo, size_t *hi) {
  groth16::proof_slice_for_tests(n, part, parts, lo, hi);
}
double bh_test_synthesis_ms(int circuit_kind, size_t size, uint64_t seed, int mode) {
  // host-only timing of circuit synthesis (no device involved): mode 0 = ProvingAssignment (the
  // reference's structure: every linear combination evaluated on the host), 1 = WitnessAssignment
  using namespace groth16;
  std::vector<Fr> constants(circuit_kind == 0 ? size : 0, Fr::from_u64(7));
  Fr wit[2] = {Fr::from_u64(123456789), Fr::from_u64(987654321)};
  double ms = -1.0;
  with_demo_circuit(circuit_kind, size, seed, wit, constants.data(), [&](bellman::Circuit &c) -> int {
    // modes 2 / 3: the same into a RECYCLED assignment (cleared, capacity kept), as create_proof does from the second
    // proof on
    static ProvingAssignment kept_pa;
    static WitnessAssignment kept_w;
    if (mode == 2) {
      kept_pa.a.clear(); kept_pa.b.clear(); kept_pa.c.clear(); kept_pa.input_assignment.clear(); kept_pa.aux_assignment.clear();
      kept_pa.a_aux_density.clear(); kept_pa.b_input_density.clear(); kept_pa.b_aux_density.clear();
    }
    if (mode == 3) { kept_w.input_assignment.clear(); kept_w.aux_assignment.clear(); }
    const auto t0 = std::chrono::steady_clock::now();
    if (mode == 2) {
      kept_pa.alloc_input([] { return Fr::one(); });
      c.synthesize(kept_pa);
    } else if (mode == 3) {
      kept_w.alloc_input([] { return Fr::one(); });
      c.synthesize(kept_w);
    } else if (mode == 0) {
      ProvingAssignment pa;
      pa.alloc_input([] { return Fr::one(); });
      c.synthesize(pa);
    } else {
      WitnessAssignment w;
      w.alloc_input([] { return Fr::one(); });
      c.synthesize(w);
    }
    ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
  });
  return ms;
}
double bh_test_capture_check(int circuit_kind, size_t size, uint64_t seed, size_t out4[4]) {
  // host only: the structure capture (R1cs's constructor without the upload) of a demo circuit, checked against the
  // ProvingAssignment of the same circuit - captured A, B, C times the assignment == its a, b, c rows
  using namespace groth16;
  if (!out4) return -1.0;
  std::vector<Fr> constants(circuit_kind == 0 ? size : 0);
  for (size_t i = 0; i < constants.size(); i++) constants[i] = Fr::from_u64(0x9E3779B97F4A7C15ULL * (i % 7 + 1));   // repeats: the table must share them
  Fr wit[2] = {Fr::from_u64(123456789), Fr::from_u64(987654321)};
  double ms = -1.0;
  with_demo_circuit(circuit_kind, size, seed, wit, constants.data(), [&](bellman::Circuit &shape) -> int {
    return with_demo_circuit(circuit_kind, size, seed, wit, constants.data(), [&](bellman::Circuit &proved) -> int {
      ms = capture_check_for_tests(shape, proved, out4);
      return 0;
    });
  });
  return ms;
}
int bh_test_demo_assignment(int circuit_kind, size_t size, uint64_t seed, const void *witness, const void *constants,
                            size_t counts3[3], void *a, void *b, void *c, void *inputs, void *aux, uint64_t *a_aux_density,
                            uint64_t *b_input_density, uint64_t *b_aux_density) {
  // host only: synthesises the demo circuit into a ProvingAssignment exactly as create_proof does (prover.rs:182-215,
  // input constraints appended) and copies its fields out.  First call with null outputs for the counts.
  using namespace groth16;
  if (!counts3) return BH_ERR_INVALID_ARG;
  try {
    return with_demo_circuit(circuit_kind, size, seed, witness, constants, [&](bellman::Circuit &circ) -> int {
      ProvingAssignment pa;
      pa.alloc_input([] { return Fr::one(); });
      circ.synthesize(pa);
      for (size_t i = 0; i < pa.input_assignment.size(); i++)
        pa.enforce([i](bellman::LinearCombination lc) { return lc + bellman::Variable::new_unchecked(bellman::Index::Input, i); },
                   [](bellman::LinearCombination lc) { return lc; }, [](bellman::LinearCombination lc) { return lc; });
      counts3[0] = pa.a.size(); counts3[1] = pa.input_assignment.size(); counts3[2] = pa.aux_assignment.size();
      if (!a) return BH_OK;
      memcpy(a, pa.a.data(), pa.a.size() * 32); memcpy(b, pa.b.data(), pa.b.size() * 32); memcpy(c, pa.c.data(), pa.c.size() * 32);
      memcpy(inputs, pa.input_assignment.data(), pa.input_assignment.size() * 32);
      memcpy(aux, pa.aux_assignment.data(), pa.aux_assignment.size() * 32);
      memcpy(a_aux_density, pa.a_aux_density.words(), (pa.aux_assignment.size() + 63) / 64 * 8);
      memcpy(b_input_density, pa.b_input_density.words(), (pa.input_assignment.size() + 63) / 64 * 8);
      memcpy(b_aux_density, pa.b_aux_density.words(), (pa.aux_assignment.size() + 63) / 64 * 8);
      return BH_OK;
    });
  } catch (...) { return BH_ERR_HIP; }
}
void bh_test_fr_from_u512_host(void *r, const void *limbs8) {
  uint64_t w[8];
  memcpy(w, limbs8, 64);
  const bellman::Fr f = bellman::Fr::from_u512(w);
  memcpy(r, &f, 32);
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

static int run_guarded(const std::function<groth16::Proof()> &f, void *proof_out) {
  try {
    groth16::Proof p = f();
    memcpy(proof_out, &p.a, 96);
    memcpy((char *)proof_out + 96, &p.b, 192);
    memcpy((char *)proof_out + 288, &p.c, 96);
    return BH_OK;
  } catch (const bellman::SynthesisError &e) { return e.code;
  } catch (const std::invalid_argument &) { return BH_ERR_INVALID_ARG;
  } catch (...) { return BH_ERR_HIP; }
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

int bh_groth16_demo_r1cs(bh_ctx *ctx, int circuit_kind, size_t size, uint64_t seed, const void *constants, bh_r1cs **out) {
  if (!ctx || !out) return BH_ERR_INVALID_ARG;
  return with_demo_circuit(circuit_kind, size, seed, nullptr, constants, [&](bellman::Circuit &c) -> int {
    try {
      groth16::R1cs r(c, ctx);
      *out = r.handle;
      r.handle = nullptr;   // ownership moves to the caller (bh_r1cs_release)
      return BH_OK;
    } catch (const bellman::SynthesisError &e) { return e.code;
    } catch (...) { return BH_ERR_HIP; }
  });
}

int bh_groth16_prove_demo_r1cs(bh_params *params, const bh_r1cs *r1cs, int circuit_kind, size_t size, uint64_t seed,
                               const void *witness, const void *constants, const void *r, const void *s, void *proof_out,
                               float *timings4) {
  using namespace groth16;
  if (!params || !r1cs || !r || !s || !proof_out) return BH_ERR_INVALID_ARG;
  Fr rr, ss;
  memcpy(&rr, r, 32); memcpy(&ss, s, 32);
  ProveTimings tm = {0, 0, 0, 0};
  int rc = with_demo_circuit(circuit_kind, size, seed, witness, constants, [&](bellman::Circuit &c) -> int {
    return run_guarded([&] { R1csView view(r1cs); return create_proof(c, view.r, *params->p, rr, ss, &tm); }, proof_out);
  });
  if (timings4) { timings4[0] = tm.synthesis_ms; timings4[1] = tm.h_poly_ms; timings4[2] = tm.msm_ms; timings4[3] = tm.total_ms; }
  return rc;
}

// ---- asynchronous proofs: synthesis on the calling thread, the device part on a helper thread ---------------------
struct bh_proof_job {
  std::unique_ptr<groth16::AsyncProof> job;
  std::unique_ptr<R1csView> view;
  int early_rc = BH_OK;
};
int bh_groth16_prove_demo_async(bh_params *params, const bh_r1cs *r1cs, int circuit_kind, size_t size, uint64_t seed,
                                const void *witness, const void *constants, const void *r, const void *s, bh_proof_job **out) {
  using namespace groth16;
  if (!params || !r || !s || !out) return BH_ERR_INVALID_ARG;
  Fr rr, ss;
  memcpy(&rr, r, 32); memcpy(&ss, s, 32);
  std::unique_ptr<bh_proof_job> pj(new bh_proof_job());
  if (r1cs) pj->view.reset(new R1csView(r1cs));
  int rc = with_demo_circuit(circuit_kind, size, seed, witness, constants, [&](bellman::Circuit &c) -> int {
    try {
      pj->job = create_proof_async(c, r1cs ? &pj->view->r : nullptr, *params->p, rr, ss);
      return BH_OK;
    } catch (const bellman::SynthesisError &e) { return e.code;
    } catch (const std::invalid_argument &) { return BH_ERR_INVALID_ARG;
    } catch (...) { return BH_ERR_HIP; }
  });
  if (rc != BH_OK) return rc;
  *out = pj.release();
  return BH_OK;
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

int bh_groth16_prove_demo_r1cs_part(bh_params *params, const bh_r1cs *r1cs, int circuit_kind, size_t size, uint64_t seed,
                                    const void *witness, const void *constants, size_t part, size_t parts, void *sums_out,
                                    float *timings4) {
  using namespace groth16;
  if (!params || !r1cs || !sums_out) return BH_ERR_INVALID_ARG;
  ProveTimings tm = {0, 0, 0, 0};
  int rc = with_demo_circuit(circuit_kind, size, seed, witness, constants, [&](bellman::Circuit &c) -> int {
    return run_guarded_sums([&] {
      R1csView view(r1cs);
      const double t0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
      WitnessAssignment w;
      w.input_assignment.reserve(view.r.num_inputs);
      w.aux_assignment.reserve(view.r.num_aux);
      w.alloc_input([] { return Fr::one(); });
      c.synthesize(w);
      const double t1 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
      MsmSums m = prove_witness_part(view.r, *params->p, w.input_assignment.data(), w.input_assignment.size(),
                                     w.aux_assignment.data(), w.aux_assignment.size(), part, parts, &tm);
      tm.synthesis_ms = (float)(t1 - t0);
      tm.total_ms += tm.synthesis_ms;
      return m;
    }, sums_out);
  });
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

int bh_groth16_prove_demo(bh_params *params, int circuit_kind, size_t size, uint64_t seed, const void *witness,
                          const void *constants, const void *r, const void *s, void *proof_out, float *timings4) {
  using namespace groth16;
  if (!params || !witness || !r || !s || !proof_out) return BH_ERR_INVALID_ARG;
  Fr rr, ss;
  memcpy(&rr, r, 32); memcpy(&ss, s, 32);
  ProveTimings tm = {0, 0, 0, 0};
  int rc;
  if (circuit_kind == 0) {   // MiMC: witness = xl | xr, constants = `size` round constants (Montgomery Fr)
    MiMCDemo c;
    memcpy(&c.xl, witness, 32); memcpy(&c.xr, (const char *)witness + 32, 32);
    c.constants = (const Fr *)constants;
    c.rounds = size;
    rc = run_guarded([&] { return create_proof(c, *params->p, rr, ss, &tm); }, proof_out);
  } else if (circuit_kind == 1) {   // chain: witness = x0, `size` rounds
    ChainCircuit c;
    c.seed = seed; c.rounds = size;
    memcpy(&c.x0, witness, 32);
    rc = run_guarded([&] { return create_proof(c, *params->p, rr, ss, &tm); }, proof_out);
  } else {
    return BH_ERR_INVALID_ARG;
  }
  if (timings4) { timings4[0] = tm.synthesis_ms; timings4[1] = tm.h_poly_ms; timings4[2] = tm.msm_ms; timings4[3] = tm.total_ms; }
  return rc;
}

}  // extern "C"

