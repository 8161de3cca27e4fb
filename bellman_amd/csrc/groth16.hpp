// C++ host mirror of the circuit-facing and prover-facing API of bellman for the proving hot path.
// Same names, argument meaning and error behaviour as the reference (the Rust toolchain is absent
// from this image, so the host side above the C ABI is C++ where the reference is compiled code):
//
//   bellman::{Variable, Index, LinearCombination, ConstraintSystem, Circuit, SynthesisError}
//                                         /root/reference/src/lib.rs:156-437
//   bellman::multiexp::DensityTracker     /root/reference/src/multiexp.rs:117-157
//   groth16::{Proof, VerifyingKey, Parameters (as ParameterSource), create_proof}
//                                         /root/reference/groth16/src/lib.rs:25-30,219-245,411-473
//                                         /root/reference/groth16/src/prover.rs:19-361
//
// Synthesis (user code + LC evaluation) runs on the host exactly as in the reference; every FFT
// and MSM goes through the C ABI of include/bellman_hip.h to the gfx950 kernels.
#pragma once
#include <stdint.h>
#include <string.h>

#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <thread>
#include <type_traits>
#include <stdexcept>
#include <utility>
#include <vector>

#include "../../include/bellman_hip.h"

namespace bellman {

// ---- scalar field element (bls12_381::Scalar): 4x64 Montgomery limbs, little-endian -------------
// The arithmetic a circuit and the linear-combination evaluation use per constraint (+, -, *, comparisons) is defined
// INLINE here: as calls into the library they were a third of the synthesis time of a 2^20-constraint circuit
// (profiles/r3_host_synthesis.txt).
namespace fr_detail {
typedef unsigned __int128 u128;
constexpr uint64_t MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
constexpr uint64_t INV = 0xfffffffeffffffffULL;
constexpr uint64_t R[4] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL};
constexpr uint64_t R2[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};
inline bool geq_mod(const uint64_t *a) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] > MOD[i]) return true;
    if (a[i] < MOD[i]) return false;
  }
  return true;
}
inline void sub_mod(uint64_t *a) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a[i] - MOD[i] - (uint64_t)br;
    a[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
// 4x64 CIOS Montgomery product, fully unrolled (synthesis is the serial part of create_proof)
__attribute__((always_inline)) inline void mont_mul(uint64_t *r, const uint64_t *a, const uint64_t *b) {
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#define BH_FR_ROW(bi)                                                                     \
  {                                                                                       \
    u128 c = (u128)a[0] * (bi) + t0; t0 = (uint64_t)c; c >>= 64;                          \
    c += (u128)a[1] * (bi) + t1; t1 = (uint64_t)c; c >>= 64;                              \
    c += (u128)a[2] * (bi) + t2; t2 = (uint64_t)c; c >>= 64;                              \
    c += (u128)a[3] * (bi) + t3; t3 = (uint64_t)c; c >>= 64;                              \
    c += t4; t4 = (uint64_t)c; const uint64_t t5 = (uint64_t)(c >> 64);                   \
    const uint64_t m = t0 * INV;                                                          \
    c = ((u128)m * MOD[0] + t0) >> 64;                                                    \
    c += (u128)m * MOD[1] + t1; t0 = (uint64_t)c; c >>= 64;                               \
    c += (u128)m * MOD[2] + t2; t1 = (uint64_t)c; c >>= 64;                               \
    c += (u128)m * MOD[3] + t3; t2 = (uint64_t)c; c >>= 64;                               \
    c += t4; t3 = (uint64_t)c; t4 = t5 + (uint64_t)(c >> 64);                             \
  }
  BH_FR_ROW(b[0]) BH_FR_ROW(b[1]) BH_FR_ROW(b[2]) BH_FR_ROW(b[3])
  uint64_t t[4] = {t0, t1, t2, t3};
  if (t4 || geq_mod(t)) sub_mod(t);
  r[0] = t[0]; r[1] = t[1]; r[2] = t[2]; r[3] = t[3];
}
// a * v for a ONE-limb second operand: the product rows of its three zero limbs vanish and only their reduction steps
// remain - 24 limb products instead of 36.  Fr::from_u64 (bls12_381's `From<u64> for Scalar` is a full product with R^2)
// is what a circuit calls for every constant it writes down; the synthetic chain circuit of BASELINE config C4 spends
// two of its three products per constraint there.
__attribute__((always_inline)) inline void mont_mul_u64(uint64_t *r, const uint64_t *a, uint64_t v) {
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
  BH_FR_ROW(v)
#define BH_FR_REDUCE_ROW                                                                  \
  {                                                                                       \
    const uint64_t m = t0 * INV;                                                          \
    u128 c = ((u128)m * MOD[0] + t0) >> 64;                                               \
    c += (u128)m * MOD[1] + t1; t0 = (uint64_t)c; c >>= 64;                               \
    c += (u128)m * MOD[2] + t2; t1 = (uint64_t)c; c >>= 64;                               \
    c += (u128)m * MOD[3] + t3; t2 = (uint64_t)c; c >>= 64;                               \
    c += t4; t3 = (uint64_t)c; t4 = (uint64_t)(c >> 64);                                  \
  }
  BH_FR_REDUCE_ROW BH_FR_REDUCE_ROW BH_FR_REDUCE_ROW
#undef BH_FR_REDUCE_ROW
  uint64_t t[4] = {t0, t1, t2, t3};
  if (t4 || geq_mod(t)) sub_mod(t);
  r[0] = t[0]; r[1] = t[1]; r[2] = t[2]; r[3] = t[3];
}
#undef BH_FR_ROW
}  // namespace fr_detail

struct Fr {
  uint64_t l[4];
  static Fr zero() { return Fr{{0, 0, 0, 0}}; }
  static Fr one() { return Fr{{fr_detail::R[0], fr_detail::R[1], fr_detail::R[2], fr_detail::R[3]}}; }
  static Fr from_u64(uint64_t v) {
    Fr r;
    fr_detail::mont_mul_u64(r.l, fr_detail::R2, v);
    return r;
  }
  static Fr from_u512(const uint64_t limbs_le[8]);   // wide reduction, as ff's Field::random does
  bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
  bool operator==(const Fr &o) const { return ((l[0] ^ o.l[0]) | (l[1] ^ o.l[1]) | (l[2] ^ o.l[2]) | (l[3] ^ o.l[3])) == 0; }
  bool operator!=(const Fr &o) const { return !(*this == o); }
  Fr operator+(const Fr &o) const {
    Fr r;
    fr_detail::u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (fr_detail::u128)l[i] + o.l[i];
      r.l[i] = (uint64_t)c;
      c >>= 64;
    }
    if (fr_detail::geq_mod(r.l)) fr_detail::sub_mod(r.l);
    return r;
  }
  Fr operator-(const Fr &o) const {
    Fr r;
    fr_detail::u128 br = 0;
    for (int i = 0; i < 4; i++) {
      fr_detail::u128 d = (fr_detail::u128)l[i] - o.l[i] - (uint64_t)br;
      r.l[i] = (uint64_t)d;
      br = (d >> 64) & 1;
    }
    if (br) {
      fr_detail::u128 c = 0;
      for (int i = 0; i < 4; i++) {
        c += (fr_detail::u128)r.l[i] + fr_detail::MOD[i];
        r.l[i] = (uint64_t)c;
        c >>= 64;
      }
    }
    return r;
  }
  Fr operator*(const Fr &o) const { Fr r; fr_detail::mont_mul(r.l, l, o.l); return r; }
  Fr neg() const { return Fr::zero() - *this; }
  void to_canonical(uint64_t out[4]) const {   // the bits of Exponent::Bits (multiexp.rs:179)
    const uint64_t one_[4] = {1, 0, 0, 0};
    fr_detail::mont_mul(out, l, one_);
  }
  Fr pow_vartime(uint64_t e) const;
  Fr invert() const;                          // *this must not be zero
};

// ---- src/lib.rs:303-319 -------------------------------------------------------------------------
struct SynthesisError : std::runtime_error {
  int code;   // the C-ABI code: 1 UnexpectedIdentity, 2 IoError(UnexpectedEof), 3 PolynomialDegreeTooLarge, 4 AssignmentMissing
  SynthesisError(int c, const char *what) : std::runtime_error(what), code(c) {}
};

// io::Error as raised by Parameters::read / VerifyingKey::read (groth16/src/lib.rs:159-215,289-398)
struct IoError : std::runtime_error {
  int code;   // BH_ERR_UNEXPECTED_EOF, BH_ERR_INVALID_POINT ("invalid G1"/"invalid G2"), BH_ERR_POINT_AT_INFINITY
  IoError(int c, const char *what) : std::runtime_error(what), code(c) {}
};

// ---- src/lib.rs:163-185 -------------------------------------------------------------------------
enum class Index { Input, Aux };
struct Variable {
  Index kind;
  size_t idx;
  static Variable new_unchecked(Index k, size_t i) { return Variable{k, i}; }
};

// ---- src/multiexp.rs:117-157 --------------------------------------------------------------------
class DensityTracker {
 public:
  void add_element() {
    if ((len_ & 63) == 0) words_.push_back(0);
    len_++;
  }
  void inc(size_t idx) {
    uint64_t &w = words_[idx >> 6];
    const uint64_t bit = uint64_t(1) << (idx & 63);
    if (!(w & bit)) { w |= bit; total_++; }
  }
  void clear() { words_.clear(); len_ = 0; total_ = 0; }   // keeps the capacity (recycled assignments)
  size_t get_total_density() const { return total_; }
  size_t get_query_size() const { return len_; }
  const uint64_t *words() const { return words_.data(); }

 private:
  std::vector<uint64_t> words_;   // LSB0, like BitVec<usize, Lsb0>
  size_t len_ = 0, total_ = 0;
};


// ---- src/lib.rs:190-300: ordered (variable, coeff) list, duplicates are NOT merged ---------------
// Terms live inline for the common short combinations (no heap traffic during synthesis, which is
// the serial part of create_proof); longer ones spill to a vector.
//
// EVALUATING combinations.  create_proof's ConstraintSystem (ProvingAssignment, prover.rs:105-145) never looks at the
// terms of a constraint's combinations again: it evaluates them and updates the query densities (prover.rs:19-55).  It
// therefore hands the closures a combination that is bound to an LcSink and does exactly that as each term is added -
// nothing is stored, the object that travels through `|lc| lc + a + b` is 56 bytes instead of 230, and there is no
// second pass.  Same values, same densities, same term order (profiles/r3_host_synthesis.txt).
struct LcSink {
  // input_assignment / aux_assignment: the VECTORS, read per term - a closure that allocates a variable while it builds
  // its combination (the reference's borrow rules forbid it, C++ does not) may reallocate them
  const std::vector<Fr> *inputs, *aux;
  DensityTracker *input_density, *aux_density;    // either may be null (prover.rs:119-141)
  // set instead of the four fields above: every term is handed to `hook` as it is added (a ConstraintSystem that
  // records the structure of the circuit - the R1CS capture - takes the terms straight into its matrices)
  void (*hook)(void *self, Variable v, const Fr &coeff) = nullptr;
  void *self = nullptr;
};
class LinearCombination {
 public:
  struct Term { Variable first; Fr second; };     // (variable, coefficient); an aggregate, so copies are plain memcpy
  static LinearCombination zero() { return LinearCombination(); }
  static LinearCombination evaluating(const LcSink *sink) {
    LinearCombination r;
    r.sink_ = sink;
    r.acc_ = Fr::zero();
    return r;
  }
  LinearCombination() : n_(0), sink_(nullptr) {}
  LinearCombination(const LinearCombination &o) : n_(o.n_), sink_(o.sink_) {
    if (sink_) acc_ = o.acc_; else { memcpy(inl_, o.inl_, sizeof inl_); more_ = o.more_; }
  }
  LinearCombination(LinearCombination &&o) noexcept : n_(o.n_), sink_(o.sink_) {
    if (sink_) acc_ = o.acc_; else { memcpy(inl_, o.inl_, sizeof inl_); more_ = std::move(o.more_); }
  }
  LinearCombination &operator=(const LinearCombination &o) {
    if (this == &o) return *this;
    n_ = o.n_; sink_ = o.sink_;
    if (sink_) acc_ = o.acc_; else { memcpy(inl_, o.inl_, sizeof inl_); more_ = o.more_; }
    return *this;
  }
  LinearCombination &operator=(LinearCombination &&o) noexcept {
    if (this == &o) return *this;
    n_ = o.n_; sink_ = o.sink_;
    if (sink_) acc_ = o.acc_; else { memcpy(inl_, o.inl_, sizeof inl_); more_ = std::move(o.more_); }
    return *this;
  }
  // lvalue operands are copied (value semantics); a temporary is extended in place and MOVED out - returned by value,
  // so that `lc = std::move(lc) + x` is not a self-move and `auto &&r = zero() + a` does not dangle (moving an
  // evaluating combination is a 56-byte copy)
  LinearCombination operator+(Variable v) const & { LinearCombination r(*this); r.push(v, Fr::one()); return r; }
  LinearCombination operator+(Variable v) && { push(v, Fr::one()); return std::move(*this); }
  LinearCombination operator-(Variable v) const & { LinearCombination r(*this); r.push(v, Fr::one().neg()); return r; }
  LinearCombination operator-(Variable v) && { push(v, Fr::one().neg()); return std::move(*this); }
  LinearCombination operator+(std::pair<Fr, Variable> t) const & { LinearCombination r(*this); r.push(t.second, t.first); return r; }
  LinearCombination operator+(std::pair<Fr, Variable> t) && { push(t.second, t.first); return std::move(*this); }
  LinearCombination operator-(std::pair<Fr, Variable> t) const & { LinearCombination r(*this); r.push(t.second, t.first.neg()); return r; }
  LinearCombination operator-(std::pair<Fr, Variable> t) && { push(t.second, t.first.neg()); return std::move(*this); }
  size_t size() const { return n_; }
  // stored combinations only
  const Term &operator[](size_t i) const { return i < INLINE ? inl_[i] : more_[i - INLINE]; }
  bool is_evaluating() const { return sink_ != nullptr; }
  const Fr &value() const { return acc_; }   // evaluating combinations only

 private:
  static constexpr size_t INLINE = 4;
  void push(Variable v, const Fr &c) {
    n_++;
    if (sink_) {   // prover.rs:19-55 for this one term
      if (c.is_zero()) return;            // zero coefficients count for neither value nor density (:31)
      if (sink_->hook) { sink_->hook(sink_->self, v, c); return; }
      const Fr *value;
      if (v.kind == Index::Input) {
        value = sink_->inputs->data() + v.idx;
        if (sink_->input_density) sink_->input_density->inc(v.idx);
      } else {
        value = sink_->aux->data() + v.idx;
        if (sink_->aux_density) sink_->aux_density->inc(v.idx);
      }
      const Fr one = Fr::one();
      // most terms carry the coefficient one (`lc + x`), then the ubiquitous `(c, CS::one())` terms: value one
      if (c == one) acc_ = acc_ + *value;
      else if (*value == one) acc_ = acc_ + c;
      else acc_ = acc_ + *value * c;
      return;
    }
    if (n_ <= INLINE) inl_[n_ - 1] = Term{v, c}; else more_.push_back(Term{v, c});
  }
  Term inl_[INLINE];
  std::vector<Term> more_;
  size_t n_;
  const LcSink *sink_;
  Fr acc_;
};

// Non-owning callable reference (two pointers, never allocates): the C++ stand-in for the
// monomorphised `FnOnce` parameters of ConstraintSystem::{alloc, enforce} (src/lib.rs:385-416).
// The referenced callable only has to live for the duration of the call it is passed to.
template <class Sig> class FunctionRef;
template <class R, class... Args>
class FunctionRef<R(Args...)> {
 public:
  template <class F>
  FunctionRef(F &&f) : obj_((void *)&f), call_([](void *o, Args... args) -> R {
    return (*reinterpret_cast<typename std::remove_reference<F>::type *>(o))(std::forward<Args>(args)...);
  }) {}
  R operator()(Args... args) const { return call_(obj_, std::forward<Args>(args)...); }

 private:
  void *obj_;
  R (*call_)(void *, Args...);
};

typedef FunctionRef<LinearCombination(LinearCombination)> LcFn;
typedef FunctionRef<Fr()> ValueFn;

// ---- src/lib.rs:374-437 -------------------------------------------------------------------------
class ConstraintSystem {
 public:
  virtual ~ConstraintSystem() {}
  static Variable one() { return Variable::new_unchecked(Index::Input, 0); }
  virtual Variable alloc(ValueFn f) = 0;
  virtual Variable alloc_input(ValueFn f) = 0;
  virtual void enforce(LcFn a, LcFn b, LcFn c) = 0;
};

// ---- src/lib.rs:156-159 -------------------------------------------------------------------------
class Circuit {
 public:
  virtual ~Circuit() {}
  virtual void synthesize(ConstraintSystem &cs) = 0;
};

}  // namespace bellman

namespace groth16 {
using bellman::Fr;

struct G1Affine { uint64_t v[12]; bool is_identity() const; };   // x | y, Montgomery; all-zero = identity
struct G2Affine { uint64_t v[24]; bool is_identity() const; };

struct Proof {                                                    // groth16/src/lib.rs:25-30
  G1Affine a; G2Affine b; G1Affine c;
  void write(unsigned char out[192]) const;                       // :38-46 compressed A | B | C
};

struct VerifyingKey {                                             // groth16/src/lib.rs:91-128
  G1Affine alpha_g1, beta_g1;
  G2Affine beta_g2;
  G1Affine delta_g1;
  G2Affine delta_g2;
  // verifier-side elements: not used by create_proof; filled by Parameters::read and the generator
  G2Affine gamma_g2 = G2Affine{};
  std::vector<G1Affine> ic;
};

class R1cs;

// `&Parameters` as ParameterSource (groth16/src/lib.rs:435-473): the five query vectors live in HBM.
class Parameters {
 public:
  Parameters(bh_ctx *ctx, const VerifyingKey &vk, const G1Affine *h, size_t nh, const G1Affine *l, size_t nl,
             const G1Affine *a, size_t na, const G1Affine *b_g1, size_t nb1, const G2Affine *b_g2, size_t nb2);
  // Parameters::read (groth16/src/lib.rs:289-398): the serialized CRS, decoded (and with
  // `checked` validated: on the curve, in the prime-order subgroup) on the device.  Throws IoError.
  Parameters(bh_ctx *ctx, const void *bytes, size_t len, bool checked);
  // generate_parameters (groth16/src/generator.rs:163-510) for the circuit whose matrices are `r1cs`:
  // powers of tau, ifft to the Lagrange basis, the QAP polynomials at tau and every fixed-base
  // multiplication run on the device.  Throws SynthesisError (UnexpectedIdentity for a zero gamma or
  // delta, UnconstrainedVariable, PolynomialDegreeTooLarge).
  Parameters(bh_ctx *ctx, R1cs &r1cs, const G1Affine &g1, const G2Affine &g2, const Fr &alpha, const Fr &beta,
             const Fr &gamma, const Fr &delta, const Fr &tau);
  // Parameters::write (groth16/src/lib.rs:258-287)
  std::vector<unsigned char> write() const;
  size_t serialized_size() const;                            // bytes `write` produces
  void write_into(unsigned char *dst, size_t cap) const;     // the same bytes into the caller's buffer (no intermediate copy)
  ~Parameters();
  Parameters(const Parameters &) = delete;
  bh_ctx *ctx;
  VerifyingKey vk;
  bh_bases *h = nullptr, *l = nullptr, *a = nullptr, *b_g1 = nullptr, *b_g2 = nullptr;
};

// prover.rs:57-162: the ConstraintSystem that records evaluations, assignments and query densities
class ProvingAssignment : public bellman::ConstraintSystem {
 public:
  bellman::DensityTracker a_aux_density, b_input_density, b_aux_density;
  std::vector<Fr> a, b, c;
  std::vector<Fr> input_assignment, aux_assignment;
  bellman::Variable alloc(bellman::ValueFn f) override;
  bellman::Variable alloc_input(bellman::ValueFn f) override;
  void enforce(bellman::LcFn a, bellman::LcFn b, bellman::LcFn c) override;
};

struct ProveTimings { float synthesis_ms, h_poly_ms, msm_ms, total_ms; };

// The eight multiexp results of create_proof in the order the reference waits for them
// (prover.rs:339-354).  With the proof spread over several GPUs every rank produces the sums over its
// slice of the scalars; the per-slot group sums over all ranks are what assemble_proof needs.
struct MsmSums {
  G1Affine a_in, a_aux, b1_in, b1_aux;
  G2Affine b2_in, b2_aux;
  G1Affine h, l;
  void add(const MsmSums &other);   // slot-wise group addition
};
static_assert(sizeof(MsmSums) == 6 * 96 + 2 * 192, "packed: the C ABI hands it over as 960 bytes");

// The circuit's three constraint matrices, captured once and kept in HBM.  Capture
// runs `synthesize` against a structure-only ConstraintSystem that never calls the value closures -
// the same thing generator.rs:43-131 (KeypairAssembly) does to build the CRS for this circuit.
class R1cs {
 public:
  R1cs(bellman::Circuit &shape_of, bh_ctx *ctx);
  explicit R1cs(bh_r1cs *existing);   // adopt a handle made through the C ABI
  ~R1cs();
  R1cs(const R1cs &) = delete;
  bh_r1cs *handle = nullptr;
  size_t num_inputs = 0, num_aux = 0, num_constraints = 0;
};

// prover.rs:57-162 reduced to witness generation: alloc/alloc_input run the value closures,
// enforce does nothing (the evaluations come from the device-resident matrices).
class WitnessAssignment : public bellman::ConstraintSystem {
 public:
  std::vector<Fr> input_assignment, aux_assignment;
  bellman::Variable alloc(bellman::ValueFn f) override;
  bellman::Variable alloc_input(bellman::ValueFn f) override;
  void enforce(bellman::LcFn, bellman::LcFn, bellman::LcFn) override {}
};

// prover.rs:182-361.  Throws bellman::SynthesisError.
Proof create_proof(bellman::Circuit &circuit, Parameters &params, const Fr &r, const Fr &s,
                   ProveTimings *timings = nullptr);
// prover.rs:164-180: r and s drawn from `rng` (any callable returning uint64_t; 512 bits each, reduced
// mod q like ff's Field::random), then create_proof.
template <class Rng>
Proof create_random_proof(bellman::Circuit &circuit, Parameters &params, Rng &&rng) {
  uint64_t w[8];
  for (uint64_t &x : w) x = rng();
  const Fr r = Fr::from_u512(w);
  for (uint64_t &x : w) x = rng();
  const Fr s = Fr::from_u512(w);
  return create_proof(circuit, params, r, s);
}
// prover.rs:217-360 on an already synthesised assignment (input constraints already appended)
Proof prove_assignment(ProvingAssignment &prover, Parameters &params, const Fr &r, const Fr &s,
                       ProveTimings *timings = nullptr);
// the same on plain views of the assignment's fields (what the C entry point receives: nothing is copied on the host)
struct AssignmentView {
  const Fr *a, *b, *c; size_t n_constraints;
  const Fr *input_assignment; size_t n_inputs;
  const Fr *aux_assignment; size_t n_aux;
  const uint64_t *a_aux_density, *b_input_density, *b_aux_density;   // LSB0 words (DensityTracker::words)
};
Proof prove_assignment(const AssignmentView &v, Parameters &params, const Fr &r, const Fr &s,
                       ProveTimings *timings = nullptr);
// The same two entry points with the constraint evaluation on the device: only the witness
// closures of `circuit` run on the host.  The circuit must have the shape `r1cs` was captured from
// (it does whenever `params` belongs to it); a different variable count throws std::invalid_argument.
Proof create_proof(bellman::Circuit &circuit, const R1cs &r1cs, Parameters &params, const Fr &r, const Fr &s,
                   ProveTimings *timings = nullptr);
// One proof over `parts` GPUs: every rank holds the CRS and the matrices, generates
// the witness and runs the h block (replicated, small), but computes each multiexp only over part
// `part` of the scalar indices; the ranks exchange their MsmSums (960 B), add them slot-wise and
// assemble the identical proof.
MsmSums prove_witness_part(const R1cs &r1cs, Parameters &params, const Fr *input_assignment, size_t n_inputs,
                           const Fr *aux_assignment, size_t n_aux, size_t part, size_t parts,
                           ProveTimings *timings = nullptr);
Proof assemble_proof(const Parameters &params, const MsmSums &sums, const Fr &r, const Fr &s);
Proof prove_witness(const R1cs &r1cs, Parameters &params, const Fr *input_assignment, size_t n_inputs,
                    const Fr *aux_assignment, size_t n_aux, const Fr &r, const Fr &s,
                    ProveTimings *timings = nullptr);

// ---- one caller, proofs back to back ---------------------------------------------------------------------------
// create_proof is synthesis on the host followed by the device part (prover.rs:182-215, then :217-360).  A single
// caller that proves in a loop leaves the GPU idle during every synthesis and the host idle during every device part.
// AsyncProof runs the device part of ONE proof on a helper thread; ProofPipeline keeps up to `depth` of them in flight
// while the calling thread synthesises the next circuit - proofs come back in submission order, each identical to what
// create_proof returns for the same circuit, r and s.
struct AsyncProof {
  std::thread worker;
  Proof proof;
  ProveTimings timings = {0, 0, 0, 0};
  std::exception_ptr error;
  std::unique_ptr<WitnessAssignment> witness;     // device-evaluated constraints (r1cs != null)
  std::unique_ptr<ProvingAssignment> assignment;  // host-evaluated constraints, as in the reference
  ~AsyncProof();   // joins the helper thread; the assignment buffers go back to the recycling pool
  Proof wait(ProveTimings *tm = nullptr);          // joins; rethrows what the device part threw
};
// synthesises `circuit` on the calling thread (into a WitnessAssignment when `r1cs` is given, else into a
// ProvingAssignment with the input constraints of prover.rs:208-215) and starts the device part
std::unique_ptr<AsyncProof> create_proof_async(bellman::Circuit &circuit, const R1cs *r1cs, Parameters &params, const Fr &r,
                                               const Fr &s);
// [r4] the same split for a host that synthesised by itself (a C or Rust caller holding bellman's ProvingAssignment, or
// just the witness when the constraint matrices are resident): the device part of prove_assignment / prove_witness on
// a helper thread.  prove_assignment_async READS THE CALLER'S ARRAYS IN PLACE until wait() returns (five 32 MiB vectors
// at 2^20 constraints: not copied); prove_witness_async copies the two witness vectors (they may be unaligned) before
// it returns, so the caller may reuse its buffers for the next witness at once.
std::unique_ptr<AsyncProof> prove_assignment_async(const AssignmentView &v, Parameters &params, const Fr &r, const Fr &s);
std::unique_ptr<AsyncProof> prove_witness_async(const R1cs &r1cs, Parameters &params, const void *input_assignment, size_t n_inputs,
                                                const void *aux_assignment, size_t n_aux, const Fr &r, const Fr &s);
class ProofPipeline {
 public:
  ProofPipeline(Parameters &params, const R1cs *r1cs, size_t depth = 2) : params_(params), r1cs_(r1cs), depth_(depth ? depth : 1) {}
  // blocks (by finishing the oldest proof into an internal queue) while `depth` proofs are in flight
  void submit(bellman::Circuit &circuit, const Fr &r, const Fr &s);
  Proof next(ProveTimings *tm = nullptr);   // the oldest submitted proof
  size_t pending() const { return inflight_.size() + done_.size(); }

 private:
  Parameters &params_;
  const R1cs *r1cs_;
  size_t depth_;
  std::deque<std::unique_ptr<AsyncProof>> inflight_;
  struct Done { Proof proof; ProveTimings tm; std::exception_ptr error; };
  std::deque<Done> done_;
  void retire_oldest();
};

}  // namespace groth16
