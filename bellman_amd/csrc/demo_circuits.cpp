// libbellman_hip_test.so, groth16 part: the built-in demo circuits (MiMC, the multiplicative chain of BASELINE config C4)
// written against the mirror exactly like bellman user code, their C entry points (Python cannot define a C++ circuit:
// bench.py and the tests reach create_proof on these circuits through ctypes) and the host-only test hooks of
// include/bellman_hip_test.h.  Links against libbellman_hip.so (the C++ mirror of groth16.hpp) and is not part of the
// product.
#include <string.h>

#include <chrono>
#include <functional>
#include <memory>
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

// A boolean-heavy circuit in the shape of the reference's bit-level gadgets (src/gadgets/boolean.rs: AllocatedBit::alloc
// enforces (1 - b) * b = 0, `and` enforces a * b = c, `xor` enforces (a + a) * b = a + b - c; src/gadgets/sha256.rs:307-331
// builds its rounds from exactly these): 64 state bits taken from the witness, then `rounds` steps
//   i % 3 == 0:  u = s[a] AND s[b]                 (kept aside)
//   i % 3 == 1:  s[c] = s[c] XOR u
//   i % 3 == 2:  s[a] = s[a] XOR s[b]
// with a, b, c walking over the state, and every 64th step the state packed into ONE field element
// (sum_j 2^j s[j]) * 1 = num (as the gadgets' `pack_into_inputs` / UInt32 arithmetic do); the last packed value is the
// public input.  More than 98 % of the aux assignment is 0 or 1 - the vectors Exponent::Zero / One exist for
// (src/multiexp.rs:172-182,245-252).  tests/circuits.py::boolmix_circuit is the same circuit.
class BoolMixCircuit : public Circuit {
 public:
  uint64_t seed;
  size_t rounds;
  Fr x0;
  void synthesize(ConstraintSystem &cs) override {
    uint64_t canon[4];
    x0.to_canonical(canon);
    uint64_t st = seed;
    const uint64_t init = canon[0] ^ ChainCircuit::splitmix(st);
    const Variable one = ConstraintSystem::one();
    Variable s[64];
    bool v[64];
    for (int j = 0; j < 64; j++) {
      v[j] = (init >> j) & 1;
      const Fr bv = v[j] ? Fr::one() : Fr::zero();
      s[j] = cs.alloc([&] { return bv; });
      const Variable b = s[j];
      cs.enforce([&](LinearCombination lc) { return lc + one - b; }, [&](LinearCombination lc) { return lc + b; },
                 [&](LinearCombination lc) { return lc; });
    }
    Fr pow2[64];
    pow2[0] = Fr::one();
    for (int j = 1; j < 64; j++) pow2[j] = pow2[j - 1] + pow2[j - 1];
    auto pack_value = [&] { uint64_t w = 0; for (int j = 0; j < 64; j++) w |= (uint64_t)v[j] << j; return Fr::from_u64(w); };
    auto pack_lc = [&](LinearCombination lc) { for (int j = 0; j < 64; j++) lc = std::move(lc) + std::make_pair(pow2[j], s[j]); return lc; };
    auto xor_into = [&](int dst, Variable other, bool other_v) {
      const bool tv = v[dst] ^ other_v;
      const Fr tf = tv ? Fr::one() : Fr::zero();
      const Variable x = s[dst];
      Variable t = cs.alloc([&] { return tf; });
      cs.enforce([&](LinearCombination lc) { return lc + x + x; }, [&](LinearCombination lc) { return lc + other; },
                 [&](LinearCombination lc) { return lc + x + other - t; });
      s[dst] = t; v[dst] = tv;
    };
    Variable pending = s[0];
    bool pending_v = v[0];
    for (size_t i = 0; i < rounds; i++) {
      const int a = (int)((7 * i + 1) % 64), c = (int)((29 * i + 11) % 64);
      int b = (int)((13 * i + 5) % 64);
      if (a == b) b = (b + 1) % 64;
      switch (i % 3) {
        case 0: {
          const bool uv = v[a] && v[b];
          const Fr uf = uv ? Fr::one() : Fr::zero();
          const Variable xa = s[a], xb = s[b];
          Variable u = cs.alloc([&] { return uf; });
          cs.enforce([&](LinearCombination lc) { return lc + xa; }, [&](LinearCombination lc) { return lc + xb; },
                     [&](LinearCombination lc) { return lc + u; });
          pending = u; pending_v = uv;
          break;
        }
        case 1: xor_into(c, pending, pending_v); break;
        default: xor_into(a, s[b], v[b]); break;
      }
      if (i % 64 == 63) {
        const Fr nv = pack_value();
        Variable num = cs.alloc([&] { return nv; });
        cs.enforce(pack_lc, [&](LinearCombination lc) { return lc + one; }, [&](LinearCombination lc) { return lc + num; });
      }
    }
    const Fr outv = pack_value();
    Variable out = cs.alloc_input([&] { return outv; });
    cs.enforce(pack_lc, [&](LinearCombination lc) { return lc + one; }, [&](LinearCombination lc) { return lc + out; });
  }
};

// Every way a linear combination can reach `enforce` through the mirror (test fixture: tests/circuits.py::forms_circuit is
// the same circuit against the oracle's ConstraintSystem).  Per round i, on variables v (aux), w (aux), p (a public input
// every third round) and constants k, k2 from SplitMix64(seed):
//   A: evaluating terms that need a product ((k, v): coefficient and value both != 1), `-`, a zero coefficient, a repeat
//   B: a STORED combination built from LinearCombination::zero(), ignoring the closure's argument - five or six terms, so
//      that it spills out of the inline storage - with both kinds of variable
//   C: the argument returned untouched by every third round (the empty combination), else `lc + out`
// The constraints are not meant to be satisfiable: ProvingAssignment evaluates whatever it is given.
class FormsCircuit : public Circuit {
 public:
  uint64_t seed;
  size_t rounds;
  Fr x0;
  void synthesize(ConstraintSystem &cs) override {
    uint64_t st = seed;
    Fr v_value = x0, w_value = x0 * x0 + Fr::from_u64(3);
    Variable v = cs.alloc([&] { return v_value; });
    Variable w = cs.alloc([&] { return w_value; });
    for (size_t i = 0; i < rounds; i++) {
      const Fr k = Fr::from_u64(ChainCircuit::splitmix(st)), k2 = Fr::from_u64(ChainCircuit::splitmix(st) | 1);
      const Fr out_value = (v_value * k - w_value) * k2;
      Variable out = (i % 3 == 2) ? cs.alloc_input([&] { return out_value; }) : cs.alloc([&] { return out_value; });
      const Variable one = ConstraintSystem::one();
      cs.enforce(
          [&](LinearCombination lc) { return lc + std::make_pair(k, v) - w - std::make_pair(k2, one) + std::make_pair(Fr::zero(), out) + v; },
          [&](LinearCombination) {
            LinearCombination s = LinearCombination::zero() + v + std::make_pair(k2, w) - std::make_pair(k, one) + one - out;
            if (i & 1) s = s + std::make_pair(k, out);
            return s;
          },
          [&](LinearCombination lc) { return (i % 3 == 0) ? lc : lc + out; });
      v = w; v_value = w_value;
      w = out; w_value = out_value;
    }
  }
};

// A circuit whose STRUCTURE is drawn from SplitMix64(seed) - how many variables a round allocates and of which kind, and for
// each of a constraint's three combinations: empty, a chain of 1..9 terms on the closure's argument, or a stored
// combination of 0..9 terms built from zero(); each term +v, -v, +(k, v) or -(k, v) with k from {0, 1, -1, small, random}
// and v any variable allocated so far (ONE included).  tests/circuits.py::random_circuit draws the same sequence.
// Test fixture (not satisfiable): the mirror's ProvingAssignment and structure capture against the oracle's.
class RandomCircuit : public Circuit {
 public:
  uint64_t seed;
  size_t rounds;
  Fr x0;
  struct Draw {
    uint64_t st;
    uint64_t next() { return ChainCircuit::splitmix(st); }
    uint64_t below(uint64_t n) { return next() % n; }
  };
  void synthesize(ConstraintSystem &cs) override {
    Draw d{seed};
    std::vector<Variable> vars;
    vars.push_back(ConstraintSystem::one());
    Fr value = x0;
    auto fresh_value = [&] { value = value * value + Fr::from_u64(d.next()); return value; };
    auto coefficient = [&]() -> Fr {
      switch (d.below(5)) {
        case 0: return Fr::zero();
        case 1: return Fr::one();
        case 2: return Fr::one().neg();
        case 3: return Fr::from_u64(d.below(16));
        default: return Fr::from_u64(d.next()) * Fr::from_u64(d.next());
      }
    };
    struct TermSpec { int op; Fr k; Variable v; };
    struct Side { int form; std::vector<TermSpec> terms; };
    auto apply = [](LinearCombination lc, const std::vector<TermSpec> &terms) {
      for (const TermSpec &t : terms) {
        switch (t.op) {
          case 0: lc = std::move(lc) + t.v; break;
          case 1: lc = std::move(lc) - t.v; break;
          case 2: lc = std::move(lc) + std::make_pair(t.k, t.v); break;
          default: lc = std::move(lc) - std::make_pair(t.k, t.v); break;
        }
      }
      return lc;
    };
    for (size_t i = 0; i < rounds; i++) {
      const uint64_t n_new = d.below(3);
      for (uint64_t j = 0; j < n_new; j++) {
        const bool input = d.below(4) == 0;
        const Fr v = fresh_value();
        vars.push_back(input ? cs.alloc_input([&] { return v; }) : cs.alloc([&] { return v; }));
      }
      Side side[3];
      for (Side &s : side) {
        s.form = (int)d.below(3);
        const uint64_t n_terms = s.form == 0 ? 0 : (s.form == 1 ? 1 + d.below(9) : d.below(10));
        for (uint64_t t = 0; t < n_terms; t++) {
          TermSpec ts;
          ts.op = (int)d.below(4);
          ts.k = ts.op >= 2 ? coefficient() : Fr::one();
          ts.v = vars[d.below(vars.size())];
          s.terms.push_back(ts);
        }
      }
      auto closure = [&](const Side &s) {
        return [&s, &apply](LinearCombination lc) {
          if (s.form == 2) return apply(LinearCombination::zero(), s.terms);   // stored; the argument is not touched
          return apply(std::move(lc), s.terms);                                // (form 0: no terms - the argument itself)
        };
      };
      cs.enforce(closure(side[0]), closure(side[1]), closure(side[2]));
    }
  }
};

// Closures that BREAK the contract of the evaluating combination (groth16.hpp LcSink): after `rounds` well-behaved
// constraints, one whose A closure - by seed % 3 - (0) adds a term to a copy of its argument and discards the copy,
// (1) builds two combinations from its argument and returns one of them, (2) touches its argument and then returns a
// stored combination.  The reference's `eval` would see only the returned terms; this library's closures are compiled with
// BELLMAN_HIP_CHECK_CLOSURES, so `enforce` must refuse each of them (std::invalid_argument -> BH_ERR_INVALID_ARG).
class MisuseCircuit : public Circuit {
 public:
  uint64_t seed;
  size_t rounds;
  Fr x0;
  void synthesize(ConstraintSystem &cs) override {
    Fr v_value = x0;
    Variable v = cs.alloc([&] { return v_value; });
    for (size_t i = 0; i < rounds; i++) {
      const Fr sq = v_value * v_value;
      Variable n = cs.alloc([&] { return sq; });
      cs.enforce([&](LinearCombination lc) { return lc + v; }, [&](LinearCombination lc) { return lc + v; },
                 [&](LinearCombination lc) { return lc + n; });
      v = n; v_value = sq;
    }
    const Variable one = ConstraintSystem::one();
    const auto same = [&](LinearCombination lc) { return lc + v; };
    switch (seed % 3) {
      case 0:
        cs.enforce([&](LinearCombination lc) { LinearCombination t = lc + one; (void)t; return lc + v; }, same, same);
        break;
      case 1:
        cs.enforce([&](LinearCombination lc) {
          LinearCombination p = lc + v, q = lc + one + v;
          return (rounds & 1) ? p : q;
        }, same, same);
        break;
      default:
        cs.enforce([&](LinearCombination lc) { LinearCombination t = std::move(lc) + v; (void)t; return LinearCombination::zero() + v; }, same, same);
        break;
    }
  }
};

}  // namespace groth16

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
  if (circuit_kind == 3) {   // structure drawn from the seed (test fixture): witness = x0, `size` rounds
    RandomCircuit c;
    c.seed = seed; c.rounds = size;
    c.x0 = Fr::zero();
    if (witness) memcpy(&c.x0, witness, 32);
    return f(c);
  }
  if (circuit_kind == 2) {   // every form of linear combination (test fixture): witness = x0, `size` rounds
    FormsCircuit c;
    c.seed = seed; c.rounds = size;
    c.x0 = Fr::zero();
    if (witness) memcpy(&c.x0, witness, 32);
    return f(c);
  }
  if (circuit_kind == 5) {   // boolean-heavy bit-mixing circuit: witness = x0 (its low 64 bits seed the state), `size` steps
    BoolMixCircuit c;
    c.seed = seed; c.rounds = size;
    c.x0 = Fr::zero();
    if (witness) memcpy(&c.x0, witness, 32);
    return f(c);
  }
  if (circuit_kind == 4) {   // closures that break the contract of the evaluating combination (test fixture)
    MisuseCircuit c;
    c.seed = seed; c.rounds = size;
    c.x0 = Fr::zero();
    if (witness) memcpy(&c.x0, witness, 32);
    return f(c);
  }
  return BH_ERR_INVALID_ARG;
}

extern "C" {

#ifndef BH_DEMO_TIMED_BUILD
void bh_test_proof_slice(size_t n, size_t part, size_t parts, size_t *lo, size_t *hi) {
  groth16::proof_slice_for_tests(n, part, parts, lo, hi);
}
#endif
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
#ifndef BH_DEMO_TIMED_BUILD   // (fixtures and host-side hooks: the checked test library only)
double bh_test_capture_check(int circuit_kind, size_t size, uint64_t seed, size_t out4[4]) {
  // host only: the structure capture (R1cs's constructor without the upload) of a demo circuit, checked against the
  // ProvingAssignment of the same circuit - captured A, B, C times the assignment == its a, b, c rows
  using namespace groth16;
  if (!out4) return -1.0;
  std::vector<Fr> constants(circuit_kind == 0 ? size : 0);
  for (size_t i = 0; i < constants.size(); i++) constants[i] = Fr::from_u64(0x9E3779B97F4A7C15ULL * (i % 7 + 1));   // repeats: the table must share them
  Fr wit[2] = {Fr::from_u64(123456789), Fr::from_u64(987654321)};
  double ms = -1.0;
  try {
  with_demo_circuit(circuit_kind, size, seed, wit, constants.data(), [&](bellman::Circuit &shape) -> int {
    return with_demo_circuit(circuit_kind, size, seed, wit, constants.data(), [&](bellman::Circuit &proved) -> int {
      ms = capture_check_for_tests(shape, proved, out4);
      return 0;
    });
  });
  } catch (...) { return -1.0; }   // (a closure refused by the checking build: MisuseCircuit)
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
  } catch (const std::invalid_argument &) { return BH_ERR_INVALID_ARG;
  } catch (...) { return BH_ERR_HIP; }
}
void bh_test_fr_from_u512_host(void *r, const void *limbs8) {
  uint64_t w[8];
  memcpy(w, limbs8, 64);
  const bellman::Fr f = bellman::Fr::from_u512(w);
  memcpy(r, &f, 32);
}
void bh_test_fr_ops_host(int op, void *r, const void *a, const void *b, size_t n) {
  using bellman::Fr;
  for (size_t i = 0; i < n; i++) {
    Fr x, y = Fr::zero(), z;
    memcpy(&x, (const char *)a + 32 * i, 32);
    if (b) memcpy(&y, (const char *)b + 32 * i, 32);
    switch (op) {
      case 0: z = x + y; break;
      case 1: z = x - y; break;
      case 2: z = x * y; break;
      case 3: z = x.neg(); break;
      case 4: z = Fr::from_u64(x.l[0]); break;
      case 5: x.to_canonical(z.l); break;
      default: z = x.invert(); break;
    }
    memcpy((char *)r + 32 * i, &z, 32);
  }
}
#endif
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
  } else if (circuit_kind == 5) {   // boolean-heavy bit mixing: witness = x0, `size` steps
    BoolMixCircuit c;
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
