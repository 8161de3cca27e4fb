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
#if defined(__x86_64__)
#include <x86intrin.h>   // _addcarry_u64 / _subborrow_u64 (the host-side field arithmetic below)
#endif

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
// (profiles/archive/r3_host_synthesis.txt).
namespace fr_detail {
typedef unsigned __int128 u128;
constexpr uint64_t MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
constexpr uint64_t INV = 0xfffffffeffffffffULL;
constexpr uint64_t R[4] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL};
constexpr uint64_t R2[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};
// add / subtract with carry as the instruction the CPU has for it (the 128-bit-integer spelling of a borrow chain compiles to
// three or four instructions per limb with clang and g++ alike)
#if defined(__x86_64__)
typedef unsigned char carry_t;
__attribute__((always_inline)) inline uint64_t adc(uint64_t a, uint64_t b, carry_t &c) {
  unsigned long long r; c = _addcarry_u64(c, a, b, &r); return r;
}
__attribute__((always_inline)) inline uint64_t sbb(uint64_t a, uint64_t b, carry_t &c) {
  unsigned long long r; c = _subborrow_u64(c, a, b, &r); return r;
}
#else
typedef uint64_t carry_t;
__attribute__((always_inline)) inline uint64_t adc(uint64_t a, uint64_t b, carry_t &c) {
  const u128 x = (u128)a + b + c; c = (uint64_t)(x >> 64); return (uint64_t)x;
}
__attribute__((always_inline)) inline uint64_t sbb(uint64_t a, uint64_t b, carry_t &c) {
  const u128 x = (u128)a - b - c; c = (uint64_t)(x >> 64) & 1; return (uint64_t)x;
}
#endif
__attribute__((always_inline)) inline uint64_t sel(uint64_t mask, uint64_t if_set, uint64_t if_clear) {
  return if_clear ^ ((if_clear ^ if_set) & mask);
}
// Whether a sum, a difference or a product needs its final correction is a coin flip for random field elements: the
// corrections below are selections by mask, not branches (a mispredicted branch costs more than the addition itself).
// t < 2q in four limbs -> [0, q)
__attribute__((always_inline)) inline void final_sub(uint64_t *r, uint64_t t0, uint64_t t1, uint64_t t2, uint64_t t3) {
  carry_t b = 0;
  const uint64_t d0 = sbb(t0, MOD[0], b), d1 = sbb(t1, MOD[1], b), d2 = sbb(t2, MOD[2], b), d3 = sbb(t3, MOD[3], b);
  const uint64_t keep = 0 - (uint64_t)b;   // all ones: t < q
  r[0] = sel(keep, t0, d0); r[1] = sel(keep, t1, d1); r[2] = sel(keep, t2, d2); r[3] = sel(keep, t3, d3);
}
__attribute__((always_inline)) inline void add_mod(uint64_t *r, const uint64_t *a, const uint64_t *b) {   // a, b < q
  carry_t c = 0;
  const uint64_t s0 = adc(a[0], b[0], c), s1 = adc(a[1], b[1], c), s2 = adc(a[2], b[2], c), s3 = adc(a[3], b[3], c);
  final_sub(r, s0, s1, s2, s3);   // q < 2^255: the sum has no carry out
}
__attribute__((always_inline)) inline void sub_mod(uint64_t *r, const uint64_t *a, const uint64_t *b) {   // a, b < q
  carry_t c = 0;
  const uint64_t d0 = sbb(a[0], b[0], c), d1 = sbb(a[1], b[1], c), d2 = sbb(a[2], b[2], c), d3 = sbb(a[3], b[3], c);
  const uint64_t m = 0 - (uint64_t)c;   // borrowed: add q back
  c = 0;
  r[0] = adc(d0, MOD[0] & m, c); r[1] = adc(d1, MOD[1] & m, c); r[2] = adc(d2, MOD[2] & m, c); r[3] = adc(d3, MOD[3] & m, c);
}
// 4x64 CIOS Montgomery product, fully unrolled (synthesis is the serial part of create_proof): r = a * b / 2^256 mod q.
// PRECONDITION: a < q; b may be ANY 256-bit value (every running sum then stays below 2q + q * 2^64 - five limbs inside a
// row, four between rows - and the result is below 2q before its correction).
#if defined(__x86_64__) && defined(__BMI2__) && defined(__ADX__)
// mulx + the two independent carry chains of adcx / adox: one row of the product and one reduction step per asm block
// (compilers do not generate dual carry chains from C: the u128 form below compiles to ~250 instructions, this to ~150;
// 25 instead of 38-41 ns per product in a dependent chain on the build host, tools/host_synthesis.py for the whole).
#define BH_FR_ROW0(bi)                                                                                               \
  asm("mulx %[a0], %[t0], %[t1]\n\t"                                                                                 \
      "mulx %[a1], %%rax, %[t2]\n\t add %%rax, %[t1]\n\t"                                                            \
      "mulx %[a2], %%rax, %[t3]\n\t adc %%rax, %[t2]\n\t"                                                            \
      "mulx %[a3], %%rax, %[t4]\n\t adc %%rax, %[t3]\n\t adc $0, %[t4]\n\t"                                          \
      : [t0] "=&r"(t0), [t1] "=&r"(t1), [t2] "=&r"(t2), [t3] "=&r"(t3), [t4] "=&r"(t4)                               \
      : "d"(bi), [a0] "m"(a[0]), [a1] "m"(a[1]), [a2] "m"(a[2]), [a3] "m"(a[3])                                      \
      : "rax", "cc")
#define BH_FR_ROW(bi)                                                                                                \
  asm("xorl %%eax, %%eax\n\t"                                                                                        \
      "mulx %[a0], %[lo], %[hi]\n\t adcx %[lo], %[t0]\n\t adox %[hi], %[t1]\n\t"                                     \
      "mulx %[a1], %[lo], %[hi]\n\t adcx %[lo], %[t1]\n\t adox %[hi], %[t2]\n\t"                                     \
      "mulx %[a2], %[lo], %[hi]\n\t adcx %[lo], %[t2]\n\t adox %[hi], %[t3]\n\t"                                     \
      "mulx %[a3], %[lo], %[t4]\n\t adcx %[lo], %[t3]\n\t"                                                           \
      "mov $0, %[lo]\n\t adox %[lo], %[t4]\n\t adcx %[lo], %[t4]\n\t"                                                \
      : [t0] "+r"(t0), [t1] "+r"(t1), [t2] "+r"(t2), [t3] "+r"(t3), [t4] "=&r"(t4), [lo] "=&r"(lo), [hi] "=&r"(hi)   \
      : "d"(bi), [a0] "m"(a[0]), [a1] "m"(a[1]), [a2] "m"(a[2]), [a3] "m"(a[3])                                      \
      : "rax", "cc")
// t += (t0 * INV mod 2^64) * q, then one limb down (the low limb is zero by construction)
#define BH_FR_REDUCE()                                                                                               \
  {                                                                                                                  \
    const uint64_t m = t0 * INV;                                                                                     \
    asm("xorl %%eax, %%eax\n\t"                                                                                      \
        "mulx %[q0], %[lo], %[hi]\n\t adcx %[t0], %[lo]\n\t adox %[hi], %[t1]\n\t"                                   \
        "mulx %[q1], %[lo], %[hi]\n\t adcx %[lo], %[t1]\n\t adox %[hi], %[t2]\n\t"                                   \
        "mulx %[q2], %[lo], %[hi]\n\t adcx %[lo], %[t2]\n\t adox %[hi], %[t3]\n\t"                                   \
        "mulx %[q3], %[lo], %[hi]\n\t adcx %[lo], %[t3]\n\t adox %[hi], %[t4]\n\t"                                   \
        "mov $0, %[lo]\n\t adcx %[lo], %[t4]\n\t"                                                                    \
        : [t0] "+r"(t0), [t1] "+r"(t1), [t2] "+r"(t2), [t3] "+r"(t3), [t4] "+r"(t4), [lo] "=&r"(lo), [hi] "=&r"(hi)  \
        : "d"(m), [q0] "m"(MOD[0]), [q1] "m"(MOD[1]), [q2] "m"(MOD[2]), [q3] "m"(MOD[3])                             \
        : "rax", "cc");                                                                                              \
    t0 = t1; t1 = t2; t2 = t3; t3 = t4; t4 = 0;                                                                      \
  }
__attribute__((always_inline)) inline void mont_mul(uint64_t *r, const uint64_t *a, const uint64_t *b) {
  uint64_t t0, t1, t2, t3, t4, lo, hi;
  BH_FR_ROW0(b[0]); BH_FR_REDUCE();
  BH_FR_ROW(b[1]); BH_FR_REDUCE();
  BH_FR_ROW(b[2]); BH_FR_REDUCE();
  BH_FR_ROW(b[3]); BH_FR_REDUCE();
  final_sub(r, t0, t1, t2, t3);
}
#undef BH_FR_ROW0
#undef BH_FR_ROW
#undef BH_FR_REDUCE
#else
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
  final_sub(r, t0, t1, t2, t3);
}
#undef BH_FR_ROW
#endif
// v * 2^256 mod q for a 64-bit v - Fr::from_u64, what a circuit calls for every constant it writes down (bls12_381's
// `From<u64> for Scalar` is a full Montgomery product with R^2: 36 limb products; the one-limb form of it that round 4
// used: 24).  Here: P = v * (2^256 mod q) < 2^64 q, its quotient by q estimated from the top 64 bits of P with
// mu = floor(2^319 / q) = 2^64 + MU_LO (Barrett; the estimate is short by at most 3), and P - Q q corrected by 2q and q
// without branches: 9 limb products.  The synthetic chain circuit of BASELINE config C4 spends two of its three products
// per constraint here.
constexpr uint64_t MU_LO = 0x1aa84a76ff6f1bbeULL;
__attribute__((always_inline)) inline void mont_from_u64(uint64_t *r, uint64_t v) {
  // P = v * R (five limbs)
  u128 c = (u128)R[0] * v; const uint64_t p0 = (uint64_t)c;
  c = (c >> 64) + (u128)R[1] * v; const uint64_t p1 = (uint64_t)c;
  c = (c >> 64) + (u128)R[2] * v; const uint64_t p2 = (uint64_t)c;
  c = (c >> 64) + (u128)R[3] * v; const uint64_t p3 = (uint64_t)c, p4 = (uint64_t)(c >> 64);
  // Q = floor(floor(P / 2^255) * mu / 2^64): Q <= floor(P / q) <= Q + 3
  const uint64_t ph = (p4 << 1) | (p3 >> 63);
  const uint64_t qh = ph + (uint64_t)(((u128)ph * MU_LO) >> 64);
  // t = P - Q q in [0, 4q): 258 bits (the fifth limb holds two of them)
  c = (u128)MOD[0] * qh; const uint64_t m0 = (uint64_t)c;
  c = (c >> 64) + (u128)MOD[1] * qh; const uint64_t m1 = (uint64_t)c;
  c = (c >> 64) + (u128)MOD[2] * qh; const uint64_t m2 = (uint64_t)c;
  c = (c >> 64) + (u128)MOD[3] * qh; const uint64_t m3 = (uint64_t)c, m4 = (uint64_t)(c >> 64);
  carry_t b = 0;
  uint64_t t0 = sbb(p0, m0, b), t1 = sbb(p1, m1, b), t2 = sbb(p2, m2, b), t3 = sbb(p3, m3, b), t4 = sbb(p4, m4, b);
  // minus 2q if that does not borrow, then minus q if that does not borrow (selections by mask: data, not branches)
  constexpr uint64_t Q2[4] = {MOD[0] << 1, (MOD[1] << 1) | (MOD[0] >> 63), (MOD[2] << 1) | (MOD[1] >> 63), (MOD[3] << 1) | (MOD[2] >> 63)};
  {
    b = 0;
    const uint64_t d0 = sbb(t0, Q2[0], b), d1 = sbb(t1, Q2[1], b), d2 = sbb(t2, Q2[2], b), d3 = sbb(t3, Q2[3], b);
    (void)sbb(t4, 0, b);
    const uint64_t keep = 0 - (uint64_t)b;   // all ones: it borrowed, keep t
    t0 = sel(keep, t0, d0); t1 = sel(keep, t1, d1); t2 = sel(keep, t2, d2); t3 = sel(keep, t3, d3);
  }
  {   // now t < 2q < 2^256: the fifth limb is zero
    b = 0;
    const uint64_t d0 = sbb(t0, MOD[0], b), d1 = sbb(t1, MOD[1], b), d2 = sbb(t2, MOD[2], b), d3 = sbb(t3, MOD[3], b);
    const uint64_t keep = 0 - (uint64_t)b;
    r[0] = sel(keep, t0, d0); r[1] = sel(keep, t1, d1); r[2] = sel(keep, t2, d2); r[3] = sel(keep, t3, d3);
  }
}
}  // namespace fr_detail

struct Fr {
  uint64_t l[4];
  static Fr zero() { return Fr{{0, 0, 0, 0}}; }
  static Fr one() { return Fr{{fr_detail::R[0], fr_detail::R[1], fr_detail::R[2], fr_detail::R[3]}}; }
  static Fr from_u64(uint64_t v) {
    Fr r;
    fr_detail::mont_from_u64(r.l, v);
    return r;
  }
  static Fr from_u512(const uint64_t limbs_le[8]);   // wide reduction, as ff's Field::random does
  bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
  bool operator==(const Fr &o) const { return ((l[0] ^ o.l[0]) | (l[1] ^ o.l[1]) | (l[2] ^ o.l[2]) | (l[3] ^ o.l[3])) == 0; }
  bool operator!=(const Fr &o) const { return !(*this == o); }
  __attribute__((always_inline)) Fr operator+(const Fr &o) const { Fr r; fr_detail::add_mod(r.l, l, o.l); return r; }
  __attribute__((always_inline)) Fr operator-(const Fr &o) const { Fr r; fr_detail::sub_mod(r.l, l, o.l); return r; }
  __attribute__((always_inline)) Fr operator*(const Fr &o) const { Fr r; fr_detail::mont_mul(r.l, l, o.l); return r; }
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
// One 64-bit word (the index, with the kind in the top bit), not the two-field struct of the reference's `Variable(Index)`:
// circuits copy variables all the time (`x = next;`), a 16-byte struct is copied with ONE vector load, and a variable that
// `alloc` has just returned in two registers sits in memory as two 8-byte stores - which cannot be forwarded to that load
// (6 % of the witness generation of a 2^20-constraint chain, tools/host_profile.py).
struct Variable {
  uint64_t bits;
  static Variable new_unchecked(Index k, size_t i) { return Variable{(uint64_t)i | (k == Index::Aux ? uint64_t(1) << 63 : 0)}; }
  Index kind() const { return (bits >> 63) ? Index::Aux : Index::Input; }
  size_t idx() const { return (size_t)(bits & ~(uint64_t(1) << 63)); }
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
// second pass.  Same values, same densities, same term order (profiles/archive/r3_host_synthesis.txt).
struct LcSink {
  // input_assignment / aux_assignment: the VECTORS, read per term - a closure that allocates a variable while it builds
  // its combination (the reference's borrow rules forbid it, C++ does not) may reallocate them
  const std::vector<Fr> *inputs, *aux;
  DensityTracker *input_density, *aux_density;    // either may be null (prover.rs:119-141)
  // set instead of the four fields above: every term is handed to `hook` as it is added (a ConstraintSystem that
  // records the structure of the circuit - the R1CS capture - takes the terms straight into its matrices)
  void (*hook)(void *self, Variable v, const Fr &coeff) = nullptr;
  void *self = nullptr;
  // the running value of the combination bound to this sink.  It lives HERE, not in the LinearCombination objects that
  // travel through `|lc| lc + a + b` (one copy, one or two moves and a return per closure): those then carry two words,
  // and the accumulator is updated in place instead of being copied from object to object between additions (synthesis
  // of the 2^20-constraint chain 139 -> 125 ms in the build container).  Consequence, and the CONTRACT of enforce's closures:
  // every term added to ANY copy of the closure's argument counts - in the value as in the density maps.  A closure returns a
  // combination derived linearly from its argument (`|lc| lc + a + (c, b)`, as every closure of the reference's circuits and
  // gadgets does; in Rust `lc + a` consumes `lc`, so using it twice takes an explicit clone) or a stored combination that
  // it built from LinearCombination::zero() without touching its argument.  Counting the terms in the sink as well, to
  // detect the misuse, was measured at 5 % of the synthesis: it is compiled into the closures of a translation unit built
  // with -DBELLMAN_HIP_CHECK_CLOSURES=1 (the test library, a user's debug build) and nowhere else; `enforce` compares
  // whenever the closure's unit counted (pushed != 0) and throws std::invalid_argument for a closure that breaks the contract.
  mutable Fr acc = Fr::zero();
  mutable size_t pushed = 0;   // terms added through ANY copy of the closure's argument (counted by checking builds only)
};
// Copy of a field element that was just computed: limb by limb through general registers.  A struct copy compiles to two
// 16-byte vector moves, and a 16-byte load from a location that two 8-byte stores have just written cannot be forwarded
// from the store buffer - it waits until both stores have reached the cache (12+ cycles, once per copy of a combination
// whose accumulator the previous term has just updated).
__attribute__((always_inline)) inline void copy_fresh(Fr &dst, const Fr &src) {
  uint64_t a = src.l[0], b = src.l[1], c = src.l[2], d = src.l[3];
#if defined(__GNUC__)
  asm("" : "+r"(a), "+r"(b), "+r"(c), "+r"(d));   // (keeps the four loads scalar)
#endif
  dst.l[0] = a; dst.l[1] = b; dst.l[2] = c; dst.l[3] = d;
}
class LinearCombination {
 public:
  struct Term { Variable first; Fr second; };     // (variable, coefficient); an aggregate, so copies are plain memcpy
  static LinearCombination zero() { return LinearCombination(); }
  static LinearCombination evaluating(const LcSink *sink) {
    LinearCombination r;
    r.sink_ = sink;
    sink->acc = Fr::zero();
    sink->pushed = 0;
    return r;
  }
  LinearCombination() : n_(0), sink_(nullptr) {}
  LinearCombination(const LinearCombination &o) : n_(o.n_), sink_(o.sink_) {
    if (!sink_) { memcpy(inl_, o.inl_, sizeof inl_); more_ = o.more_; }
  }
  LinearCombination(LinearCombination &&o) noexcept : n_(o.n_), sink_(o.sink_) {
    if (!sink_) { memcpy(inl_, o.inl_, sizeof inl_); more_ = std::move(o.more_); }
  }
  LinearCombination &operator=(const LinearCombination &o) {
    if (this == &o) return *this;
    n_ = o.n_; sink_ = o.sink_;
    if (!sink_) { memcpy(inl_, o.inl_, sizeof inl_); more_ = o.more_; }
    return *this;
  }
  LinearCombination &operator=(LinearCombination &&o) noexcept {
    if (this == &o) return *this;
    n_ = o.n_; sink_ = o.sink_;
    if (!sink_) { memcpy(inl_, o.inl_, sizeof inl_); more_ = std::move(o.more_); }
    return *this;
  }
  // (always_inline: as calls from a closure into the header's out-of-line copy they cost 4 % of a 2^20-constraint synthesis)
  // lvalue operands are copied (value semantics); a temporary is extended in place and MOVED out - returned by value,
  // so that `lc = std::move(lc) + x` is not a self-move and `auto &&r = zero() + a` does not dangle (moving an
  // evaluating combination is a 56-byte copy)
  __attribute__((always_inline)) LinearCombination operator+(Variable v) const & { LinearCombination r(*this); r.push_one(v); return r; }
  __attribute__((always_inline)) LinearCombination operator+(Variable v) && { push_one(v); return std::move(*this); }
  __attribute__((always_inline)) LinearCombination operator-(Variable v) const & { LinearCombination r(*this); r.push(v, Fr::one().neg()); return r; }
  __attribute__((always_inline)) LinearCombination operator-(Variable v) && { push(v, Fr::one().neg()); return std::move(*this); }
  // ((coefficient, variable) terms by reference: a 56-byte pair passed by value is copied with 16-byte loads that straddle
  //  the 8-byte stores which have just built it - a failed store forwarding per term)
  __attribute__((always_inline)) LinearCombination operator+(const std::pair<Fr, Variable> &t) const & { LinearCombination r(*this); r.push(t.second, t.first); return r; }
  __attribute__((always_inline)) LinearCombination operator+(const std::pair<Fr, Variable> &t) && { push(t.second, t.first); return std::move(*this); }
  __attribute__((always_inline)) LinearCombination operator-(const std::pair<Fr, Variable> &t) const & { LinearCombination r(*this); r.push(t.second, t.first.neg()); return r; }
  __attribute__((always_inline)) LinearCombination operator-(const std::pair<Fr, Variable> &t) && { push(t.second, t.first.neg()); return std::move(*this); }
  size_t size() const { return n_; }
  // stored combinations only
  const Term &operator[](size_t i) const { return i < INLINE ? inl_[i] : more_[i - INLINE]; }
  bool is_evaluating() const { return sink_ != nullptr; }
  Fr value() const { Fr r; copy_fresh(r, sink_->acc); return r; }   // evaluating combinations only
  void value_into(Fr &dst) const { copy_fresh(dst, sink_->acc); }

 private:
  static constexpr size_t INLINE = 4;
  // The evaluating form is the hot path of create_proof's synthesis: it is inlined into the closures, with the two rare
  // cases (a term that needs a product; a term that is stored) out of line so that a closure stays a few dozen
  // instructions per term.
  __attribute__((always_inline)) const Fr *locate(Variable v) const {   // the variable's value; counts it in the density map
    const size_t i = v.idx();
    if (v.kind() == Index::Input) {
      if (sink_->input_density) sink_->input_density->inc(i);
      return sink_->inputs->data() + i;
    }
    if (sink_->aux_density) sink_->aux_density->inc(i);
    return sink_->aux->data() + i;
  }
  __attribute__((always_inline)) void push(Variable v, const Fr &c) {
    n_++;
    if (__builtin_expect(sink_ != nullptr, 1)) {   // prover.rs:19-55 for this one term
#ifdef BELLMAN_HIP_CHECK_CLOSURES
      sink_->pushed++;
#endif
      if (c.is_zero()) return;            // zero coefficients count for neither value nor density (:31)
      if (__builtin_expect(sink_->hook != nullptr, 0)) { sink_->hook(sink_->self, v, c); return; }
      const Fr *value = locate(v);
      // most terms carry the coefficient one (`lc + x`: push_one), then the ubiquitous `(c, CS::one())` terms: value one
      Fr &acc = sink_->acc;
      if (*value == Fr::one()) acc = acc + c;
      else if (c == Fr::one()) acc = acc + *value;
      else add_product(*value, c);
      return;
    }
    push_stored(v, c);
  }
  __attribute__((always_inline)) void push_one(Variable v) {   // coefficient one
    if (__builtin_expect(sink_ != nullptr && sink_->hook == nullptr, 1)) {
      n_++;
#ifdef BELLMAN_HIP_CHECK_CLOSURES
      sink_->pushed++;
#endif
      sink_->acc = sink_->acc + *locate(v);
      return;
    }
    push(v, Fr::one());
  }
  __attribute__((noinline)) void add_product(const Fr &value, const Fr &c) { sink_->acc = sink_->acc + value * c; }
  __attribute__((noinline)) void push_stored(Variable v, const Fr &c) {
    if (n_ <= INLINE) inl_[n_ - 1] = Term{v, c}; else more_.push_back(Term{v, c});
  }
  Term inl_[INLINE];
  std::vector<Term> more_;
  size_t n_;
  const LcSink *sink_;
};

// Non-owning callable reference (two pointers, never allocates): the C++ stand-in for the
// monomorphised `FnOnce` parameters of ConstraintSystem::{alloc, enforce} (src/lib.rs:385-416).
// The referenced callable only has to live for the duration of the call it is passed to.
template <class Sig> class FunctionRef;
template <class R, class... Args>
class FunctionRef<R(Args...)> {
 public:
  template <class F>
  FunctionRef(F &&f) : obj_((void *)&f), call_([](void *o, Args &&...args) -> R {
    return (*reinterpret_cast<typename std::remove_reference<F>::type *>(o))(std::forward<Args>(args)...);
  }) {}
  R operator()(Args... args) const { return call_(obj_, std::forward<Args>(args)...); }

 private:
  void *obj_;
  R (*call_)(void *, Args &&...);   // (arguments travel by reference up to the callable: one move of a combination less)
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
  // (the three callable references travel by reference: by value the third one goes over the stack as one 16-byte vector
  //  load of two 8-byte stores, which the store buffer cannot forward - 4 % of a 2^20-constraint synthesis)
  virtual void enforce(const LcFn &a, const LcFn &b, const LcFn &c) = 0;
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
  void enforce(const bellman::LcFn &a, const bellman::LcFn &b, const bellman::LcFn &c) override;
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
  void enforce(const bellman::LcFn &, const bellman::LcFn &, const bellman::LcFn &) override {}
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
