// Host-only BLS12-381 Fp / Fp2 arithmetic on 6x64-bit limbs (unsigned __int128), used for the short
// serial tails that run on the CPU (the 255-step double-and-add that closes an MSM, the handful of
// scalar multiplications in create_proof).  Same Montgomery representation and byte layout as the
// device types (12 x u32 little-endian == 6 x u64 little-endian), so records are memcpy-compatible
// and the curve templates of ec.cuh are reused through the HostFpOps / HostFp2Ops bundles.
#pragma once
#include <stdint.h>
#include <string.h>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <x86intrin.h>   // _addcarry_u64 / _subborrow_u64
#endif

namespace bh {

struct alignas(16) hfp_t {
  uint64_t l[6];
};
struct alignas(16) hfp2_t {
  hfp_t c0, c1;
};

namespace hostfp {
typedef unsigned __int128 u128;
static const uint64_t MOD[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const uint64_t INV = 0x89f3fffcfffcfffdULL;
static const uint64_t ONE[6] = {0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL,
                                0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL};

// Operands are canonical (below p) everywhere on the host.  The final corrections are selections by mask, not branches:
// whether a random sum or product needs its correction is a coin flip, and these functions run in dependent chains (the
// doubling ladder that closes every multiexp, create_proof's few scalar multiplications) where a mispredicted branch is
// paid in full.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
typedef unsigned char carry_t;
static inline __attribute__((always_inline)) uint64_t adc(uint64_t a, uint64_t b, carry_t &c) { unsigned long long r; c = _addcarry_u64(c, a, b, &r); return r; }
static inline __attribute__((always_inline)) uint64_t sbb(uint64_t a, uint64_t b, carry_t &c) { unsigned long long r; c = _subborrow_u64(c, a, b, &r); return r; }
#else
typedef uint64_t carry_t;
static inline uint64_t adc(uint64_t a, uint64_t b, carry_t &c) { const u128 x = (u128)a + b + c; c = (uint64_t)(x >> 64); return (uint64_t)x; }
static inline uint64_t sbb(uint64_t a, uint64_t b, carry_t &c) { const u128 x = (u128)a - b - c; c = (uint64_t)(x >> 64) & 1; return (uint64_t)x; }
#endif
// t < 2p in six limbs -> [0, p)
static inline __attribute__((always_inline)) void final_sub(uint64_t *r, const uint64_t *t) {
  carry_t b = 0;
  uint64_t d[6];
  d[0] = sbb(t[0], MOD[0], b); d[1] = sbb(t[1], MOD[1], b); d[2] = sbb(t[2], MOD[2], b);
  d[3] = sbb(t[3], MOD[3], b); d[4] = sbb(t[4], MOD[4], b); d[5] = sbb(t[5], MOD[5], b);
  const uint64_t keep = 0 - (uint64_t)b;   // all ones: t < p
  for (int i = 0; i < 6; i++) r[i] = d[i] ^ ((d[i] ^ t[i]) & keep);
}
static inline void add(hfp_t &r, const hfp_t &a, const hfp_t &b) {
  carry_t c = 0;
  uint64_t t[6];
  t[0] = adc(a.l[0], b.l[0], c); t[1] = adc(a.l[1], b.l[1], c); t[2] = adc(a.l[2], b.l[2], c);
  t[3] = adc(a.l[3], b.l[3], c); t[4] = adc(a.l[4], b.l[4], c); t[5] = adc(a.l[5], b.l[5], c);   // 2p < 2^384: no carry out
  final_sub(r.l, t);
}
static inline void sub(hfp_t &r, const hfp_t &a, const hfp_t &b) {
  carry_t c = 0;
  uint64_t t[6];
  t[0] = sbb(a.l[0], b.l[0], c); t[1] = sbb(a.l[1], b.l[1], c); t[2] = sbb(a.l[2], b.l[2], c);
  t[3] = sbb(a.l[3], b.l[3], c); t[4] = sbb(a.l[4], b.l[4], c); t[5] = sbb(a.l[5], b.l[5], c);
  const uint64_t m = 0 - (uint64_t)c;   // borrowed: add p back
  c = 0;
  r.l[0] = adc(t[0], MOD[0] & m, c); r.l[1] = adc(t[1], MOD[1] & m, c); r.l[2] = adc(t[2], MOD[2] & m, c);
  r.l[3] = adc(t[3], MOD[3] & m, c); r.l[4] = adc(t[4], MOD[4] & m, c); r.l[5] = adc(t[5], MOD[5] & m, c);
}
// 6x64 CIOS Montgomery product, r = a * b / 2^384 mod p; a < p, b < 2^384 (every running sum then fits seven limbs inside a
// row and six between rows).
#if defined(__x86_64__) && defined(__BMI2__) && defined(__ADX__) && !defined(__HIP_DEVICE_COMPILE__)
// mulx with the two carry chains of adcx / adox (compilers do not produce them from C): one product row and one reduction
// step per asm block; the Makefile passes -mbmi2 -madx to the HOST half of the .hip files for this.
#define BH_HFP_MULADD(src, tlo, thi) "mulx " src ", %[lo], %[hi]\n\t adcx %[lo], " tlo "\n\t adox %[hi], " thi "\n\t"
#define BH_HFP_ROW(bi)                                                                                               \
  asm("xorl %%eax, %%eax\n\t"                                                                                        \
      BH_HFP_MULADD("%[a0]", "%[t0]", "%[t1]") BH_HFP_MULADD("%[a1]", "%[t1]", "%[t2]")                              \
      BH_HFP_MULADD("%[a2]", "%[t2]", "%[t3]") BH_HFP_MULADD("%[a3]", "%[t3]", "%[t4]")                              \
      BH_HFP_MULADD("%[a4]", "%[t4]", "%[t5]")                                                                       \
      "mulx %[a5], %[lo], %[t6]\n\t adcx %[lo], %[t5]\n\t"                                                           \
      "mov $0, %[lo]\n\t adox %[lo], %[t6]\n\t adcx %[lo], %[t6]\n\t"                                                \
      : [t0] "+r"(t0), [t1] "+r"(t1), [t2] "+r"(t2), [t3] "+r"(t3), [t4] "+r"(t4), [t5] "+r"(t5), [t6] "=&r"(t6),    \
        [lo] "=&r"(lo), [hi] "=&r"(hi)                                                                               \
      : "d"(bi), [a0] "m"(a.l[0]), [a1] "m"(a.l[1]), [a2] "m"(a.l[2]), [a3] "m"(a.l[3]), [a4] "m"(a.l[4]),           \
        [a5] "m"(a.l[5])                                                                                             \
      : "rax", "cc")
#define BH_HFP_REDUCE()                                                                                              \
  {                                                                                                                  \
    const uint64_t m = t0 * INV;                                                                                     \
    asm("xorl %%eax, %%eax\n\t"                                                                                      \
        "mulx %[q0], %[lo], %[hi]\n\t adcx %[t0], %[lo]\n\t adox %[hi], %[t1]\n\t"                                   \
        BH_HFP_MULADD("%[q1]", "%[t1]", "%[t2]") BH_HFP_MULADD("%[q2]", "%[t2]", "%[t3]")                            \
        BH_HFP_MULADD("%[q3]", "%[t3]", "%[t4]") BH_HFP_MULADD("%[q4]", "%[t4]", "%[t5]")                            \
        BH_HFP_MULADD("%[q5]", "%[t5]", "%[t6]")                                                                     \
        "mov $0, %[lo]\n\t adcx %[lo], %[t6]\n\t"                                                                    \
        : [t0] "+r"(t0), [t1] "+r"(t1), [t2] "+r"(t2), [t3] "+r"(t3), [t4] "+r"(t4), [t5] "+r"(t5), [t6] "+r"(t6),   \
          [lo] "=&r"(lo), [hi] "=&r"(hi)                                                                             \
        : "d"(m), [q0] "m"(MOD[0]), [q1] "m"(MOD[1]), [q2] "m"(MOD[2]), [q3] "m"(MOD[3]), [q4] "m"(MOD[4]),          \
          [q5] "m"(MOD[5])                                                                                           \
        : "rax", "cc");                                                                                              \
    t0 = t1; t1 = t2; t2 = t3; t3 = t4; t4 = t5; t5 = t6;                                                            \
  }
static inline void mul(hfp_t &r, const hfp_t &a, const hfp_t &b) {
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6, lo, hi;
  BH_HFP_ROW(b.l[0]); BH_HFP_REDUCE();
  BH_HFP_ROW(b.l[1]); BH_HFP_REDUCE();
  BH_HFP_ROW(b.l[2]); BH_HFP_REDUCE();
  BH_HFP_ROW(b.l[3]); BH_HFP_REDUCE();
  BH_HFP_ROW(b.l[4]); BH_HFP_REDUCE();
  BH_HFP_ROW(b.l[5]); BH_HFP_REDUCE();
  const uint64_t t[6] = {t0, t1, t2, t3, t4, t5};
  final_sub(r.l, t);
}
#undef BH_HFP_MULADD
#undef BH_HFP_ROW
#undef BH_HFP_REDUCE
#else
static inline void mul(hfp_t &r, const hfp_t &a, const hfp_t &b) {
  uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 6; i++) {
    u128 c = 0;
    for (int j = 0; j < 6; j++) {
      c += (u128)a.l[j] * b.l[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[6];
    t[6] = (uint64_t)c;
    const uint64_t m = t[0] * INV;
    c = ((u128)m * MOD[0] + t[0]) >> 64;
    for (int j = 1; j < 6; j++) {
      c += (u128)m * MOD[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[6];
    t[5] = (uint64_t)c;
    t[6] = 0;
  }
  final_sub(r.l, t);
}
#endif
static inline bool is_zero(const hfp_t &a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3] | a.l[4] | a.l[5]) == 0; }
static inline bool eq(const hfp_t &a, const hfp_t &b) {
  uint64_t o = 0;
  for (int i = 0; i < 6; i++) o |= a.l[i] ^ b.l[i];
  return o == 0;
}
static inline void inv(hfp_t &r, const hfp_t &a) {  // a^(p-2)
  uint64_t e[6];
  memcpy(e, MOD, sizeof e);
  e[0] -= 2;
  hfp_t acc;
  memcpy(acc.l, ONE, sizeof ONE);
  for (int i = 383; i >= 0; i--) {
    mul(acc, acc, acc);
    if ((e[i / 64] >> (i % 64)) & 1) mul(acc, acc, a);
  }
  r = acc;
}
}  // namespace hostfp

struct HostFpOps {
  typedef hfp_t T;
  static constexpr bool FUSED_Y3_TAIL = false, FUSED_Y3 = false;   // (the fused last line of ec.cuh's mixed addition: device only)
  static void zero(T &r) { memset(&r, 0, sizeof r); }
  static void one(T &r) { memcpy(r.l, hostfp::ONE, sizeof hostfp::ONE); }
  static bool is_zero(const T &a) { return hostfp::is_zero(a); }
  static bool is_zero_canonical(const T &a, const T &b) { return hostfp::is_zero(a) && hostfp::is_zero(b); }
  static bool eq(const T &a, const T &b) { return hostfp::eq(a, b); }
  static void add(T &r, const T &a, const T &b) { hostfp::add(r, a, b); }
  static void sub(T &r, const T &a, const T &b) { hostfp::sub(r, a, b); }
  static void neg(T &r, const T &a) { T z; zero(z); hostfp::sub(r, z, a); }
  static void dbl(T &r, const T &a) { hostfp::add(r, a, a); }
  static void canon(T &) {}   // host values are always canonical
  static void mul(T &r, const T &a, const T &b) { hostfp::mul(r, a, b); }
  static void sqr(T &r, const T &a) { hostfp::mul(r, a, a); }
  static void inv(T &r, const T &a) { hostfp::inv(r, a); }
};

struct HostFp2Ops {
  typedef hfp2_t T;
  typedef HostFpOps B;
  static constexpr bool FUSED_Y3_TAIL = false, FUSED_Y3 = false;
  static void zero(T &r) { memset(&r, 0, sizeof r); }
  static void one(T &r) { B::one(r.c0); B::zero(r.c1); }
  static bool is_zero(const T &a) { return B::is_zero(a.c0) && B::is_zero(a.c1); }
  static bool is_zero_canonical(const T &a, const T &b) { return is_zero(a) && is_zero(b); }
  static bool eq(const T &a, const T &b) { return B::eq(a.c0, b.c0) && B::eq(a.c1, b.c1); }
  static void add(T &r, const T &a, const T &b) { B::add(r.c0, a.c0, b.c0); B::add(r.c1, a.c1, b.c1); }
  static void sub(T &r, const T &a, const T &b) { B::sub(r.c0, a.c0, b.c0); B::sub(r.c1, a.c1, b.c1); }
  static void neg(T &r, const T &a) { B::neg(r.c0, a.c0); B::neg(r.c1, a.c1); }
  static void dbl(T &r, const T &a) { add(r, a, a); }
  static void canon(T &) {}   // host values are always canonical
  static void mul(T &r, const T &a, const T &b) {
    hfp_t t0, t1, t2, t3;
    B::mul(t0, a.c0, b.c0);
    B::mul(t1, a.c1, b.c1);
    B::add(t2, a.c0, a.c1);
    B::add(t3, b.c0, b.c1);
    B::mul(t2, t2, t3);
    B::sub(t2, t2, t0);
    B::sub(r.c1, t2, t1);
    B::sub(r.c0, t0, t1);
  }
  static void sqr(T &r, const T &a) {
    hfp_t s, d, p;
    B::add(s, a.c0, a.c1);
    B::sub(d, a.c0, a.c1);
    B::mul(p, a.c0, a.c1);
    B::mul(r.c0, s, d);
    B::add(r.c1, p, p);
  }
  static void inv(T &r, const T &a) {
    hfp_t n, t;
    B::sqr(n, a.c0);
    B::sqr(t, a.c1);
    B::add(n, n, t);
    B::inv(n, n);
    B::mul(r.c0, a.c0, n);
    B::mul(t, a.c1, n);
    B::neg(r.c1, t);
  }
};

// device ops bundle -> host ops bundle with the same record layout
template <class F> struct HostOf;

}  // namespace bh
