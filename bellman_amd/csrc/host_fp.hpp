// Host-only BLS12-381 Fp / Fp2 arithmetic on 6x64-bit limbs (unsigned __int128), used for the short
// serial tails that run on the CPU (the 255-step double-and-add that closes an MSM, the handful of
// scalar multiplications in create_proof).  Same Montgomery representation and byte layout as the
// device types (12 x u32 little-endian == 6 x u64 little-endian), so records are memcpy-compatible
// and the curve templates of ec.cuh are reused through the HostFpOps / HostFp2Ops bundles.
#pragma once
#include <stdint.h>
#include <string.h>

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

static inline bool geq_mod(const uint64_t *a) {
  for (int i = 5; i >= 0; i--) {
    if (a[i] > MOD[i]) return true;
    if (a[i] < MOD[i]) return false;
  }
  return true;
}
static inline void sub_mod(uint64_t *a) {
  u128 br = 0;
  for (int i = 0; i < 6; i++) {
    u128 d = (u128)a[i] - MOD[i] - (uint64_t)br;
    a[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
static inline void add(hfp_t &r, const hfp_t &a, const hfp_t &b) {
  uint64_t t[6];
  u128 c = 0;
  for (int i = 0; i < 6; i++) {
    c += (u128)a.l[i] + b.l[i];
    t[i] = (uint64_t)c;
    c >>= 64;
  }
  if (geq_mod(t)) sub_mod(t);
  memcpy(r.l, t, sizeof t);
}
static inline void sub(hfp_t &r, const hfp_t &a, const hfp_t &b) {
  uint64_t t[6];
  u128 br = 0;
  for (int i = 0; i < 6; i++) {
    u128 d = (u128)a.l[i] - b.l[i] - (uint64_t)br;
    t[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  if (br) {
    u128 c = 0;
    for (int i = 0; i < 6; i++) {
      c += (u128)t[i] + MOD[i];
      t[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  memcpy(r.l, t, sizeof t);
}
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
    t[7] = (uint64_t)(c >> 64);
    const uint64_t m = t[0] * INV;
    c = ((u128)m * MOD[0] + t[0]) >> 64;
    for (int j = 1; j < 6; j++) {
      c += (u128)m * MOD[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[6];
    t[5] = (uint64_t)c;
    t[6] = t[7] + (uint64_t)(c >> 64);
  }
  if (t[6] || geq_mod(t)) sub_mod(t);
  memcpy(r.l, t, 6 * sizeof(uint64_t));
}
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
