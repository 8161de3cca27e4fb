// bellman::Fr: the scalar field on the host (4 x 64-bit Montgomery limbs), used by circuit synthesis and
// by the handful of scalars create_proof / generate_parameters handle on the CPU.  See groth16.hpp.
#include "groth16.hpp"

#include <string.h>

namespace bellman {
namespace {
typedef unsigned __int128 u128;
const uint64_t FR_MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
const uint64_t FR_INV = 0xfffffffeffffffffULL;
const uint64_t FR_R[4] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL};
const uint64_t FR_R2[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};

inline bool geq_mod(const uint64_t *a) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] > FR_MOD[i]) return true;
    if (a[i] < FR_MOD[i]) return false;
  }
  return true;
}
inline void sub_mod(uint64_t *a) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a[i] - FR_MOD[i] - (uint64_t)br;
    a[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
// 4x64 CIOS Montgomery product, fully unrolled (synthesis is the serial part of create_proof)
__attribute__((always_inline)) inline void mont_mul(uint64_t *r, const uint64_t *a, const uint64_t *b) {
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#define BH_ROW(bi)                                                                        \
  {                                                                                       \
    u128 c = (u128)a[0] * (bi) + t0; t0 = (uint64_t)c; c >>= 64;                          \
    c += (u128)a[1] * (bi) + t1; t1 = (uint64_t)c; c >>= 64;                              \
    c += (u128)a[2] * (bi) + t2; t2 = (uint64_t)c; c >>= 64;                              \
    c += (u128)a[3] * (bi) + t3; t3 = (uint64_t)c; c >>= 64;                              \
    c += t4; t4 = (uint64_t)c; const uint64_t t5 = (uint64_t)(c >> 64);                   \
    const uint64_t m = t0 * FR_INV;                                                       \
    c = ((u128)m * FR_MOD[0] + t0) >> 64;                                                 \
    c += (u128)m * FR_MOD[1] + t1; t0 = (uint64_t)c; c >>= 64;                            \
    c += (u128)m * FR_MOD[2] + t2; t1 = (uint64_t)c; c >>= 64;                            \
    c += (u128)m * FR_MOD[3] + t3; t2 = (uint64_t)c; c >>= 64;                            \
    c += t4; t3 = (uint64_t)c; t4 = t5 + (uint64_t)(c >> 64);                             \
  }
  BH_ROW(b[0]) BH_ROW(b[1]) BH_ROW(b[2]) BH_ROW(b[3])
#undef BH_ROW
  uint64_t t[4] = {t0, t1, t2, t3};
  if (t4 || geq_mod(t)) sub_mod(t);
  r[0] = t[0]; r[1] = t[1]; r[2] = t[2]; r[3] = t[3];
}
}  // namespace

Fr Fr::zero() { Fr r; memset(r.l, 0, sizeof r.l); return r; }
Fr Fr::one() { Fr r; memcpy(r.l, FR_R, sizeof FR_R); return r; }
Fr Fr::from_u64(uint64_t v) {
  uint64_t c[4] = {v, 0, 0, 0};
  Fr r;
  mont_mul(r.l, c, FR_R2);
  return r;
}
Fr Fr::operator+(const Fr &o) const {
  Fr r;
  u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (u128)l[i] + o.l[i];
    r.l[i] = (uint64_t)c;
    c >>= 64;
  }
  if (geq_mod(r.l)) sub_mod(r.l);
  return r;
}
Fr Fr::operator-(const Fr &o) const {
  Fr r;
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)l[i] - o.l[i] - (uint64_t)br;
    r.l[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  if (br) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)r.l[i] + FR_MOD[i];
      r.l[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  return r;
}
Fr Fr::operator*(const Fr &o) const { Fr r; mont_mul(r.l, l, o.l); return r; }
Fr Fr::neg() const { return Fr::zero() - *this; }
void Fr::to_canonical(uint64_t out[4]) const {
  const uint64_t one[4] = {1, 0, 0, 0};
  mont_mul(out, l, one);
}
// little-endian 512-bit integer -> Fr (ff's wide reduction behind Field::random): lo + hi * 2^256 mod q
Fr Fr::from_u512(const uint64_t limbs[8]) {
  Fr lo, hi, r2;
  memcpy(lo.l, limbs, 32);
  memcpy(hi.l, limbs + 4, 32);
  memcpy(r2.l, FR_R2, 32);
  // mont_mul(x, R^2) = x * R mod q for any 256-bit x: the Montgomery form of x mod q
  const Fr lo_m = lo * r2, hi_m = hi * r2;
  return lo_m + hi_m * r2;   // Montgomery form of 2^256 is R * R = R^2
}
Fr Fr::pow_vartime(uint64_t e) const {
  Fr acc = Fr::one();
  for (int i = 63; i >= 0; i--) {
    acc = acc * acc;
    if ((e >> i) & 1) acc = acc * *this;
  }
  return acc;
}
Fr Fr::invert() const {   // a^(q-2)
  uint64_t e[4] = {FR_MOD[0] - 2, FR_MOD[1], FR_MOD[2], FR_MOD[3]};   // no borrow: the low limb ends in ...00000001 + 0xffffffff00000000
  Fr acc = Fr::one();
  for (int i = 255; i >= 0; i--) {
    acc = acc * acc;
    if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * *this;
  }
  return acc;
}
}  // namespace bellman
