// GLV decomposition of BLS12-381 scalars for the G1 multiexp (host + device).
//
// G1 has the endomorphism phi(x, y) = (beta * x, y) = [lambda] (x, y) with lambda = z^2 - 1 (z = -0xd201000000010000 the
// curve parameter) a primitive cube root of unity mod r: r = lambda^2 + lambda + 1 EXACTLY.  A scalar s < r is written
//     s = k1 + k2 * lambda   (mod r),   |k1|, |k2| < 2^127
// so that  sum s_i P_i = sum k1_i P_i + sum k2_i phi(P_i): twice the terms with half-length scalars, i.e. the same number
// of bucket additions over HALF the windows - half the bucket reductions and half the host tail of a classic-plan
// multiexp (src/multiexp.rs:271-275, :295-300 are what those stages replace).  The result is the same group element.
//
//   q = floor(s / lambda), t = s - q * lambda          (q <= lambda + 1 because r - 1 = lambda (lambda + 1))
//   k1 = t, k2 = q
//   k1 > lambda / 2        ->  k1 -= lambda,       k2 += 1
//   k2 > (lambda + 1) / 2  ->  k2 -= (lambda + 1), k1 -= 1      ((lambda + 1) lambda = r - 1 = -1 mod r)
// leaves |k1| <= lambda / 2 + 1, |k2| <= (lambda + 1) / 2 + 1 < 0.68 * 2^127: the top 16-bit digit of a magnitude is below
// 2^15, so the signed-digit recoding into eight 16-bit windows never carries out of the top window.
// Pinned against Python integers on the host (tests/test_abi_cpu.py::test_glv_decomposition_host).
#pragma once
#include "ff.cuh"

namespace bh {

struct GlvHalf {
  u64 lo, hi;   // magnitude < 2^127
  bool neg;
};

namespace glv {
constexpr u64 LAM0 = 0x00000000ffffffffull, LAM1 = 0xac45a4010001a402ull;                                  // lambda
constexpr u64 MU0 = 0x63f6e522f6cfee30ull, MU1 = 0x7c6becf1e01faaddull;                                    // floor(2^256 / lambda) = 2^128 + MU1:MU0
constexpr u64 HALF0 = 0x000000007fffffffull, HALF1 = 0x5622d2008000d201ull;                                // lambda >> 1
constexpr u64 HALFB0 = 0x0000000080000000ull, HALFB1 = 0x5622d2008000d201ull;                              // (lambda + 1) >> 1
constexpr u64 LAMP0 = 0x0000000100000000ull, LAMP1 = 0xac45a4010001a402ull;                                // lambda + 1
// beta in Montgomery form (2^384 beta mod p): phi(x, y) = (beta x, y) for THIS lambda ([lambda] G == (beta G.x, G.y) checked with
// Python integers when the constant was derived; the other cube root belongs to lambda^2)
BH_HD constexpr u32 beta_mont(int i) {
  constexpr u32 m[12] = {0x8671f071u, 0xcd03c9e4u, 0x1fcda5d2u, 0x5dab2246u, 0xd3851b95u, 0x587042afu,
                         0x01bacb9eu, 0x8eb60ebeu, 0x83d050d2u, 0x03f97d6eu, 0x54638741u, 0x18f02065u};
  return m[i];
}
BH_HD u64 mulhi(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}
// r += a (one limb) at position i of a little-endian limb array of length n, with carry propagation
template <int N>
BH_HD void add_at(u64 (&r)[N], int i, u64 a) {
#pragma unroll
  for (int k = 0; k < N; k++) {
    if (k < i) continue;
    const u64 s = r[k] + a;
    a = s < a ? 1 : 0;
    r[k] = s;
  }
}
BH_HD bool ge2(u64 a0, u64 a1, u64 b0, u64 b1) { return a1 > b1 || (a1 == b1 && a0 >= b0); }   // a >= b
BH_HD bool gt2(u64 a0, u64 a1, u64 b0, u64 b1) { return a1 > b1 || (a1 == b1 && a0 > b0); }    // a > b
}  // namespace glv

// s: canonical, < r (8 x 32-bit little-endian limbs)
BH_HD void glv_decompose(const fr_t &s, GlvHalf &k1, GlvHalf &k2) {
  using namespace glv;
  const u64 sl[4] = {(u64)s.l[0] | ((u64)s.l[1] << 32), (u64)s.l[2] | ((u64)s.l[3] << 32), (u64)s.l[4] | ((u64)s.l[5] << 32),
                     (u64)s.l[6] | ((u64)s.l[7] << 32)};
  // q_est = floor(s * mu / 2^256), mu = 2^128 + MU1:MU0  (exact product: q_est <= q <= q_est + 2)
  u64 prod[7] = {0, 0, 0, 0, 0, 0, 0};
  const u64 mul[2] = {MU0, MU1};
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      add_at(prod, i + j, sl[i] * mul[j]);
      add_at(prod, i + j + 1, mulhi(sl[i], mul[j]));
    }
    add_at(prod, i + 2, sl[i]);   // the 2^128 term of mu
  }
  u64 q[3] = {prod[4], prod[5], prod[6]};
  // t = s - q * lambda (mod 2^192; the true value is below 3 lambda < 2^130)
  u64 ql[3] = {0, 0, 0};
  const u64 lam[2] = {LAM0, LAM1};
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      if (i + j < 3) add_at(ql, i + j, q[i] * lam[j]);
      if (i + j + 1 < 3) add_at(ql, i + j + 1, mulhi(q[i], lam[j]));
    }
  }
  u64 t[3];
  {
    u64 br = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const u64 a = sl[i], b = ql[i];
      const u64 d = a - b - br;
      br = (a < b || (a == b && br)) ? 1 : 0;
      t[i] = d;
    }
  }
#pragma unroll
  for (int it = 0; it < 2; it++) {   // at most two corrections
    if (t[2] != 0 || ge2(t[0], t[1], LAM0, LAM1)) {
      const u64 b0 = t[0] < LAM0 ? 1 : 0;
      t[0] -= LAM0;
      const u64 b1 = (t[1] < LAM1 || (t[1] == LAM1 && b0)) ? 1 : 0;
      t[1] = t[1] - LAM1 - b0;
      t[2] -= b1;
      add_at(q, 0, 1);
    }
  }
  // balance: k1 in (-lambda/2, lambda/2], then k2 likewise against lambda + 1
  k1.neg = false;
  k1.lo = t[0]; k1.hi = t[1];
  if (gt2(k1.lo, k1.hi, HALF0, HALF1)) {      // k1 = -(lambda - t)
    const u64 b0 = LAM0 < k1.lo ? 1 : 0;
    k1.lo = LAM0 - k1.lo;
    k1.hi = LAM1 - k1.hi - b0;
    k1.neg = true;
    add_at(q, 0, 1);
  }
  k2.neg = false;
  k2.lo = q[0]; k2.hi = q[1];                  // q <= lambda + 2 < 2^128: q[2] == 0
  if (gt2(k2.lo, k2.hi, HALFB0, HALFB1)) {     // k2 = -((lambda + 1) - q), k1 -= 1
    const u64 b0 = LAMP0 < k2.lo ? 1 : 0;
    k2.lo = LAMP0 - k2.lo;
    k2.hi = LAMP1 - k2.hi - b0;
    k2.neg = true;
    if (k1.neg) {                              // -(m) - 1 = -(m + 1)
      k1.lo += 1;
      if (k1.lo == 0) k1.hi += 1;
    } else if ((k1.lo | k1.hi) == 0) {         // 0 - 1
      k1.lo = 1;
      k1.neg = true;
    } else {                                   // m - 1
      if (k1.lo == 0) k1.hi -= 1;
      k1.lo -= 1;
    }
  }
  if ((k1.lo | k1.hi) == 0) k1.neg = false;
  if ((k2.lo | k2.hi) == 0) k2.neg = false;
}

}  // namespace bh
