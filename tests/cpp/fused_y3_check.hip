// Host check of the fused last line of the mixed addition (bellman_amd/csrc/ec.cuh: Y3 = R*(Q - X3) - Y1*PPP as ONE
// product pair under one reduction) and of the two-products-one-reduction multiplier under it (ff.cuh fe_mul2, also the
// lane-pair Fp2 product of fp2pair.cuh).  The curve and field code is __host__ __device__: this program compiles it for
// the host and compares
//   * fe_mul2 / FpOps::mul2_sub_tail with the separate products (random operands, operands with every 30-bit limb set),
//   * both overloads of xyzz_madd with xyzz_add (the general addition, which has no fused line) on chains of
//     additions that include the doubling and the inverse cases,
// as canonical affine coordinates.  Nothing here runs on a device (the device runs the same code in every G1 and large G2
// multiexp of the parity suite); built and run by tests/test_round3_cpu.py.
// Reference being restated by that code: src/multiexp.rs:39 (bucket += base).
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../bellman_amd/csrc/ec.cuh"

using namespace bh;

static uint64_t sm_state = 0x9E3779B97F4A7C15ull;
static uint64_t splitmix() {
  uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static fp_t random_below_2p() {
  for (;;) {
    fp_t r;
    for (int i = 0; i < 12; i += 2) { const uint64_t v = splitmix(); r.l[i] = (u32)v; r.l[i + 1] = (u32)(v >> 32); }
    r.l[11] &= 0x3fffffffu;   // < 2^382
    // accept if r < 2p
    bool lt = false;
    for (int i = 11; i >= 0; i--) {
      if (r.l[i] != fp_mod2(i)) { lt = r.l[i] < fp_mod2(i); break; }
    }
    if (lt) return r;
  }
}
static bool same(const fp_t &a, const fp_t &b) {
  fp_t x, y;
  fpl_canon(x, a);
  fpl_canon(y, b);
  return memcmp(&x, &y, sizeof x) == 0;
}
static bool below_2p(const fp_t &r) {
  for (int i = 11; i >= 0; i--)
    if (r.l[i] != fp_mod2(i)) return r.l[i] < fp_mod2(i);
  return false;
}

static const u32 GX[12] = {0xfd530c16u, 0x5cb38790u, 0x9976fff5u, 0x7817fc67u, 0x143ba1c1u, 0x154f95c7u,
                           0xf3d0e747u, 0xf0ae6acdu, 0x21dbf440u, 0xedce6eccu, 0x9e0bfb75u, 0x12017741u};
static const u32 GY[12] = {0x0ce72271u, 0xbaac93d5u, 0x7918fd8eu, 0x8c22631au, 0x570725ceu, 0xdd595f13u,
                           0x50405194u, 0x51ac5829u, 0xad0059c0u, 0x0e1c8c3fu, 0x5008a26au, 0x0bbc3efcu};

static Affine<FpOps> to_affine(const XYZZ<FpOps> &p) {
  Affine<FpOps> a;
  if (xyzz_is_identity(p)) { fe_zero(a.x); fe_zero(a.y); return a; }
  fp_t izz, izzz;
  FpOps::inv(izz, p.zz);
  FpOps::inv(izzz, p.zzz);
  FpOps::mul(a.x, p.x, izz);
  FpOps::mul(a.y, p.y, izzz);
  fpl_canon(a.x, a.x);
  fpl_canon(a.y, a.y);
  return a;
}
static bool same_point(const XYZZ<FpOps> &a, const XYZZ<FpOps> &b) {
  const Affine<FpOps> x = to_affine(a), y = to_affine(b);
  return memcmp(&x, &y, sizeof x) == 0;
}

// ---- G2: the same two checks over Fp2 (prefetching overload of xyzz_madd only: the one-lane accumulate kernel) ------
static const u32 G2W[48] = {
    0x02940a10u, 0xf5f28fa2u, 0x87b4961au, 0xb3f5fb26u, 0x3e2ae580u, 0xa1a893b5u, 0x1a3caee9u, 0x9894999du, 0x1863366bu, 0x6f67b763u, 0x4350bcd7u, 0x05819192u,
    0x9e23f606u, 0xa5a9c075u, 0xbccd60c3u, 0xaaa0c59du, 0xe2867806u, 0x3bb17e18u, 0x8541b367u, 0x1b1ab6ccu, 0xf2158547u, 0xc2b6ed0eu, 0x7360edf3u, 0x11922a09u,
    0x60494c4au, 0x4c730af8u, 0x5e369c5au, 0x597cfa1fu, 0xaa0a635au, 0xe7e6856cu, 0x6e0d495fu, 0xbbefb5e9u, 0xf0ef25a2u, 0x07d3a975u, 0x7e80dae5u, 0x0083fd8eu,
    0xdf64b05du, 0xadc0fc92u, 0x2b1461dcu, 0x18aa270au, 0x3be4eba0u, 0x86adac6au, 0xc93da33au, 0x79495c4eu, 0xa43ccaedu, 0xe7175850u, 0x63de1bf2u, 0x0b2bc2a1u};
static Affine<Fp2Ops> to_affine2(const XYZZ<Fp2Ops> &p) {
  Affine<Fp2Ops> a;
  if (xyzz_is_identity(p)) { Fp2Ops::zero(a.x); Fp2Ops::zero(a.y); return a; }
  fp2_t izz, izzz;
  Fp2Ops::inv(izz, p.zz);
  Fp2Ops::inv(izzz, p.zzz);
  Fp2Ops::mul(a.x, p.x, izz);
  Fp2Ops::mul(a.y, p.y, izzz);
  Fp2Ops::canon(a.x);
  Fp2Ops::canon(a.y);
  return a;
}
static bool same_point2(const XYZZ<Fp2Ops> &a, const XYZZ<Fp2Ops> &b) {
  const Affine<Fp2Ops> x = to_affine2(a), y = to_affine2(b);
  return memcmp(&x, &y, sizeof x) == 0;
}
static int check_g2() {
  static_assert(Fp2Ops::FUSED_Y3_TAIL, "the one-lane G2 accumulation takes the fused line");
  int bad = 0;
  for (int it = 0; it < 2000; it++) {
    fp2_t a, b, c, d, ab, cd, dif, g;
    a.c0 = random_below_2p(); a.c1 = random_below_2p(); b.c0 = random_below_2p(); b.c1 = random_below_2p();
    c.c0 = random_below_2p(); c.c1 = random_below_2p(); d.c0 = random_below_2p(); d.c1 = random_below_2p();
    Fp2Ops::mul(ab, a, b);
    Fp2Ops::mul(cd, c, d);
    Fp2Ops::sub(dif, ab, cd);
    Fp2Ops::mul2_sub_tail(g, a, b, c, d);
    if (!same(g.c0, dif.c0) || !same(g.c1, dif.c1) || !below_2p(g.c0) || !below_2p(g.c1)) {
      if (bad++ < 5) printf("Fp2 mul2_sub_tail mismatch at %d\n", it);
    }
  }
  Affine<Fp2Ops> gen;
  memcpy(&gen, G2W, sizeof gen);
  constexpr int NP = 24;
  Affine<Fp2Ops> pts[NP];
  {
    XYZZ<Fp2Ops> g1, acc;
    xyzz_from_affine(g1, gen);
    acc = g1;
    for (int i = 0; i < NP; i++) {
      pts[i] = to_affine2(acc);
      XYZZ<Fp2Ops> t;
      for (int r = 0; r < 1 + (i % 3); r++) { xyzz_add(t, acc, g1); acc = t; }
      if (i % 5 == 3) { xyzz_dbl(t, acc); acc = t; }
    }
  }
  auto nop = [] {};
  XYZZ<Fp2Ops> acc, ref;
  xyzz_set_identity(acc);
  xyzz_set_identity(ref);
  for (int round = 0; round < 2; round++) {
    for (int i = 0; i < NP; i++) {
      Affine<Fp2Ops> q = pts[(i * 5 + round) % NP];
      if ((i + round) % 4 == 1) { Fp2Ops::neg(q.y, q.y); Fp2Ops::canon(q.y); }
      XYZZ<Fp2Ops> qx, t;
      xyzz_from_affine(qx, q);
      xyzz_add(t, ref, qx);
      ref = t;
      xyzz_madd(acc, q, nop);
      if (!same_point2(acc, ref)) { if (bad++ < 5) printf("G2 madd mismatch round %d i %d\n", round, i); }
    }
  }
  Affine<Fp2Ops> same_pt = to_affine2(acc);
  XYZZ<Fp2Ops> d2 = acc, r2;
  xyzz_madd(d2, same_pt, nop);
  xyzz_dbl(r2, acc);
  if (!same_point2(d2, r2)) { bad++; printf("G2 doubling path\n"); }
  Affine<Fp2Ops> neg_pt = same_pt;
  Fp2Ops::neg(neg_pt.y, neg_pt.y);
  XYZZ<Fp2Ops> z = acc;
  xyzz_madd(z, neg_pt, nop);
  if (!xyzz_is_identity(z)) { bad++; printf("G2 inverse path\n"); }
  return bad;
}

int main() {
  static_assert(FpOps::FUSED_Y3_TAIL && FpOps::FUSED_Y3, "G1 takes the fused line in both overloads");
  int bad = check_g2();
  // ---- 1. the multiplier ----------------------------------------------------------------------------------------------
  fp_t allones, top;   // every 30-bit limb of the operand (and of its pre-shifted form) set
  for (int i = 0; i < 12; i++) allones.l[i] = 0xffffffffu;
  top = allones;
  top.l[11] = 0x1fffffffu;   // < 2^381: the result of four such operands still fits 384 bits
  for (int it = 0; it < 4000; it++) {
    fp_t a = random_below_2p(), b = random_below_2p(), c = random_below_2p(), d = random_below_2p();
    if (it < 16) {   // maximal limbs in every combination of the four operands
      if (it & 1) a = top;
      if (it & 2) b = top;
      if (it & 4) c = top;
      if (it & 8) d = top;
    }
    fp_t ab, cd, sum, dif, f, g;
    fe_mul<FpParams, false>(ab, a, b);
    fe_mul<FpParams, false>(cd, c, d);
    fpl_canon(ab, ab);
    fpl_canon(cd, cd);
    fpl_add(sum, ab, cd);
    fe_mul2<FpParams, false>(f, a, b, c, d);
    if (!same(f, sum)) { if (bad++ < 5) printf("fe_mul2 mismatch at %d\n", it); }
    if (it >= 16) {
      fpl_sub(dif, ab, cd);
      FpOps::mul2_sub_tail(g, a, b, c, d);
      if (!same(g, dif) || !below_2p(g)) { if (bad++ < 5) printf("mul2_sub_tail mismatch at %d\n", it); }
    }
  }
  // ---- 2. the group law -----------------------------------------------------------------------------------------------
  Affine<FpOps> gen;
  for (int i = 0; i < 12; i++) { gen.x.l[i] = GX[i]; gen.y.l[i] = GY[i]; }
  constexpr int NP = 40;
  Affine<FpOps> pts[NP];   // [k+1]G for a few irregular k
  {
    XYZZ<FpOps> g1, acc;
    xyzz_from_affine(g1, gen);
    acc = g1;
    for (int i = 0; i < NP; i++) {
      pts[i] = to_affine(acc);
      XYZZ<FpOps> t;
      for (int r = 0; r < 1 + (i % 3); r++) { xyzz_add(t, acc, g1); acc = t; }
      if (i % 7 == 3) { xyzz_dbl(t, acc); acc = t; }
    }
  }
  auto nop = [] {};
  for (int mode = 0; mode < 2; mode++) {
    XYZZ<FpOps> acc, ref;
    xyzz_set_identity(acc);
    xyzz_set_identity(ref);
    for (int round = 0; round < 3; round++) {
      for (int i = 0; i < NP; i++) {
        Affine<FpOps> q = pts[(i * 7 + round) % NP];
        if ((i + round) % 5 == 2) FpOps::neg(q.y, q.y);   // negative digits add -P
        fpl_canon(q.y, q.y);
        XYZZ<FpOps> qx, t;
        xyzz_from_affine(qx, q);
        xyzz_add(t, ref, qx);
        ref = t;
        if (mode == 0) xyzz_madd(acc, q); else xyzz_madd(acc, q, nop);
        if (!same_point(acc, ref)) { if (bad++ < 5) printf("madd mismatch mode %d round %d i %d\n", mode, round, i); }
        // the accumulator's coordinates stay in the lazily reduced range the multiplier is fed with
        if (!below_2p(acc.x) || !below_2p(acc.y) || !below_2p(acc.zz) || !below_2p(acc.zzz)) {
          if (bad++ < 5) printf("range mode %d round %d i %d\n", mode, round, i);
        }
      }
    }
    // acc + (the same point) -> doubling path;  acc + (-acc) -> identity
    Affine<FpOps> same_pt = to_affine(acc);
    XYZZ<FpOps> d2 = acc, r2;
    if (mode == 0) xyzz_madd(d2, same_pt); else xyzz_madd(d2, same_pt, nop);
    xyzz_dbl(r2, acc);
    if (!same_point(d2, r2)) { bad++; printf("doubling path mode %d\n", mode); }
    Affine<FpOps> neg_pt = same_pt;
    FpOps::neg(neg_pt.y, neg_pt.y);
    XYZZ<FpOps> z = acc;
    if (mode == 0) xyzz_madd(z, neg_pt); else xyzz_madd(z, neg_pt, nop);
    if (!xyzz_is_identity(z)) { bad++; printf("inverse path mode %d\n", mode); }
  }
  printf(bad ? "FAILED %d\n" : "fused Y3: ok\n", bad);
  return bad ? 1 : 0;
}
