// BLS12-381 G1 / G2 group law for the bucket MSM, written once over a field-ops bundle
// (FpOps -> G1, Fp2Ops -> G2).
//
// Replaces (through `group` traits) what bellman's multiexp calls per base:
//   src/multiexp.rs:39       bucket += affine base      -> xyzz_madd
//   src/multiexp.rs:273-274  running_sum += bucket ...  -> xyzz_add
//   src/multiexp.rs:299      acc.double()               -> xyzz_dbl
// The reference (bls12_381 0.8.0) uses homogeneous projective coordinates; bellman only ever
// consumes the result as a GROUP ELEMENT (multiexp.rs:377, prover.rs:356-360), so the
// representation is free.  XYZZ (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2) has the cheapest mixed
// addition for short-Weierstrass a = 0 curves: 8M + 2S vs 11M for the complete formulas.
// Identity: ZZ == 0 (an all-zero record is the identity, so hipMemset(0) clears buckets).
#pragma once
#include "ff.cuh"

namespace bh {

template <class F>
struct alignas(16) Affine {  // identity encoded as (0, 0): not on the curve, unambiguous
  typename F::T x, y;
};
template <class F>
struct alignas(16) XYZZ {
  typename F::T x, y, zz, zzz;
};

// records in memory (Affine<F::Mem> / XYZZ<F::Mem>) <-> what a lane computes with (identical unless F spreads
// an element over several lanes, fp2k3.cuh)
template <class F>
BH_HD void load_affine(Affine<F> &q, const Affine<typename F::Mem> *p) {
  F::load(q.x, &p->x);
  F::load(q.y, &p->y);
}
template <class F>
BH_HD void load_xyzz(XYZZ<F> &q, const XYZZ<typename F::Mem> *p) {
  F::load(q.x, &p->x);
  F::load(q.y, &p->y);
  F::load(q.zz, &p->zz);
  F::load(q.zzz, &p->zzz);
}
template <class F>
BH_HD void store_xyzz(XYZZ<typename F::Mem> *p, const XYZZ<F> &q) {
  F::store(&p->x, q.x);
  F::store(&p->y, q.y);
  F::store(&p->zz, q.zz);
  F::store(&p->zzz, q.zzz);
}

// affine records are CANONICAL wherever they come from (memory, xyzz_to_affine): the identity is the all-zero record, and
// testing for it needs none of the "0 or p" comparisons of the lazily reduced is_zero (48 instructions fewer per base of
// the bucket accumulation)
template <class F>
BH_HD bool aff_is_identity(const Affine<F> &p) {
  return F::is_zero_canonical(p.x, p.y);
}
template <class F>
BH_HD void xyzz_set_identity(XYZZ<F> &p) {
  F::zero(p.x);
  F::zero(p.y);
  F::zero(p.zz);
  F::zero(p.zzz);
}
template <class F>
BH_HD bool xyzz_is_identity(const XYZZ<F> &p) {
  return F::is_zero(p.zz);
}
template <class F>
BH_HD void xyzz_from_affine(XYZZ<F> &r, const Affine<F> &p) {
  if (aff_is_identity(p)) {
    xyzz_set_identity(r);
    return;
  }
  r.x = p.x;
  r.y = p.y;
  F::one(r.zz);
  F::one(r.zzz);
}

// dbl-2008-s-1 (a = 0): 2 [X:Y:ZZ:ZZZ]
template <class F>
BH_HD void xyzz_dbl(XYZZ<F> &r, const XYZZ<F> &p) {
  typedef typename F::T T;
  if (xyzz_is_identity(p) || F::is_zero(p.y)) {  // (y == 0 cannot occur in a prime-order subgroup)
    xyzz_set_identity(r);
    return;
  }
  T u, v, w, s, m, t, x3, y3;
  F::dbl(u, p.y);        // U = 2*Y1
  F::sqr(v, u);          // V = U^2
  F::mul(w, u, v);       // W = U*V
  F::mul(s, p.x, v);     // S = X1*V
  F::sqr(t, p.x);
  F::dbl(m, t);
  F::add(m, m, t);       // M = 3*X1^2
  F::sqr(x3, m);
  F::sub(x3, x3, s);
  F::sub(x3, x3, s);     // X3 = M^2 - 2S
  F::sub(t, s, x3);
  F::mul(y3, m, t);
  F::mul(t, w, p.y);
  F::sub(y3, y3, t);     // Y3 = M*(S-X3) - W*Y1
  F::mul(r.zz, v, p.zz);
  F::mul(r.zzz, w, p.zzz);
  r.x = x3;
  r.y = y3;
}

// affine doubling straight into XYZZ (mdbl-2008-s-1): used when acc == base in xyzz_madd
template <class F>
BH_HD void xyzz_dbl_affine(XYZZ<F> &r, const Affine<F> &p) {
  typedef typename F::T T;
  T u, s, m, t;
  F::dbl(u, p.y);          // U = 2*Y1
  F::sqr(r.zz, u);         // V = ZZ3
  F::mul(r.zzz, u, r.zz);  // W = ZZZ3
  F::mul(s, p.x, r.zz);    // S
  F::sqr(t, p.x);
  F::dbl(m, t);
  F::add(m, m, t);         // M = 3 X1^2
  F::sqr(r.x, m);
  F::sub(r.x, r.x, s);
  F::sub(r.x, r.x, s);
  F::sub(t, s, r.x);
  F::mul(t, m, t);
  F::mul(u, r.zzz, p.y);
  F::sub(r.y, t, u);
}

// madd-2008-s: acc += affine q  (q must not be the identity; callers check).
// `prefetch` is called exactly once on every path, at the point after which no out-of-line call follows: every
// non-kernel function starts with s_waitcnt vmcnt(0) (the AMDGPU calling convention cannot track the caller's
// loads), so a load issued before ANY product call is waited for at that call - software prefetching across
// calls is impossible.  The last product of the formula is therefore an INLINE copy of the multiplier
// (F::mul_tail); loads issued by `prefetch` right before it have that whole product (1.2 us G1, 3.5 us G2) to land.
// Pins the order "v is computed -> what follows": the products are pure functions, so without it the compiler sinks
// the calls before `prefetch` below it (and hoists the loads above them).
template <class T>
BH_HD void order_after(T &v) {
#ifdef __HIP_DEVICE_COMPILE__
  u32 *w = reinterpret_cast<u32 *>(&v);
  __asm__ volatile("" : "+v"(w[0]), "+v"(w[sizeof(T) / 4 - 1]) : : "memory");   // a word of every Fp component
#else
  (void)v;
#endif
}
// Returns false when acc was the identity (the "addition" is a copy), true when a group addition was executed.
template <class F, class PF>
BH_HD bool xyzz_madd(XYZZ<F> &acc, const Affine<F> &q, PF prefetch) {
  typedef typename F::T T;
  if (xyzz_is_identity(acc)) {
    acc.x = q.x;
    acc.y = q.y;
    F::one(acc.zz);
    F::one(acc.zzz);
    prefetch();
    return false;
  }
  T p, r, pp, ppp, qq, t;
  F::mul(p, q.x, acc.zz);
  F::sub(p, p, acc.x);     // P = U2 - X1
  F::mul(r, q.y, acc.zzz);
  F::sub(r, r, acc.y);     // R = S2 - Y1
  if (F::is_zero(p)) {
    if (F::is_zero(r)) {
      xyzz_dbl_affine(acc, q);  // same point
    } else {
      xyzz_set_identity(acc);   // opposite points
    }
    prefetch();
    return true;
  }
  // ordered so that at most four temporaries are live at once (PP and PPP die early): the G2
  // instantiation is register-bound
  F::sqr(pp, p);
  F::mul(ppp, p, pp);                 // p dead
  F::mul(qq, acc.x, pp);              // Q = X1*PP
  F::mul(acc.zz, acc.zz, pp);         // ZZ3 = ZZ1*PP          (pp dead)
  F::mul(acc.zzz, acc.zzz, ppp);      // ZZZ3 = ZZZ1*PPP
  F::sqr(t, r);
  F::sub(t, t, ppp);
  F::sub(t, t, qq);
  F::sub(t, t, qq);                   // X3 = R^2 - PPP - 2Q
  if constexpr (F::FUSED_Y3_TAIL) {
    // Y3 = R*(Q - X3) - Y1*PPP as two products under ONE reduction (ff.cuh fe_mul2: 169 mads, one out-of-line call and
    // one subtraction fewer per Fp-level product pair; profiles/archive/r4_call1_fused_y3.txt: G1 accumulate -4.3 %, G2 -6 %)
    F::sub(qq, qq, t);
    acc.x = t;
    order_after(qq);
    prefetch();
    F::mul2_sub_tail(acc.y, r, qq, acc.y, ppp);
  } else {
    F::mul(ppp, acc.y, ppp);            // Y1*PPP                (reuses ppp)
    F::sub(qq, qq, t);
    acc.x = t;
    order_after(ppp);
    prefetch();
    F::mul_tail(qq, r, qq);             // R*(Q - X3)            (r dead); inline, see above
    F::sub(acc.y, qq, ppp);
  }
  return true;
}
template <class F>
BH_HD bool xyzz_madd(XYZZ<F> &acc, const Affine<F> &q) {
  typedef typename F::T T;
  if (xyzz_is_identity(acc)) {
    acc.x = q.x;
    acc.y = q.y;
    F::one(acc.zz);
    F::one(acc.zzz);
    return false;
  }
  T p, r, pp, ppp, qq, t;
  F::mul(p, q.x, acc.zz);
  F::sub(p, p, acc.x);     // P = U2 - X1
  F::mul(r, q.y, acc.zzz);
  F::sub(r, r, acc.y);     // R = S2 - Y1
  if (F::is_zero(p)) {
    if (F::is_zero(r)) {
      xyzz_dbl_affine(acc, q);  // same point
    } else {
      xyzz_set_identity(acc);   // opposite points
    }
    return true;
  }
  F::sqr(pp, p);
  F::mul(ppp, p, pp);                 // p dead
  F::mul(qq, acc.x, pp);              // Q = X1*PP
  F::mul(acc.zz, acc.zz, pp);         // ZZ3 = ZZ1*PP          (pp dead)
  F::mul(acc.zzz, acc.zzz, ppp);      // ZZZ3 = ZZZ1*PPP
  F::sqr(t, r);
  F::sub(t, t, ppp);
  F::sub(t, t, qq);
  F::sub(t, t, qq);                   // X3 = R^2 - PPP - 2Q
  if constexpr (F::FUSED_Y3) {   // see the overload above
    F::sub(qq, qq, t);
    F::mul2_sub(acc.y, r, qq, acc.y, ppp);
  } else {
    F::mul(ppp, acc.y, ppp);            // Y1*PPP                (reuses ppp)
    F::sub(qq, qq, t);
    F::mul(qq, r, qq);                  // R*(Q - X3)            (r dead)
    F::sub(acc.y, qq, ppp);
  }
  acc.x = t;
  return true;
}

// add-2008-s: r = a + b (general)
template <class F>
BH_HD void xyzz_add(XYZZ<F> &r, const XYZZ<F> &a, const XYZZ<F> &b) {
  typedef typename F::T T;
  if (xyzz_is_identity(a)) {
    r = b;
    return;
  }
  if (xyzz_is_identity(b)) {
    r = a;
    return;
  }
  T u1, u2, s1, s2, p, rr, pp, ppp, q, t;
  F::mul(u1, a.x, b.zz);
  F::mul(u2, b.x, a.zz);
  F::mul(s1, a.y, b.zzz);
  F::mul(s2, b.y, a.zzz);
  F::sub(p, u2, u1);
  F::sub(rr, s2, s1);
  if (F::is_zero(p)) {
    if (F::is_zero(rr)) {
      xyzz_dbl(r, a);
    } else {
      xyzz_set_identity(r);
    }
    return;
  }
  // (r may alias a: every read of a.x / a.y / b.x / b.y happened above; zz / zzz are read here)
  F::sqr(pp, p);
  F::mul(ppp, p, pp);                 // p dead
  F::mul(q, u1, pp);                  // u1 dead
  F::mul(t, a.zz, b.zz);
  F::mul(r.zz, t, pp);                // pp dead
  F::mul(t, a.zzz, b.zzz);
  F::mul(r.zzz, t, ppp);
  F::sqr(t, rr);
  F::sub(t, t, ppp);
  F::sub(t, t, q);
  F::sub(t, t, q);                    // X3
  F::mul(s1, s1, ppp);                // ppp dead
  F::sub(q, q, t);
  F::mul(q, rr, q);
  F::sub(r.y, q, s1);
  r.x = t;
}

// XYZZ -> affine (one field inversion): x = X/ZZ, y = Y/ZZZ
template <class F>
BH_HD void xyzz_to_affine(Affine<F> &r, const XYZZ<F> &p) {
  typedef typename F::T T;
  if (xyzz_is_identity(p)) {
    F::zero(r.x);
    F::zero(r.y);
    return;
  }
  T zi, zi2, zi3;                // 1/ZZZ, then 1/ZZ = ZZ^2/ZZZ^2 * ... use: ZZ^3 = ZZZ^2
  F::inv(zi, p.zzz);             // 1/ZZZ
  F::mul(r.y, p.y, zi);
  F::mul(zi2, zi, p.zz);         // ZZ/ZZZ = 1/Z
  F::sqr(zi3, zi2);              // 1/Z^2 = 1/ZZ
  F::mul(r.x, p.x, zi3);
  (void)zi3;
  F::canon(r.x);   // the curve code computes with lazily reduced values (ff.cuh): affine records are canonical
  F::canon(r.y);
}

}  // namespace bh
