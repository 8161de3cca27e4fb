// Fp2 spread over lane TRIPLES ("K3" form) for the G2 kernels.
//
// An element a = a0 + a1*u of Fp2 = Fp[u]/(u^2+1) is held by three neighbouring lanes of a wavefront as
// (a0, a1, a0+a1).  Karatsuba's three Fp products of an Fp2 product are then ONE lane-local product per lane,
//     t0 = a0*b0,  t1 = a1*b1,  t2 = (a0+a1)*(b0+b1),
// followed by a neighbour exchange (two DPP wave shifts) and two subtractions to get back to the same form:
//     c0 = t0 - t1,   c1 = t2 - t0 - t1,   c0 + c1 = t2 - 2*t1.
// Addition, subtraction, negation and doubling are lane-local.  Squaring is the same code with t_i = a_i^2.
//
// Why (MI355X): a G2 point in one lane needs 96 VGPRs for the running XYZZ sum alone and 28 out-of-line Fp
// products per mixed addition - the single-lane kernels sit at 256 VGPR + AGPR spills, ONE wavefront per SIMD,
// where every non-multiplier instruction costs a 4-cycle issue slot.  In K3 form a lane carries exactly the
// state of a G1 lane (48 words per XYZZ point), so the G2 kernels run at the G1 kernels' two wavefronts per
// SIMD, and a G2 point addition has the LATENCY of a G1 addition (+ the exchange) instead of 3-4x of it -
// which is what the latency-bound bucket reductions are made of.  63 of 64 lanes work (21 triples).
//
// Replaces what bellman gets from bls12_381's `Fp2` through the `group` traits on the G2 multiexp
// (src/multiexp.rs:39,273-274 with G = G2Projective); values stay lazily reduced in [0, 2p) like FpOps.
#pragma once
#include "ec.cuh"

namespace bh {

__device__ __forceinline__ u32 k3_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// lane / 3 and lane % 3 for lane < 64 (43/128 rounds 1/3 up just enough)
__device__ __forceinline__ u32 k3_triple(u32 lane) { return (lane * 43u) >> 7; }
__device__ __forceinline__ u32 k3_role() {
  const u32 l = k3_lane();
  return l - 3u * k3_triple(l);
}

// value held by the lane below (wave_shr:1) / above (wave_shl:1); 0 where that lane does not exist or is
// inactive (bound_ctrl off, old = 0).  Lanes of one triple always run together.
template <int CTRL>
__device__ __forceinline__ fp_t k3_shift(const fp_t &v) {
  fp_t r;
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)v.l[i], CTRL, 0xf, 0xf, false);
  return r;
}
__device__ __forceinline__ fp_t k3_from_below(const fp_t &v) { return k3_shift<0x138>(v); }   // lane i <- lane i-1
__device__ __forceinline__ fp_t k3_from_above(const fp_t &v) { return k3_shift<0x130>(v); }   // lane i <- lane i+1

// true iff the c0 lane AND the c1 lane of my triple say true (the sum lane's opinion is redundant)
__device__ __forceinline__ bool k3_all(bool mine) {
  const u32 lane = k3_lane(), role = lane - 3u * k3_triple(lane);
  const u64 m = __ballot(mine || role == 2);
  return ((m >> (lane - role)) & 7u) == 7u;
}

// (t0, t1, t2) -> (t0 - t1, t2 - t0 - t1, t2 - 2 t1), written as X - Y - Z with lane-dependent operands:
//   c0 lane: own - above - 0      c1 lane: above - below - own      sum lane: own - below - below
__device__ __forceinline__ fp_t k3_recombine(const fp_t &t) {
  const u32 role = k3_role();
  const fp_t below = k3_from_below(t), above = k3_from_above(t);
  fp_t x, y, z, r;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    x.l[i] = (role == 1) ? above.l[i] : t.l[i];
    y.l[i] = (role == 0) ? above.l[i] : below.l[i];
    z.l[i] = (role == 0) ? 0u : (role == 1) ? t.l[i] : below.l[i];
  }
  fpl_sub(x, x, y);
  fpl_sub(r, x, z);
  return r;
}
// Out-of-line LEAF functions (product + exchange + recombination in one body; operands and result in VGPRs like
// fp_mul_vec): inlined into the point formulas the exchange temporaries pushed the G2 kernels from the G1
// kernels' 222 registers to 370 and back to one wavefront per SIMD.
__device__ __attribute__((noinline)) static fp_t k3_mul_vec(u32x4 a0, u32x4 a1, u32x4 a2, u32x4 b0, u32x4 b1, u32x4 b2) {
  fp_t a, b, t;
  a.l[0] = a0.x; a.l[1] = a0.y; a.l[2] = a0.z; a.l[3] = a0.w;
  a.l[4] = a1.x; a.l[5] = a1.y; a.l[6] = a1.z; a.l[7] = a1.w;
  a.l[8] = a2.x; a.l[9] = a2.y; a.l[10] = a2.z; a.l[11] = a2.w;
  b.l[0] = b0.x; b.l[1] = b0.y; b.l[2] = b0.z; b.l[3] = b0.w;
  b.l[4] = b1.x; b.l[5] = b1.y; b.l[6] = b1.z; b.l[7] = b1.w;
  b.l[8] = b2.x; b.l[9] = b2.y; b.l[10] = b2.z; b.l[11] = b2.w;
  fe_mul<FpParams, false>(t, a, b);
  return k3_recombine(t);
}
__device__ __attribute__((noinline)) static fp_t k3_sqr_vec(u32x4 a0, u32x4 a1, u32x4 a2) {
  fp_t a, t;
  a.l[0] = a0.x; a.l[1] = a0.y; a.l[2] = a0.z; a.l[3] = a0.w;
  a.l[4] = a1.x; a.l[5] = a1.y; a.l[6] = a1.z; a.l[7] = a1.w;
  a.l[8] = a2.x; a.l[9] = a2.y; a.l[10] = a2.z; a.l[11] = a2.w;
  fe_sqr<FpParams, false>(t, a);
  return k3_recombine(t);
}

struct Fp2K3Ops {
  typedef fp_t T;        // what ONE lane holds
  typedef Fp2Ops Mem;    // the record format in memory (c0 | c1)
  static constexpr int WORDS = 12;
  static constexpr int LANES = 3;
  static constexpr bool FUSED_Y3_TAIL = false, FUSED_Y3 = false;

  __device__ __forceinline__ static void zero(T &r) { fe_zero(r); }
  __device__ __forceinline__ static void one(T &r) {   // 1 = (1, 0, 1)
    const bool c1 = k3_role() == 1;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = c1 ? 0u : FpParams::one(i);
  }
  __device__ __forceinline__ static bool is_zero(const T &a) { return k3_all(fpl_is_zero(a)); }
  __device__ __forceinline__ static bool is_zero_canonical(const T &a, const T &b) { return k3_all(FpOps::is_zero_canonical(a, b)); }
  __device__ __forceinline__ static bool eq(const T &a, const T &b) {
    fp_t d;
    fpl_sub(d, a, b);
    return k3_all(fpl_is_zero(d));
  }
  __device__ __forceinline__ static void add(T &r, const T &a, const T &b) { fpl_add(r, a, b); }
  __device__ __forceinline__ static void sub(T &r, const T &a, const T &b) { fpl_sub(r, a, b); }
  __device__ __forceinline__ static void neg(T &r, const T &a) { fpl_neg(r, a); }
  __device__ __forceinline__ static void dbl(T &r, const T &a) { fpl_add(r, a, a); }
  __device__ __forceinline__ static void canon(T &r) { fpl_canon(r, r); }

  __device__ __forceinline__ static void mul(T &r, const T &a, const T &b);
  __device__ __forceinline__ static void sqr(T &r, const T &a);
  // lane triples run where the job is latency-bound and two wavefronts share a SIMD: no inline copy of the product
  __device__ __forceinline__ static void mul_tail(T &r, const T &a, const T &b) { mul(r, a, b); }
  __device__ __forceinline__ static void curve_b(T &r) {   // 4(1 + u) = (4, 4, 8)
    FpOps::curve_b(r);
    if (k3_role() == 2) fpl_add(r, r, r);
  }
  // memory <-> lanes: the c0 lane reads c0, the c1 lane c1, the sum lane both
  __device__ __forceinline__ static void load(T &r, const fp2_t *p) {
    const u32 role = k3_role();
    r = (role == 1) ? p->c1 : p->c0;
    if (role == 2) {
      const fp_t o = p->c1;
      fpl_add(r, r, o);
    }
  }
  __device__ __forceinline__ static void store(fp2_t *p, const T &v) {
    const u32 role = k3_role();
    if (role == 0) p->c0 = v;
    if (role == 1) p->c1 = v;
  }
};

__device__ __forceinline__ void Fp2K3Ops::mul(T &r, const T &a, const T &b) {
  r = k3_mul_vec(u32x4{a.l[0], a.l[1], a.l[2], a.l[3]}, u32x4{a.l[4], a.l[5], a.l[6], a.l[7]},
                 u32x4{a.l[8], a.l[9], a.l[10], a.l[11]}, u32x4{b.l[0], b.l[1], b.l[2], b.l[3]},
                 u32x4{b.l[4], b.l[5], b.l[6], b.l[7]}, u32x4{b.l[8], b.l[9], b.l[10], b.l[11]});
}
__device__ __forceinline__ void Fp2K3Ops::sqr(T &r, const T &a) {
  r = k3_sqr_vec(u32x4{a.l[0], a.l[1], a.l[2], a.l[3]}, u32x4{a.l[4], a.l[5], a.l[6], a.l[7]},
                 u32x4{a.l[8], a.l[9], a.l[10], a.l[11]});
}

}  // namespace bh
