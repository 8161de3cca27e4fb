// Fp2 on lane PAIRS for the throughput-bound G2 kernels.
//
// An element a = a0 + a1*u of Fp2 = Fp[u]/(u^2+1) is held by two neighbouring lanes of a wavefront: the even lane
// holds a0, the odd lane a1.  A product is the schoolbook form with ONE reduction per lane (ff.cuh fe_mul2):
//     even lane:  c0 = a0*b0 + a1*(2p - b1)        odd lane:  c1 = a0*b1 + a1*b0
// i.e. 2 x 169 product mads + 169 reduction mads per lane = 1014 per Fp2 product - exactly Karatsuba's three
// Montgomery products in one lane (3 x 338) - but none of Karatsuba's five additions, half the operand slicing, and
// above all HALF THE STATE PER LANE: a lane carries what a G1 lane carries (48 words per XYZZ point), so the kernel runs
// at two wavefronts per SIMD, where the one-lane-per-point kernel (256 VGPR + 105 AGPR) runs at one and pays 5.5 SIMD
// cycles per instruction instead of 4.2.  The operands a lane lacks come over DPP quad permutes (both
// lanes of a pair sit in the same quad): a0 / a1 broadcast to the pair, b swapped.  Squaring:
//     even lane:  (a0 + a1)*(a0 - a1)              odd lane:  (2 a0)*a1
// Addition, subtraction, negation and doubling are lane-local; predicates are made pair-uniform with a ballot.
//
// Compared with the lane-triple form (fp2k3.cuh: one plain product per lane, 3 lanes): 2 x ~800 instead of 3 x ~740
// instructions per Fp2 product - fewer lane-instructions per point, slightly longer latency per product.  So: pairs
// where the job is throughput-bound (the bucket accumulation of large jobs), triples where it is latency-bound.
//
// Replaces what bellman gets from bls12_381's `Fp2` through the `group` traits on the G2 multiexp
// (src/multiexp.rs:39 with G = G2Projective); values stay lazily reduced in [0, 2p) like FpOps.
#pragma once
#include "fp2k3.cuh"

namespace bh {

__device__ __forceinline__ u32 pair_role() { return k3_lane() & 1u; }
template <int CTRL>
__device__ __forceinline__ fp_t pair_perm(const fp_t &v) {   // quad permute: both lanes of a pair are in one quad
  fp_t r;
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = (u32)__builtin_amdgcn_mov_dpp((int)v.l[i], CTRL, 0xf, 0xf, true);   // every source lane exists
  return r;
}
constexpr int PAIR_SWAP = 0xB1;   // quad_perm [1,0,3,2]: the partner's value
constexpr int PAIR_EVEN = 0xA0;   // quad_perm [0,0,2,2]: the even lane's value in both lanes
constexpr int PAIR_ODD = 0xF5;    // quad_perm [1,1,3,3]: the odd lane's value in both lanes

// true iff both lanes of my pair say true
__device__ __forceinline__ bool pair_all(bool mine) {
  const u32 lane = k3_lane();
  const u64 m = __ballot(mine);
  return ((m >> (lane & ~1u)) & 3u) == 3u;
}

// Out-of-line LEAF functions like fp_mul_vec (24 argument words in VGPRs, 12 back).
__device__ __attribute__((noinline)) static fp_t pair_mul_vec(u32x4 a0, u32x4 a1, u32x4 a2, u32x4 b0, u32x4 b1, u32x4 b2) {
  fp_t a, b, r;
  a.l[0] = a0.x; a.l[1] = a0.y; a.l[2] = a0.z; a.l[3] = a0.w;
  a.l[4] = a1.x; a.l[5] = a1.y; a.l[6] = a1.z; a.l[7] = a1.w;
  a.l[8] = a2.x; a.l[9] = a2.y; a.l[10] = a2.z; a.l[11] = a2.w;
  b.l[0] = b0.x; b.l[1] = b0.y; b.l[2] = b0.z; b.l[3] = b0.w;
  b.l[4] = b1.x; b.l[5] = b1.y; b.l[6] = b1.z; b.l[7] = b1.w;
  b.l[8] = b2.x; b.l[9] = b2.y; b.l[10] = b2.z; b.l[11] = b2.w;
  const bool odd = pair_role();
  const fp_t ae = pair_perm<PAIR_EVEN>(a), ao = pair_perm<PAIR_ODD>(a);   // a0, a1 in both lanes
  fp_t w = pair_perm<PAIR_SWAP>(b);                                       // even lane: b1     odd lane: b0
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    const u32 n = subb(fp_mod2(i), w.l[i], br, br);   // 2p - b1 in (0, 2p]
    w.l[i] = odd ? w.l[i] : n;
  }
  fe_mul2<FpParams, false>(r, ae, b, ao, w);          // < 1.82 p for operands <= 2p (ff.cuh)
  return r;
}
__device__ __attribute__((noinline)) static fp_t pair_sqr_vec(u32x4 a0, u32x4 a1, u32x4 a2) {
  fp_t a, r;
  a.l[0] = a0.x; a.l[1] = a0.y; a.l[2] = a0.z; a.l[3] = a0.w;
  a.l[4] = a1.x; a.l[5] = a1.y; a.l[6] = a1.z; a.l[7] = a1.w;
  a.l[8] = a2.x; a.l[9] = a2.y; a.l[10] = a2.z; a.l[11] = a2.w;
  const bool odd = pair_role();
  const fp_t o = pair_perm<PAIR_SWAP>(a);
  // even lane: (a0 + a1)(a0 - a1)        odd lane: (a0 + a0) a1
  fp_t x, y, t;
#pragma unroll
  for (int i = 0; i < 12; i++) t.l[i] = odd ? o.l[i] : a.l[i];
  fpl_add(x, o, t);
  fpl_sub(y, a, o);
#pragma unroll
  for (int i = 0; i < 12; i++) y.l[i] = odd ? a.l[i] : y.l[i];
  fe_mul<FpParams, false>(r, x, y);
  return r;
}

struct Fp2PairOps {
  typedef fp_t T;        // what ONE lane holds
  typedef Fp2Ops Mem;    // the record format in memory (c0 | c1)
  static constexpr int WORDS = 12;
  static constexpr int LANES = 2;
  static constexpr bool FUSED_Y3_TAIL = false, FUSED_Y3 = false;

  __device__ __forceinline__ static void zero(T &r) { fe_zero(r); }
  __device__ __forceinline__ static void one(T &r) {   // 1 = (1, 0)
    const bool odd = pair_role();
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = odd ? 0u : FpParams::one(i);
  }
  __device__ __forceinline__ static bool is_zero(const T &a) { return pair_all(fpl_is_zero(a)); }
  __device__ __forceinline__ static bool is_zero_canonical(const T &a, const T &b) { return pair_all(FpOps::is_zero_canonical(a, b)); }
  __device__ __forceinline__ static bool eq(const T &a, const T &b) {
    fp_t d;
    fpl_sub(d, a, b);
    return pair_all(fpl_is_zero(d));
  }
  __device__ __forceinline__ static void add(T &r, const T &a, const T &b) { fpl_add(r, a, b); }
  __device__ __forceinline__ static void sub(T &r, const T &a, const T &b) { fpl_sub(r, a, b); }
  __device__ __forceinline__ static void neg(T &r, const T &a) { fpl_neg(r, a); }
  __device__ __forceinline__ static void dbl(T &r, const T &a) { fpl_add(r, a, a); }
  __device__ __forceinline__ static void canon(T &r) { fpl_canon(r, r); }
  __device__ __forceinline__ static void mul(T &r, const T &a, const T &b) {
    r = pair_mul_vec(u32x4{a.l[0], a.l[1], a.l[2], a.l[3]}, u32x4{a.l[4], a.l[5], a.l[6], a.l[7]},
                     u32x4{a.l[8], a.l[9], a.l[10], a.l[11]}, u32x4{b.l[0], b.l[1], b.l[2], b.l[3]},
                     u32x4{b.l[4], b.l[5], b.l[6], b.l[7]}, u32x4{b.l[8], b.l[9], b.l[10], b.l[11]});
  }
  __device__ __forceinline__ static void sqr(T &r, const T &a) {
    r = pair_sqr_vec(u32x4{a.l[0], a.l[1], a.l[2], a.l[3]}, u32x4{a.l[4], a.l[5], a.l[6], a.l[7]},
                     u32x4{a.l[8], a.l[9], a.l[10], a.l[11]});
  }
  // two wavefronts share a SIMD and cover each other's loads: no inline copy of the product
  __device__ __forceinline__ static void mul_tail(T &r, const T &a, const T &b) { mul(r, a, b); }
  __device__ __forceinline__ static void curve_b(T &r) { FpOps::curve_b(r); }   // 4(1 + u) = (4, 4)
  // memory <-> lanes: the even lane reads / writes c0, the odd lane c1
  __device__ __forceinline__ static void load(T &r, const fp2_t *p) { r = *(pair_role() ? &p->c1 : &p->c0); }
  __device__ __forceinline__ static void store(fp2_t *p, const T &v) { *(pair_role() ? &p->c1 : &p->c0) = v; }
};

}  // namespace bh
