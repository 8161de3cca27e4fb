// BLS12-381 prime-field arithmetic for gfx950 (and the host, for unit tests and the
// short serial tails that run on the CPU): Fr (255-bit, 8x32-bit limbs), Fp (381-bit,
// 12x32-bit limbs), Fp2 = Fp[u]/(u^2+1).  Montgomery form throughout; every value is kept
// FULLY REDUCED (< modulus) so results are bit-identical to the reference's in-memory
// representation (bls12_381 0.8.0 `Scalar`/`Fp`: 4x64 / 6x64 little-endian Montgomery limbs,
// which is the same byte string as 8x32 / 12x32 little-endian limbs).
//
// Replaces the trait calls bellman makes into `ff`/`bls12_381`:
//   src/domain.rs:250-258 (Fr mul/add/sub in the FFT butterflies),
//   src/multiexp.rs:39,273-274 (point additions -> Fp/Fp2 mul/add/sub).
//
// 32-bit limbs because the widest integer multiplier on CDNA4 is v_mad_u64_u32
// (32x32+64 -> 64); no MFMA: this is modular big-integer arithmetic, not a contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BH_HD __host__ __device__ __forceinline__
// Out-of-line, by-value (operands and result travel in VGPRs): a 12-limb Montgomery product is
// ~700 instructions; inlining 10-40 of them per point operation blows the 64 KiB instruction
// cache and makes hipcc take tens of minutes.
#define BH_NOINLINE_HD __host__ __device__ __attribute__((noinline))

namespace bh {
typedef uint32_t u32;
typedef uint64_t u64;

// ---------------------------------------------------------------------------------------
// Field parameter packs.  mod(i)/one(i)/r2(i) are constexpr so that, after full unrolling,
// every limb is an immediate / SGPR constant.
// ---------------------------------------------------------------------------------------
struct FrParams {
  static constexpr int N = 8;
  static constexpr u32 INV = 0xffffffffu;  // -q^-1 mod 2^32
  BH_HD static constexpr u32 mod(int i) {
    constexpr u32 m[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                          0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    return m[i];
  }
  BH_HD static constexpr u32 one(int i) {  // R = 2^256 mod q
    constexpr u32 m[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                          0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
    return m[i];
  }
  BH_HD static constexpr u32 r2(int i) {  // R^2 mod q
    constexpr u32 m[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                          0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
    return m[i];
  }
};

struct FpParams {
  static constexpr int N = 12;
  static constexpr u32 INV = 0xfffcfffdu;  // -p^-1 mod 2^32
  BH_HD static constexpr u32 mod(int i) {
    constexpr u32 m[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu,
                           0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u,
                           0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
    return m[i];
  }
  BH_HD static constexpr u32 one(int i) {  // R = 2^384 mod p
    constexpr u32 m[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu,
                           0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u,
                           0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
    return m[i];
  }
  BH_HD static constexpr u32 r2(int i) {  // R^2 mod p
    constexpr u32 m[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u,
                           0x4c95b6d5u, 0x8de5476cu, 0x939d83c0u, 0x67eb88a9u,
                           0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
    return m[i];
  }
};

template <class P>
struct alignas(16) Fe {
  u32 l[P::N];
};
typedef Fe<FrParams> fr_t;
typedef Fe<FpParams> fp_t;

// ---------------------------------------------------------------------------------------
// carry helpers (single definition so the lowering can be tuned in one place)
// ---------------------------------------------------------------------------------------
// The clang carry builtins lower to v_add_co_u32 / v_addc_co_u32 (carry in VCC): one instruction per
// limb.  Written with 64-bit arithmetic the same chains became v_lshl_add_u64 + shift + mask per limb.
BH_HD u32 addc(u32 a, u32 b, u32 cin, u32 &cout) {
  unsigned co;
  const u32 s = __builtin_addc(a, b, cin, &co);
  cout = co;
  return s;
}
BH_HD u32 subb(u32 a, u32 b, u32 bin, u32 &bout) {
  unsigned bo;
  const u32 d = __builtin_subc(a, b, bin, &bo);
  bout = bo;
  return d;
}

template <class P>
BH_HD void fe_zero(Fe<P> &r) {
#pragma unroll
  for (int i = 0; i < P::N; i++) r.l[i] = 0;
}
template <class P>
BH_HD void fe_one(Fe<P> &r) {
#pragma unroll
  for (int i = 0; i < P::N; i++) r.l[i] = P::one(i);
}
template <class P>
BH_HD bool fe_is_zero(const Fe<P> &a) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < P::N; i++) o |= a.l[i];
  return o == 0;
}
template <class P>
BH_HD bool fe_eq(const Fe<P> &a, const Fe<P> &b) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < P::N; i++) o |= a.l[i] ^ b.l[i];
  return o == 0;
}

// r = t - mod if t >= mod else t   (t < 2*mod, no overflow word)
template <class P>
BH_HD void fe_reduce_once(Fe<P> &r, const u32 *t) {
  u32 d[P::N];
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < P::N; i++) d[i] = subb(t[i], P::mod(i), br, br);
#pragma unroll
  for (int i = 0; i < P::N; i++) r.l[i] = br ? t[i] : d[i];
}

template <class P>
BH_HD void fe_add(Fe<P> &r, const Fe<P> &a, const Fe<P> &b) {
  // a + b < 2*mod < 2^(32N): both moduli leave >= 1 spare top bit, so no carry out.
  u32 t[P::N];
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < P::N; i++) t[i] = addc(a.l[i], b.l[i], c, c);
  fe_reduce_once<P>(r, t);
}

template <class P>
BH_HD void fe_sub(Fe<P> &r, const Fe<P> &a, const Fe<P> &b) {
  u32 t[P::N];
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < P::N; i++) t[i] = subb(a.l[i], b.l[i], br, br);
  u32 mask = 0u - br;  // all ones when a < b: add the modulus back
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < P::N; i++) r.l[i] = addc(t[i], P::mod(i) & mask, c, c);
}

template <class P>
BH_HD void fe_neg(Fe<P> &r, const Fe<P> &a) {
  bool z = fe_is_zero(a);
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < P::N; i++) {
    u32 d = subb(P::mod(i), a.l[i], br, br);
    r.l[i] = z ? 0u : d;
  }
}

template <class P>
BH_HD void fe_dbl(Fe<P> &r, const Fe<P> &a) {
  fe_add(r, a, a);
}

// ---------------------------------------------------------------------------------------
// Montgomery product  r = a*b*2^(-32N) mod m.
//
// gfx950's only wide integer multiplier is v_mad_u64_u32 (32x32 + 64 -> 64, ~26 T/s measured,
// profiles/r1_microbench_int.txt) and it has no carry-in, so a 32-bit-limb CIOS spends more
// time on carry adds and register moves than on multiplying (1200 instructions, 288 of them
// mads).  Instead the operands are re-sliced into L limbs of 30 bits: a column of the
// schoolbook product is then a sum of <= L products < 2^60, which a 64-bit accumulator holds
// without overflow, i.e. a pure chain of mads with no carry handling at all.  The reduction is
// done column-wise in the same radix with R' = 2^(30L); feeding b*2^(30L-32N) instead of b
// makes the result a*b*2^(-32N), the Montgomery form used in memory.  Inputs and output are the
// fully reduced 32-bit-limb representation.
// ---------------------------------------------------------------------------------------
template <class P>
struct Radix30 {
  static constexpr int N = P::N;
  static constexpr int L = (32 * N + 29) / 30;        // 13 for Fp, 9 for Fr
  static constexpr int SHIFT = 30 * L - 32 * N;       // 6 for Fp, 14 for Fr
  static constexpr u32 MASK = 0x3fffffffu;
  static constexpr u32 INV = P::INV & MASK;           // -m^-1 mod 2^30
  // bits [30*i, 30*i+30) of the modulus
  BH_HD static constexpr u32 mod(int i) {
    const int bp = 30 * i, w = bp / 32, sh = bp % 32;
    u64 v = (w < N) ? P::mod(w) : 0u;
    if (w + 1 < N) v |= (u64)P::mod(w + 1) << 32;
    return (u32)(v >> sh) & MASK;
  }
  // Column k of the reduction below sums c[k] (<= min(k+1, 2L-1-k) limb products), the carry of column k-1 and up to L
  // quotient-times-modulus products.  With every limb of both operands at 2^30 - 1 that stays below 2^64 in all 18
  // columns of Fr and in 21 of the 26 columns of Fp (all but 10..14): bit k of NOSPLIT says that column k may add c[k]
  // whole instead of sending its high part round the carry - three instructions fewer per column, two of them 64-bit
  // (an and, a 64-bit shift, a 64-bit add).  Evaluated at compile time from the modulus, for ANY operand words.
  BH_HD static constexpr u32 nosplit_mask() {
    u32 mask = 0;
    unsigned __int128 carry = 0;
    for (int k = 0; k < 2 * L; k++) {
      const int terms = k < L ? k + 1 : 2 * L - 1 - k;
      unsigned __int128 t = (unsigned __int128)terms * MASK * MASK + carry + MASK;
      for (int i = (k < L ? 0 : k - L + 1); i <= (k < L ? k : L - 1); i++) t += (unsigned __int128)MASK * mod(k - i);
      if ((t >> 64) == 0) mask |= 1u << k;
      carry = (t >> 30) + 1;   // bounds the carry of either form of the column
    }
    return mask;
  }
  static constexpr u32 NOSPLIT = nosplit_mask();
};
static_assert(Radix30<FrParams>::NOSPLIT == 0x3ffffu, "Fr: no column of the reduction needs the split");
static_assert(Radix30<FpParams>::NOSPLIT == (0x3ffffffu & ~(0x1fu << 10)), "Fp: columns 10..14 keep the split");

// limb i (30 bits) of (x << SH), x given as N 32-bit words
template <class P, int SH>
BH_HD u32 fe_limb30(const Fe<P> &x, int i) {
#ifdef BH_DIAG_NO_SLICE   // TIMING-ONLY diagnostic build (tools/r6/gpu_call12.sh): what if operands were already 30-bit limbs? (wrong results)
  return x.l[i < P::N ? i : 0] & 0x3fffffffu;
#endif
  const int bp = 30 * i - SH;
  if (bp < 0) return (x.l[0] << (-bp)) & 0x3fffffffu;   // only limb 0, since SH < 30
  const int w = bp / 32, sh = bp % 32;
  u32 v = (w < P::N) ? (x.l[w] >> sh) : 0u;
  if (sh > 2 && w + 1 < P::N) v |= x.l[w + 1] << (32 - sh);
  return v & 0x3fffffffu;
}

// column-wise Montgomery reduction of the 2L product columns c[] + repack + final subtraction
template <class P, bool CANONICAL = true, u32 WHOLE = Radix30<P>::NOSPLIT>
BH_HD void fe_mont_reduce30(Fe<P> &r, const u64 *c) {
  typedef Radix30<P> R;
  constexpr int N = P::N, L = R::L;
  // Column k holds c[k] + carry + sum m[i]*mod[k-i].  Where that can pass 2^64 (the middle columns of Fp: WHOLE = R::NOSPLIT)
  // the high part of c[k] goes straight into the next carry (t < 2^30 + carry + L*2^60 with carry < 2^35); everywhere
  // else c[k] is added whole.  Both forms give the same t mod 2^30 and the same carry.
  static_assert(2 * L <= 32, "one bit of NOSPLIT per column");
  u32 m[L], out[L];
  u64 carry = 0;
#pragma unroll
  for (int k = 0; k < L; k++) {
    const bool whole = (WHOLE >> k) & 1;
    u64 t = (whole ? c[k] : (c[k] & R::MASK)) + carry;
#pragma unroll
    for (int i = 0; i < k; i++) t += (u64)m[i] * R::mod(k - i);
    m[k] = ((u32)t * R::INV) & R::MASK;
    t += (u64)m[k] * R::mod(0);      // low 30 bits of t are now zero
    carry = whole ? (t >> 30) : (t >> 30) + (c[k] >> 30);
  }
#pragma unroll
  for (int k = L; k < 2 * L; k++) {
    const bool whole = (WHOLE >> k) & 1;
    u64 t = (whole ? c[k] : (c[k] & R::MASK)) + carry;
#pragma unroll
    for (int i = k - L + 1; i < L; i++) t += (u64)m[i] * R::mod(k - i);
    out[k - L] = (u32)t & R::MASK;
    carry = whole ? (t >> 30) : (t >> 30) + (c[k] >> 30);
  }
#ifdef BH_DIAG_NO_SLICE   // ... and stayed 30-bit limbs (no repack)
  if (!CANONICAL) {
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = out[i];
    return;
  }
#endif
  // repack 30-bit limbs into 32-bit words (value < 2m < 2^(32N))
  u32 w[N];
  u64 acc = 0;
  int bits = 0, wi = 0;
#pragma unroll
  for (int i = 0; i < L; i++) {
    acc |= (u64)out[i] << bits;
    bits += 30;
    if (bits >= 32 && wi < N) {
      w[wi++] = (u32)acc;
      acc >>= 32;
      bits -= 32;
    }
  }
  if (wi < N) w[wi] = (u32)acc;
  if (CANONICAL) {
    fe_reduce_once<P>(r, w);
  } else {   // lazily reduced result (see the fpl_* helpers): < 1.5 m for operands < 2 m
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = w[i];
  }
}

template <class P, bool CANONICAL = true>
BH_HD void fe_mul(Fe<P> &r, const Fe<P> &a, const Fe<P> &b) {
  typedef Radix30<P> R;
  constexpr int L = R::L;
  u32 A[L], B[L];
#pragma unroll
  for (int i = 0; i < L; i++) {
    A[i] = fe_limb30<P, 0>(a, i);
    B[i] = fe_limb30<P, R::SHIFT>(b, i);
  }
  // product columns: c[k] = sum_{i+j=k} A[i]*B[j]  (< L * 2^60 < 2^64)
  u64 c[2 * L];
#pragma unroll
  for (int k = 0; k < 2 * L; k++) c[k] = 0;
#pragma unroll
  for (int i = 0; i < L; i++) {
#pragma unroll
    for (int j = 0; j < L; j++) c[i + j] += (u64)A[i] * B[j];
  }
  fe_mont_reduce30<P, CANONICAL>(r, c);
}

// The same product with the second operand PRE-SLICED ("B form": the L 30-bit limbs of b << SHIFT).  Table
// entries that are only ever multiplied with - FFT twiddles, coset factors - are stored this way, which removes
// the re-slicing of one operand from every product (about 12 % of an Fr product's instructions).
template <class P>
BH_HD void fe_to_bform(u32 *B, const Fe<P> &b) {
  typedef Radix30<P> R;
#pragma unroll
  for (int i = 0; i < R::L; i++) B[i] = fe_limb30<P, R::SHIFT>(b, i);
}
template <class P, bool CANONICAL = true>
BH_HD void fe_mul_b(Fe<P> &r, const Fe<P> &a, const u32 *B) {
  typedef Radix30<P> R;
  constexpr int L = R::L;
  u32 A[L];
#pragma unroll
  for (int i = 0; i < L; i++) A[i] = fe_limb30<P, 0>(a, i);
  u64 c[2 * L];
#pragma unroll
  for (int k = 0; k < 2 * L; k++) c[k] = 0;
#pragma unroll
  for (int i = 0; i < L; i++) {
#pragma unroll
    for (int j = 0; j < L; j++) c[i + j] += (u64)A[i] * B[j];
  }
  fe_mont_reduce30<P, CANONICAL>(r, c);
}

// r = (a*b + c*d) * 2^(-32N): two products under ONE reduction (L*L fewer mads than two separate products: 169 of
// Fp's 676).  Both products accumulate into the same 2L columns; a column that could pass 2^64 under the second
// product (or in the reduction afterwards) first sends its high part to its neighbour - Radix30Fused decides per column
// at compile time, with every limb of all four operands at 2^30 - 1.  Result < (a*b + c*d) / 2^(32N) + m.
// Used for the last line of the mixed addition (ec.cuh, Y3) and for the lane-pair Fp2 product (fp2pair.cuh).
template <class P>
struct Radix30Fused {
  typedef Radix30<P> R;
  static constexpr int L = R::L;
  struct Plan { u32 relieve, whole; };
  BH_HD static constexpr Plan plan() {
    Plan pl{0, 0};
    unsigned __int128 col[2 * L] = {};
    const unsigned __int128 MM = (unsigned __int128)R::MASK * R::MASK;
    for (int k = 0; k < 2 * L; k++) col[k] = (unsigned __int128)(k < L ? k + 1 : 2 * L - 1 - k) * MM;   // first product
    for (int k = 0; k < 2 * L - 1; k++) {
      const unsigned __int128 second = (unsigned __int128)(k < L ? k + 1 : 2 * L - 1 - k) * MM;
      // relieve only where the two products could pass 2^64 (9 columns of Fp; the reduction then splits 9 - relieving
      // earlier, from 2^63, costs 17 + 5)
      if (((col[k] + second) >> 64) != 0) {
        pl.relieve |= 1u << k;
        col[k + 1] += col[k] >> 30;
        col[k] = R::MASK;
      }
    }
    for (int k = 0; k < 2 * L; k++) col[k] += (unsigned __int128)(k < L ? k + 1 : 2 * L - 1 - k) * MM;   // second product
    unsigned __int128 carry = 0;
    for (int k = 0; k < 2 * L; k++) {
      unsigned __int128 t = col[k] + carry + R::MASK;
      for (int i = (k < L ? 0 : k - L + 1); i <= (k < L ? k : L - 1); i++) t += (unsigned __int128)R::MASK * R::mod(k - i);
      if ((t >> 64) == 0) pl.whole |= 1u << k;
      carry = (t >> 30) + 1;
    }
    return pl;
  }
  static constexpr Plan PLAN = plan();
  // no column may overflow BEFORE the reduction either: every column after both products stays below 2^64
  BH_HD static constexpr bool columns_fit() {
    unsigned __int128 col[2 * L] = {};
    const unsigned __int128 MM = (unsigned __int128)R::MASK * R::MASK;
    for (int k = 0; k < 2 * L; k++) col[k] = (unsigned __int128)(k < L ? k + 1 : 2 * L - 1 - k) * MM;
    for (int k = 0; k < 2 * L - 1; k++)
      if ((PLAN.relieve >> k) & 1) {
        if ((col[k + 1] + (col[k] >> 30)) >> 64) return false;
        col[k + 1] += col[k] >> 30;
        col[k] = R::MASK;
      }
    for (int k = 0; k < 2 * L; k++) {
      col[k] += (unsigned __int128)(k < L ? k + 1 : 2 * L - 1 - k) * MM;
      if (col[k] >> 64) return false;
    }
    return true;
  }
};
template <class P, bool CANONICAL = true>
BH_HD void fe_mul2(Fe<P> &r, const Fe<P> &a, const Fe<P> &b, const Fe<P> &c2, const Fe<P> &d) {
  typedef Radix30<P> R;
  typedef Radix30Fused<P> RF;
  constexpr int L = R::L;
  static_assert(RF::columns_fit(), "a product column overflows before the reduction");
  u64 c[2 * L];
#pragma unroll
  for (int k = 0; k < 2 * L; k++) c[k] = 0;
  {
    u32 A[L], B[L];
#pragma unroll
    for (int i = 0; i < L; i++) {
      A[i] = fe_limb30<P, 0>(a, i);
      B[i] = fe_limb30<P, R::SHIFT>(b, i);
    }
#pragma unroll
    for (int i = 0; i < L; i++) {
#pragma unroll
      for (int j = 0; j < L; j++) c[i + j] += (u64)A[i] * B[j];
    }
  }
#pragma unroll
  for (int k = 0; k < 2 * L - 1; k++) {
    if ((RF::PLAN.relieve >> k) & 1) {
      c[k + 1] += c[k] >> 30;
      c[k] &= R::MASK;
    }
  }
  {
    u32 A[L], B[L];
#pragma unroll
    for (int i = 0; i < L; i++) {
      A[i] = fe_limb30<P, 0>(c2, i);
      B[i] = fe_limb30<P, R::SHIFT>(d, i);
    }
#ifdef BH_DIAG_HALF_SECOND_PRODUCT   // TIMING-ONLY diagnostic build (tools/r6/gpu_call13.sh): the unreachable ideal of a Karatsuba
    constexpr int ROWS = (L + 1) / 2;   // split over a lane pair - 1.5 products + one reduction per lane, exchanges free (wrong results)
#else
    constexpr int ROWS = L;
#endif
#pragma unroll
    for (int i = 0; i < ROWS; i++) {
#pragma unroll
      for (int j = 0; j < L; j++) c[i + j] += (u64)A[i] * B[j];
    }
  }
  fe_mont_reduce30<P, CANONICAL, RF::PLAN.whole>(r, c);
}

// Also evaluated and rejected on the MI355X: the interleaved product-scanning form (retire each column
// immediately; 70 instead of 97 VGPRs for the out-of-line call) - 24 % more 64-bit add/shift work,
// 46 vs 57 G mul/s, and callers still do not reach three waves per SIMD.
//
// Squaring: the product above is a * (b * 2^SHIFT); with SHIFT even the same scaling comes from
// (a * 2^(SHIFT/2))^2, whose operands are equal - so the off-diagonal limb products are computed once and
// doubled: L(L-1)/2 + L = 91 instead of 169 product mads for Fp (the reduction is unchanged).
template <class P, bool CANONICAL = true>
BH_HD void fe_sqr(Fe<P> &r, const Fe<P> &a) {
  typedef Radix30<P> R;
  constexpr int L = R::L;
  static_assert(R::SHIFT % 2 == 0, "squaring splits the Montgomery pre-shift evenly");
  u32 A[L];
#pragma unroll
  for (int i = 0; i < L; i++) A[i] = fe_limb30<P, R::SHIFT / 2>(a, i);
  u64 c[2 * L];
#pragma unroll
  for (int k = 0; k < 2 * L; k++) c[k] = 0;
#pragma unroll
  for (int i = 0; i < L; i++) {
#pragma unroll
    for (int j = i + 1; j < L; j++) c[i + j] += (u64)A[i] * A[j];   // < (L/2) * 2^60 per column
  }
#pragma unroll
  for (int k = 0; k < 2 * L; k++) c[k] <<= 1;
#pragma unroll
  for (int i = 0; i < L; i++) c[2 * i] += (u64)A[i] * A[i];        // column total < L * 2^60 < 2^64
  fe_mont_reduce30<P, CANONICAL>(r, c);
}

// canonical <-> Montgomery
template <class P>
BH_HD void fe_to_mont(Fe<P> &r, const Fe<P> &a) {
  Fe<P> r2;
#pragma unroll
  for (int i = 0; i < P::N; i++) r2.l[i] = P::r2(i);
  fe_mul(r, a, r2);
}
template <class P>
BH_HD void fe_from_mont(Fe<P> &r, const Fe<P> &a) {
  Fe<P> one;
  fe_zero(one);
  one.l[0] = 1;
  fe_mul(r, a, one);
}

// a^e for a little-endian 32-bit-limb exponent (host tails / setup; not a hot path)
template <class P>
BH_HD void fe_pow(Fe<P> &r, const Fe<P> &a, const u32 *e, int nlimbs) {
  Fe<P> acc;
  fe_one(acc);
  for (int i = nlimbs * 32 - 1; i >= 0; i--) {
    fe_sqr(acc, acc);
    if ((e[i / 32] >> (i % 32)) & 1) fe_mul(acc, acc, a);
  }
  r = acc;
}
template <class P>
BH_HD void fe_inv(Fe<P> &r, const Fe<P> &a) {  // a^(m-2); a != 0
  u32 e[P::N];
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < P::N; i++) e[i] = subb(P::mod(i), i == 0 ? 2u : 0u, br, br);  // m - 2 (Fr's low limb is 1)
  fe_pow(r, a, e, P::N);
}

// out-of-line Fp product used by all curve code (the FFT keeps its 8-limb Fr product inline)
// ---------------------------------------------------------------------------------------
// Lazily reduced Fp for the curve code.  All Fp values handled by FpOps / Fp2Ops live in [0, 2p):
// the Montgomery product of two such values is < p (256 p / 2^390 + 1) < 1.5 p without its final
// conditional subtraction (R = 2^390 is 2^9 times larger than p), so the products skip it; add / sub
// keep the invariant with one conditional +-2p, exactly what the canonical versions spend on +-p.
// Zero has two representatives (0 and p): is_zero / eq know both.  Values are made canonical where they
// leave the curve code (xyzz_to_affine, the host tail of the MSM).
// ---------------------------------------------------------------------------------------
BH_HD constexpr u32 fp_mod2(int i) {   // limb i of 2p (< 2^382)
  return (FpParams::mod(i) << 1) | (i ? FpParams::mod(i - 1) >> 31 : 0u);
}
BH_HD void fpl_add(fp_t &r, const fp_t &a, const fp_t &b) {   // a + b < 4p < 2^384
  u32 t[12], d[12];
  u32 c = 0, br = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) t[i] = addc(a.l[i], b.l[i], c, c);
#pragma unroll
  for (int i = 0; i < 12; i++) d[i] = subb(t[i], fp_mod2(i), br, br);
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = br ? t[i] : d[i];
}
BH_HD void fpl_sub(fp_t &r, const fp_t &a, const fp_t &b) {
  u32 t[12];
  u32 br = 0, c = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) t[i] = subb(a.l[i], b.l[i], br, br);
  const u32 mask = 0u - br;
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = addc(t[i], fp_mod2(i) & mask, c, c);
}
// Two independent additions / subtractions with their limb operations interleaved: consecutive
// instructions then belong to different carry chains (one carry in VCC, the other in an SGPR pair), which
// removes the wait states a single v_addc/v_subb chain needs between dependent limbs.  Fp2 add/sub are
// exactly such pairs.
BH_HD void fpl_sub2(fp_t &r0, const fp_t &a0, const fp_t &b0, fp_t &r1, const fp_t &a1, const fp_t &b1) {
  u32 t0[12], t1[12];
  u32 br0 = 0, br1 = 0, c0 = 0, c1 = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    t0[i] = subb(a0.l[i], b0.l[i], br0, br0);
    t1[i] = subb(a1.l[i], b1.l[i], br1, br1);
  }
  const u32 m0 = 0u - br0, m1 = 0u - br1;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    r0.l[i] = addc(t0[i], fp_mod2(i) & m0, c0, c0);
    r1.l[i] = addc(t1[i], fp_mod2(i) & m1, c1, c1);
  }
}
BH_HD void fpl_add2(fp_t &r0, const fp_t &a0, const fp_t &b0, fp_t &r1, const fp_t &a1, const fp_t &b1) {
  u32 t0[12], t1[12], d0[12], d1[12];
  u32 c0 = 0, c1 = 0, br0 = 0, br1 = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    t0[i] = addc(a0.l[i], b0.l[i], c0, c0);
    t1[i] = addc(a1.l[i], b1.l[i], c1, c1);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) {
    d0[i] = subb(t0[i], fp_mod2(i), br0, br0);
    d1[i] = subb(t1[i], fp_mod2(i), br1, br1);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) {
    r0.l[i] = br0 ? t0[i] : d0[i];
    r1.l[i] = br1 ? t1[i] : d1[i];
  }
}
BH_HD bool fpl_is_zero(const fp_t &a) {
  u32 o = 0, q = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) { o |= a.l[i]; q |= a.l[i] ^ FpParams::mod(i); }
  return o == 0 || q == 0;
}
BH_HD void fpl_neg(fp_t &r, const fp_t &a) {   // 2p - a, and -0 = 0 (a = 0 would give 2p)
  const bool z = fe_is_zero(a);
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    const u32 d = subb(fp_mod2(i), a.l[i], br, br);
    r.l[i] = z ? 0u : d;
  }
}
BH_HD void fpl_canon(fp_t &r, const fp_t &a) {   // [0, 2p) -> [0, p)
  u32 t[12];
#pragma unroll
  for (int i = 0; i < 12; i++) t[i] = a.l[i];
  fe_reduce_once<FpParams>(r, t);
}
BH_HD bool fpl_eq(const fp_t &a, const fp_t &b) {
  fp_t d;
  fpl_sub(d, a, b);
  return fpl_is_zero(d);
}

// The operands travel as six 4-word vectors, not as two structs: clang's AMDGPU ABI passes at most 16
// registers' worth of *aggregate* arguments directly, so the second 12-word struct of a (fp_t, fp_t)
// signature went through scratch memory - a store/load round trip in front of every product that the
// one or two resident wavefronts of the G2 kernels could not hide.  Scalars and vectors are not
// subject to that cap: all 24 words arrive in v0-v23.
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
BH_NOINLINE_HD static fp_t fp_mul_vec(u32x4 a0, u32x4 a1, u32x4 a2, u32x4 b0, u32x4 b1, u32x4 b2) {
  fp_t a, b, r;
  a.l[0] = a0.x; a.l[1] = a0.y; a.l[2] = a0.z; a.l[3] = a0.w;
  a.l[4] = a1.x; a.l[5] = a1.y; a.l[6] = a1.z; a.l[7] = a1.w;
  a.l[8] = a2.x; a.l[9] = a2.y; a.l[10] = a2.z; a.l[11] = a2.w;
  b.l[0] = b0.x; b.l[1] = b0.y; b.l[2] = b0.z; b.l[3] = b0.w;
  b.l[4] = b1.x; b.l[5] = b1.y; b.l[6] = b1.z; b.l[7] = b1.w;
  b.l[8] = b2.x; b.l[9] = b2.y; b.l[10] = b2.z; b.l[11] = b2.w;
  fe_mul<FpParams, false>(r, a, b);   // lazily reduced
  return r;
}
BH_NOINLINE_HD static fp_t fp_sqr_vec(u32x4 a0, u32x4 a1, u32x4 a2) {
  fp_t a, r;
  a.l[0] = a0.x; a.l[1] = a0.y; a.l[2] = a0.z; a.l[3] = a0.w;
  a.l[4] = a1.x; a.l[5] = a1.y; a.l[6] = a1.z; a.l[7] = a1.w;
  a.l[8] = a2.x; a.l[9] = a2.y; a.l[10] = a2.z; a.l[11] = a2.w;
  fe_sqr<FpParams, false>(r, a);      // lazily reduced
  return r;
}
BH_HD fp_t fp_sqr_call(const fp_t &a) {
  return fp_sqr_vec(u32x4{a.l[0], a.l[1], a.l[2], a.l[3]}, u32x4{a.l[4], a.l[5], a.l[6], a.l[7]},
                    u32x4{a.l[8], a.l[9], a.l[10], a.l[11]});
}
BH_HD fp_t fp_mul_call(const fp_t &a, const fp_t &b) {
  return fp_mul_vec(u32x4{a.l[0], a.l[1], a.l[2], a.l[3]}, u32x4{a.l[4], a.l[5], a.l[6], a.l[7]},
                    u32x4{a.l[8], a.l[9], a.l[10], a.l[11]}, u32x4{b.l[0], b.l[1], b.l[2], b.l[3]},
                    u32x4{b.l[4], b.l[5], b.l[6], b.l[7]}, u32x4{b.l[8], b.l[9], b.l[10], b.l[11]});
}

// ---------------------------------------------------------------------------------------
// Fp2
// ---------------------------------------------------------------------------------------
struct alignas(16) fp2_t {
  fp_t c0, c1;
};

// "Field ops" bundles so curve code is written once for G1 (Fp) and G2 (Fp2).
struct FpOps {
  typedef fp_t T;
  typedef FpOps Mem;                   // record format in memory == what a lane holds
  static constexpr int WORDS = 12;
  static constexpr int LANES = 1;      // lanes per group element (3 for the K3 form of Fp2, fp2k3.cuh; 2: fp2pair.cuh)
  // the last line of the mixed addition as ONE fused product a*b - c*d (ec.cuh xyzz_madd): in the overload that
  // prefetches (..._TAIL) / in the plain one
  static constexpr bool FUSED_Y3_TAIL = true, FUSED_Y3 = true;
  BH_HD static void load(T &r, const T *p) { r = *p; }
  BH_HD static void store(T *p, const T &v) { *p = v; }
  BH_HD static void zero(T &r) { fe_zero(r); }
  BH_HD static void one(T &r) { fe_one(r); }
  BH_HD static bool is_zero(const T &a) { return fpl_is_zero(a); }
  BH_HD static bool is_zero_canonical(const T &a, const T &b) {   // both of two CANONICAL values are zero
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) o |= a.l[i] | b.l[i];
    return o == 0;
  }
  BH_HD static bool eq(const T &a, const T &b) { return fpl_eq(a, b); }
  BH_HD static void add(T &r, const T &a, const T &b) { fpl_add(r, a, b); }
  BH_HD static void sub(T &r, const T &a, const T &b) { fpl_sub(r, a, b); }
  BH_HD static void neg(T &r, const T &a) { fpl_neg(r, a); }
  BH_HD static void dbl(T &r, const T &a) { fpl_add(r, a, a); }
  BH_HD static void canon(T &r) { fpl_canon(r, r); }
  BH_HD static void mul(T &r, const T &a, const T &b) { r = fp_mul_call(a, b); }
  BH_HD static void sqr(T &r, const T &a) { r = fp_sqr_call(a); }
  // the same product INLINE (no call): for the one place where loads are in flight across it (ec.cuh, xyzz_madd)
  BH_HD static void mul_tail(T &r, const T &a, const T &b) { fe_mul<FpParams, false>(r, a, b); }
  // r = a*b - c*d, inline, one reduction (fe_mul2 on 2p - c); operands in [0, 2p) -> r < 1.82 p
  BH_HD static void mul2_sub_tail(T &r, const T &a, const T &b, const T &c, const T &d) {
    T nc;
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) nc.l[i] = subb(fp_mod2(i), c.l[i], br, br);   // 2p - c in (0, 2p]
    fe_mul2<FpParams, false>(r, a, b, nc, d);
  }
  BH_HD static void mul2_sub(T &r, const T &a, const T &b, const T &c, const T &d) { mul2_sub_tail(r, a, b, c, d); }
  BH_HD static void curve_b(T &r) {   // G1: y^2 = x^3 + 4
    T one2;
    fe_one(r);
    fe_add(one2, r, r);
    fe_add(r, one2, one2);
  }
  BH_HD static void inv(T &r, const T &a) {  // a^(p-2)
    u32 e[12];
#pragma unroll
    for (int i = 0; i < 12; i++) e[i] = FpParams::mod(i);
    e[0] -= 2;
    T acc;
    fe_one(acc);
    for (int i = 383; i >= 0; i--) {
      acc = fp_sqr_call(acc);
      if ((e[i / 32] >> (i % 32)) & 1) acc = fp_mul_call(acc, a);
    }
    r = acc;
  }
};

struct Fp2Ops {
  typedef fp2_t T;
  typedef Fp2Ops Mem;
  static constexpr int WORDS = 24;
  static constexpr int LANES = 1;
  // fused only where it was measured to pay: the software-pipelined accumulation (one lane per point, inline tail);
  // the other one-lane G2 kernels keep Karatsuba over out-of-line products (code size)
  static constexpr bool FUSED_Y3_TAIL = true, FUSED_Y3 = false;
  BH_HD static void load(T &r, const T *p) { r = *p; }
  BH_HD static void store(T *p, const T &v) { *p = v; }
  BH_HD static void zero(T &r) { fe_zero(r.c0); fe_zero(r.c1); }
  BH_HD static void one(T &r) { fe_one(r.c0); fe_zero(r.c1); }
  BH_HD static bool is_zero(const T &a) { return fpl_is_zero(a.c0) && fpl_is_zero(a.c1); }
  BH_HD static bool is_zero_canonical(const T &a, const T &b) {
    return FpOps::is_zero_canonical(a.c0, a.c1) && FpOps::is_zero_canonical(b.c0, b.c1);
  }
  BH_HD static bool eq(const T &a, const T &b) { return fpl_eq(a.c0, b.c0) && fpl_eq(a.c1, b.c1); }
  BH_HD static void add(T &r, const T &a, const T &b) { fpl_add2(r.c0, a.c0, b.c0, r.c1, a.c1, b.c1); }
  BH_HD static void sub(T &r, const T &a, const T &b) { fpl_sub2(r.c0, a.c0, b.c0, r.c1, a.c1, b.c1); }
  BH_HD static void neg(T &r, const T &a) { fpl_neg(r.c0, a.c0); fpl_neg(r.c1, a.c1); }
  BH_HD static void dbl(T &r, const T &a) { add(r, a, a); }
  BH_HD static void canon(T &r) { fpl_canon(r.c0, r.c0); fpl_canon(r.c1, r.c1); }
  BH_HD static void mul(T &r, const T &a, const T &b) {
    // Karatsuba: 3 Fp products.  (Measured: making this an out-of-line by-value call passes 16 of the
    // 48 argument words through scratch and is 1.6x slower on the G2 accumulate kernel.)
    fp_t t0, t1, t2, t3;
    t0 = fp_mul_call(a.c0, b.c0);
    t1 = fp_mul_call(a.c1, b.c1);
    fpl_add(t2, a.c0, a.c1);
    fpl_add(t3, b.c0, b.c1);
    t2 = fp_mul_call(t2, t3);
    fpl_sub(t2, t2, t0);
    fpl_sub(r.c1, t2, t1);
    fpl_sub(r.c0, t0, t1);
  }
  BH_HD static void mul_tail(T &r, const T &a, const T &b) {   // Karatsuba on the inline Fp product
    fp_t t0, t1, t2, t3;
    fe_mul<FpParams, false>(t0, a.c0, b.c0);
    fe_mul<FpParams, false>(t1, a.c1, b.c1);
    fpl_add(t2, a.c0, a.c1);
    fpl_add(t3, b.c0, b.c1);
    fe_mul<FpParams, false>(t2, t2, t3);
    fpl_sub(t2, t2, t0);
    fpl_sub(r.c1, t2, t1);
    fpl_sub(r.c0, t0, t1);
  }
  // r = a*b - c*d: Karatsuba whose three products are each ONE fused product (fe_mul2) - three reductions instead of
  // six; inline like mul_tail
  BH_HD static void mul2_sub_tail(T &r, const T &a, const T &b, const T &c, const T &d) {
    fp_t l0, l1, l2, sa, sb, sc, sd;
    FpOps::mul2_sub_tail(l0, a.c0, b.c0, c.c0, d.c0);
    FpOps::mul2_sub_tail(l1, a.c1, b.c1, c.c1, d.c1);
    fpl_add2(sa, a.c0, a.c1, sb, b.c0, b.c1);
    fpl_add2(sc, c.c0, c.c1, sd, d.c0, d.c1);
    FpOps::mul2_sub_tail(l2, sa, sb, sc, sd);
    fpl_sub(l2, l2, l0);
    fpl_sub(r.c1, l2, l1);
    fpl_sub(r.c0, l0, l1);
  }
  BH_HD static void mul2_sub(T &r, const T &a, const T &b, const T &c, const T &d) {   // not fused: the kernels that
    T t, u;                                                                             // are not worth the code
    mul(t, a, b);
    mul(u, c, d);
    sub(r, t, u);
  }
  BH_HD static void sqr(T &r, const T &a) {
    // (a0+a1)(a0-a1) + 2 a0 a1 u : 2 Fp products
    fp_t s, d, p;
    fpl_add(s, a.c0, a.c1);
    fpl_sub(d, a.c0, a.c1);
    p = fp_mul_call(a.c0, a.c1);
    r.c0 = fp_mul_call(s, d);
    fpl_add(r.c1, p, p);
  }
  BH_HD static void curve_b(T &r) {   // G2: y^2 = x^3 + 4(u + 1)
    FpOps::curve_b(r.c0);
    r.c1 = r.c0;
  }
  BH_HD static void inv(T &r, const T &a) {
    fp_t n, t;
    n = fp_sqr_call(a.c0);
    t = fp_sqr_call(a.c1);
    fpl_add(n, n, t);
    FpOps::inv(n, n);
    r.c0 = fp_mul_call(a.c0, n);
    t = fp_mul_call(a.c1, n);
    fpl_neg(r.c1, t);
  }
};

}  // namespace bh
