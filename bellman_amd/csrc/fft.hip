// Radix-2 number-theoretic FFT over BLS12-381 Fr for gfx950.
//
// Replaces bellman's EvaluationDomain transforms (src/domain.rs):
//   fft/ifft/coset_fft/icoset_fft (:81-125)  ->  ntt_run()
//   best_fft/serial_fft/parallel_fft (:261-372): any exact NTT produces the same canonical
//   field elements (Appendix A item 14 of SURVEY.md), so the CPU's log_cpus split is replaced
//   by a decomposition that fits the chip.
//   distribute_powers (:101-113), mul_assign (:154-170), sub_assign (:173-189),
//   divide_by_z_on_coset (:139-151)          ->  the element-wise kernels at the bottom,
//   plus the fused (a*b-c)*zinv pass used by create_proof's h block (prover.rs:232-236).
//
// Structure (MI355X-first): n = R_0 * ... * R_{L-1}, R_p = 2^r_p <= 2^11: ONE pass up to 2^11, TWO up to
// 2^22 (the 2^20 / 2^22 domains of the proof sizes that matter: 64 B of algorithmic traffic per element become
// 128 B of real traffic, not 192), three up to 2^31.  A pass gives every workgroup a tile of 2048 elements
// (R rows x C columns; 72 KiB of LDS, two workgroups per CU so one tile's global loads/stores overlap the other's
// butterflies) and performs all R-point sub-FFTs of the tile in register steps between LDS exchanges (LDS is two
// 16-byte planes with one pad slot per 8):
//   * [r5] sizes with one-level tables (2^12 .. 2^24, below) run 512 threads per workgroup: 4 elements per thread,
//     radix-4 steps (two DIT stages in registers), 128 VGPRs, FOUR wavefronts per SIMD.  The round-4 kernel - 256 threads,
//     8 elements each, radix-8 steps, two wavefronts per SIMD - had its wavefronts parked 24 % of the time (s_waitcnt /
//     s_barrier) and stalled at issue another 33 % (profiles/archive/r5_fft_pmc_wait_split.json): with a second wavefront as the
//     only cover, every park left a SIMD on one wavefront's issue rate.  Four wavefronts: -8..15 % by size and mode
//     (profiles/archive/r5_call1_fft_variants.txt).  The two-level kernel (single-pass sizes, sizes above 2^24, sizes over the
//     table budget) keeps the 256-thread radix-8 form: at 128 registers it spills.
//   * [r5] register steps whose tasks stay inside a wavefront's own block of the tile need no s_barrier between them (a
//     wavefront's LDS operations execute in order): 4 barriers per tile instead of 7 at r = 11, 2 instead of 5 at r = 8
//     (-3..5 %; the rule is in ntt_pass_kernel, its proof by enumeration in tests/models/ntt_model.py).
//   * the first step of a pass has only the constant twiddles w_4 (radix-4) / w_4, w_8^{1,2,3} (radix-8);
//   * twiddles are never gathered from an n-entry table.  In-tile twiddles come from ONE 1024-entry master table
//     w_2048^i (48 KiB, cache resident, shared by every pass of every size; reading it as one entry per wavefront instead
//     of one per lane changes nothing: r5_call1, lib_notw); inter-pass twiddles w_n^e, the coset factors 7^i / 7^-i and
//     1/n come from two-level tables (e = e_hi*2^LB + e_lo: two multiplications, tables of about 2^(log_n/2) entries),
//     stored PRE-SLICED for the multiplier ("B form": the nine 30-bit limbs of w << 14, ff.cuh);
//   * from 2^12 to 2^24 points the inter-pass twiddles, the coset factors 7^i and 7^-i are ONE-level tables laid out in
//     the order the kernel consumes them (tile order: entry t * 2048 + e belongs to element e of tile t, read coalesced
//     beside the data) - one product per element instead of the two of the hi x lo tables; the 1/n of the inverse
//     transforms is folded into the inverse twiddle table, so ifft has no scaling product at all.  [r5] Their entries are
//     the plain 32-byte Montgomery elements, sliced after the load (48-byte B form until round 4: a third more to stream
//     per element and pass for 17 of a product's 290 instructions; -3..7 % where a table streams).  The cache of these
//     tables has a budget (bh_ctx_set_limits): a size whose complete set does not fit runs on the two-level tables.
//   * butterflies compute LAZILY REDUCED in [0, 2q): every product is data x (canonical) table entry, whose Montgomery
//     result is < 1.91 q without the final conditional subtraction; additions / subtractions correct by +-2q (the sum
//     of two such values needs bit 256: the carry out of the eight words decides with the borrow).  The last pass
//     makes its outputs canonical, so what the caller sees is bit-identical to the reference's field elements
//     (extreme inputs at every pass plan and both table kinds: tests/test_gpu_fft_extremes.py).
// Non-last passes read and write the same positions (pass 0 moves the data into a scratch vector); the last
// pass scatters the digit-reversed result to natural order back into the caller's vector.  Workgroup -> tile
// assignment is XCD-aware: tiles that share 128-byte lines (the narrow tiles of the 2^21 / 2^22 plans) go to the
// same XCD back to back, so the other half of a line is an L2 hit instead of a second HBM fetch.
#include "common.hpp"

namespace bh {

// Threads per workgroup (two workgroups per CU either way: 72 KiB of LDS each).  The one-level kernel runs 512 threads:
// 4 elements per thread, radix-4 steps, 128 registers, FOUR wavefronts per SIMD - the two-level kernel (single-pass
// sizes, sizes above 2^24) keeps 256 threads with 8 elements each and radix-8 steps (at 128 registers it spills).
template <bool ONE>
struct NttCfg {
  static constexpr int TH = ONE ? 512 : 256;
  static constexpr int GMAX = TH == 256 ? 3 : 2;      // stages per register step: radix-8 or radix-4
  static constexpr int WAVES_PER_SIMD = TH / 128;
  static constexpr int LOG_WAVES = TH == 256 ? 2 : 3;
};
constexpr int NTT_ONE_STRIDE = 32;   // a one-level table entry: the 32-byte Montgomery element, sliced after the load
constexpr int NTT_LOG_TILE = 11;                  // 2048 Fr per workgroup
constexpr int NTT_TILE = 1 << NTT_LOG_TILE;
constexpr int NTT_MAX_R = 11;                     // rows of a tile: sub-FFT size 2^r, r <= 11
constexpr int NTT_PLANE = NTT_TILE + NTT_TILE / 8;   // padded slots per 16-byte plane

struct NttPass {
  const fr_t *in;
  fr_t *out;
  const fr_t *in_y[2];   // blockIdx.y = 1, 2: further vectors transformed by the same launch (single-pass sizes only)
  fr_t *out_y[2];
  const BTw *master;   // w_2048^(+-i), i < 1024
  const BTw *tw_lo;    // w_n^(+-i), i < 2^lb                 (inter-pass twiddles; null when L == 1)
  const BTw *tw_hi;    // w_n^(+-i * 2^lb)
  const BTw *pre_lo;   // first pass: multiply input element i by pre_hi[i >> lb] * pre_lo[i & mask] (null: none)
  const BTw *pre_hi;
  const BTw *post_lo;  // last pass: multiply output element k likewise (null: none)
  const BTw *post_hi;
  const BTw *post_const;   // ... or by this single entry (1/n of ifft; null: none)
  // one-level tables in TILE order (null: the two-level tables above; only for full tiles, log_n >= 12): entry
  // t * 2048 + e belongs to element e of tile t in the order the load phase (pre1) / the store phase (tw1: inter-pass
  // twiddle of a non-last pass; post1: last pass) walks the tile, so that consecutive lanes read consecutive entries
  // whatever the tile's shape in memory is (the 2^21 / 2^22 plans have tiles one or two elements wide)
  const BTw *pre1, *tw1, *post1;
  u32 lb;             // low bits of the two-level tables
  u32 log_n;
  u32 s;              // bits consumed by earlier passes
  u32 r;              // radix bits of this pass
  u32 log_c;          // log2(columns per tile)
  u32 is_last;
  u32 r0;             // radix bits of pass 0 (for the last pass' column dimension)
  u32 L;              // number of passes
  u32 r1;             // radix bits of pass 1 (L == 3: storage order of the middle digit)
  u32 xcd_swizzle;    // tiles % 8 == 0: blockIdx -> tile so that neighbouring tiles share an XCD
};

__device__ __forceinline__ fr_t ld_fr(const fr_t *p) {
  fr_t v;
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
  uint4 a = q[0], b = q[1];
  v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w;
  v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w;
  return v;
}
__device__ __forceinline__ void st_fr(fr_t *p, const fr_t &v) {
  uint4 *q = reinterpret_cast<uint4 *>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// ---- lazily reduced Fr: values in [0, 2q) (2q < 2^256) -----------------------------------------------------------
__device__ __forceinline__ constexpr u32 fr_mod2(int i) {   // limb i of 2q
  return (FrParams::mod(i) << 1) | (i ? FrParams::mod(i - 1) >> 31 : 0u);
}
__device__ __forceinline__ void frl_add(fr_t &r, const fr_t &a, const fr_t &b) {   // a + b < 4q: bit 256 is the carry
#ifdef BH_DIAG_CHEAP_ADDSUB   // TIMING-ONLY diagnostic build: carry-free limb-wise additions (wrong results)
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = a.l[i] + b.l[i];
  r.l[0] += a.l[7] >> 3;   // (a ninth limb's worth)
  return;
#endif
  u32 t[8], d[8];
  u32 c = 0, br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = addc(a.l[i], b.l[i], c, c);
#pragma unroll
  for (int i = 0; i < 8; i++) d[i] = subb(t[i], fr_mod2(i), br, br);
  const bool ge = c || !br;   // the 257-bit sum is >= 2q
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = ge ? d[i] : t[i];
}
__device__ __forceinline__ void frl_sub(fr_t &r, const fr_t &a, const fr_t &b) {
#ifdef BH_DIAG_CHEAP_ADDSUB   // ... and subtractions with a per-limb bias
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = a.l[i] + fr_mod2(i) - b.l[i];
  r.l[0] += (a.l[7] + fr_mod2(0) - b.l[7]) >> 3;
  return;
#endif
  u32 t[8];
  u32 br = 0, c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = subb(a.l[i], b.l[i], br, br);
  const u32 mask = 0u - br;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = addc(t[i], fr_mod2(i) & mask, c, c);
}
__device__ __forceinline__ void frl_canon(fr_t &r) {   // [0, 2q) -> [0, q)
  u32 t[8];
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = r.l[i];
  fe_reduce_once<FrParams>(r, t);
}

// ---- the multiplier with a pre-sliced second operand ---------------------------------------------------------
// x * w for a table entry w in B form.  Out of line: a radix-8 step has 12 of these, inlined they would not fit the
// instruction cache.  The entry used to be fetched inside - an L2-latency load in front of EVERY product that the
// two wavefronts of a SIMD could not hide (23 % of the wave time, profiles/archive/r2_call10_pmc_fft.json) - and a load
// issued by the caller before the call would be waited for at the callee's first instruction (every non-kernel
// function starts with s_waitcnt vmcnt(0)).  So the products are chained: a call receives its own entry IN
// REGISTERS and the address of the NEXT product's entry, issues that load first, multiplies while it is in flight
// and hands the loaded entry back with the result.
struct TwReg {   // a table entry in registers: the nine 30-bit limbs
  u32x4 a, b;
  u32 c;
};
typedef u32 u32x20 __attribute__((ext_vector_type(20)));   // 8 result words, 9 entry words, 3 unused
#define BH_GLOBAL_AS __attribute__((address_space(1)))   // global_load, not flat_load: the tables are never in LDS
__device__ __forceinline__ TwReg tw_load(const BTw *w) {
  const BH_GLOBAL_AS u32x4 *q = (const BH_GLOBAL_AS u32x4 *)w;
  TwReg t;
  t.a = q[0]; t.b = q[1]; t.c = ((const BH_GLOBAL_AS u32 *)w)[8];
  return t;
}
// entry `idx` of a one-level table: 32 bytes (48 as pre-sliced limbs: a third more to stream per element and pass for 17
// of a product's 290 instructions - profiles/archive/r5_call1_fft_variants.txt)
struct MReg { u32x4 a, b; };
__device__ __forceinline__ MReg tw_load1(const BTw *tab, u64 idx) {
  const BH_GLOBAL_AS u32x4 *q = (const BH_GLOBAL_AS u32x4 *)((const char *)tab + idx * NTT_ONE_STRIDE);
  MReg t;
  t.a = q[0]; t.b = q[1];
  return t;
}
__device__ __attribute__((noinline)) static u32x20 fr_mul_tw(u32x4 a0, u32x4 a1, u32x4 w0, u32x4 w1, u32 w2, const BTw *next) {
  const TwReg n = tw_load(next);
  fr_t a, r;
  a.l[0] = a0.x; a.l[1] = a0.y; a.l[2] = a0.z; a.l[3] = a0.w;
  a.l[4] = a1.x; a.l[5] = a1.y; a.l[6] = a1.z; a.l[7] = a1.w;
  const u32 B[9] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2};
  fe_mul_b<FrParams, false>(r, a, B);   // lazily reduced: < 1.91 q for a < 2q and a canonical entry
  u32x20 o;
  o[0] = r.l[0]; o[1] = r.l[1]; o[2] = r.l[2]; o[3] = r.l[3]; o[4] = r.l[4]; o[5] = r.l[5]; o[6] = r.l[6]; o[7] = r.l[7];
  o[8] = n.a.x; o[9] = n.a.y; o[10] = n.a.z; o[11] = n.a.w; o[12] = n.b.x; o[13] = n.b.y; o[14] = n.b.z; o[15] = n.b.w;
  o[16] = n.c; o[17] = 0; o[18] = 0; o[19] = 0;
  return o;
}
// the same without the chained load: the entry (a 32-byte Montgomery element of a one-level table) is already in registers
typedef u32 u32x8 __attribute__((ext_vector_type(8)));
__device__ __attribute__((noinline)) static u32x8 fr_mul_m(u32x4 a0, u32x4 a1, u32x4 w0, u32x4 w1) {
  fr_t a, w, r;
  a.l[0] = a0.x; a.l[1] = a0.y; a.l[2] = a0.z; a.l[3] = a0.w;
  a.l[4] = a1.x; a.l[5] = a1.y; a.l[6] = a1.z; a.l[7] = a1.w;
  w.l[0] = w0.x; w.l[1] = w0.y; w.l[2] = w0.z; w.l[3] = w0.w;
  w.l[4] = w1.x; w.l[5] = w1.y; w.l[6] = w1.z; w.l[7] = w1.w;
  fe_mul<FrParams, false>(r, a, w);   // lazily reduced: < 1.91 q for a < 2q and a canonical entry
  return u32x8{r.l[0], r.l[1], r.l[2], r.l[3], r.l[4], r.l[5], r.l[6], r.l[7]};
}
__device__ __forceinline__ void mul_w(fr_t &a, const MReg &w) {
  const u32x8 o = fr_mul_m(u32x4{a.l[0], a.l[1], a.l[2], a.l[3]}, u32x4{a.l[4], a.l[5], a.l[6], a.l[7]}, w.a, w.b);
#pragma unroll
  for (int i = 0; i < 8; i++) a.l[i] = o[i];
}
// a <- a * cur, cur <- *next
__device__ __forceinline__ void mul_tw(fr_t &a, TwReg &cur, const BTw *next) {
  const u32x20 o = fr_mul_tw(u32x4{a.l[0], a.l[1], a.l[2], a.l[3]}, u32x4{a.l[4], a.l[5], a.l[6], a.l[7]}, cur.a, cur.b, cur.c, next);
#pragma unroll
  for (int i = 0; i < 8; i++) a.l[i] = o[i];
  cur.a = u32x4{o[8], o[9], o[10], o[11]};
  cur.b = u32x4{o[12], o[13], o[14], o[15]};
  cur.c = o[16];
}

// ---- LDS tile: element `lin` (= col * R + position) lives in slot lin + lin/8 of two 16-byte planes -------------
__device__ __forceinline__ fr_t tile_ld(const uint4 *p0, const uint4 *p1, u32 lin) {
  const u32 a = lin + (lin >> 3);
  const uint4 x = p0[a], y = p1[a];
  fr_t v;
  v.l[0] = x.x; v.l[1] = x.y; v.l[2] = x.z; v.l[3] = x.w;
  v.l[4] = y.x; v.l[5] = y.y; v.l[6] = y.z; v.l[7] = y.w;
  return v;
}
__device__ __forceinline__ void tile_st(uint4 *p0, uint4 *p1, u32 lin, const fr_t &v) {
  const u32 a = lin + (lin >> 3);
  p0[a] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  p1[a] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// G DIT stages [s, s + G) of every sub-FFT of the tile, in registers.  A task owns the 2^G positions
//   pos_t = hi * 2^(s+G) + t * 2^s + lo   (t < 2^G)   of one column;
// stage s + j pairs t with t + 2^j (bit j of t clear) under the twiddle w_R^(((t mod 2^j) * 2^s + lo) * R / 2^(s+j+1)),
// which is entry ((t mod 2^j) * 2^s + lo) * (1024 >> (s + j)) of the master table w_2048^i.
// FIRST (s == 0): lo == 0, so stage 0 has no multiplication at all and the others only the constants w_4, w_8^k.
// call order of a task's products: stage j ascending, then t ascending over the t with bit j clear
template <int G, bool FIRST>
__device__ __forceinline__ constexpr bool step_has_mul(int j, int t) {
  return !(t & (1 << j)) && !(FIRST && (t & ((1 << j) - 1)) == 0);
}
template <int G, bool FIRST>
__device__ __forceinline__ constexpr int step_next_mul(int j, int t) {   // (j' << 8 | t') of the product after (j, t); -1: none
  for (int jj = j; jj < G; jj++)
    for (int tt = (jj == j ? t + 1 : 0); tt < (1 << G); tt++)
      if (step_has_mul<G, FIRST>(jj, tt)) return (jj << 8) | tt;
  return -1;
}
template <int G, bool FIRST, int TH>
__device__ __forceinline__ void ntt_step(uint4 *p0, uint4 *p1, u32 tid, u32 total, u32 r, u32 s, const BTw *master) {
  constexpr int N = 1 << G;
  const u32 m = 1u << s;
  const u32 ntasks = total >> G;
  const u32 hi_bits = r - s - G;
  for (u32 task = tid; task < ntasks; task += TH) {
    const u32 lo = FIRST ? 0u : (task & (m - 1));
    const u32 rest = task >> s;
    const u32 hi = rest & ((1u << hi_bits) - 1), col = rest >> hi_bits;
    const u32 pos0 = (col << r) + (hi << (s + G)) + lo;
    auto tw_at = [&](int j, int t) { return master + ((u32)(t & ((1 << j) - 1)) * m + lo) * (1024u >> (s + j)); };
    // the task's first entry travels with the LDS reads; each product fetches the next one's (fr_mul_tw)
    constexpr int first_mul = step_next_mul<G, FIRST>(0, -1);
    TwReg cur;
    if (first_mul >= 0) cur = tw_load(tw_at(first_mul >> 8, first_mul & 255));
    fr_t e[N];
#pragma unroll
    for (int t = 0; t < N; t++) e[t] = tile_ld(p0, p1, pos0 + ((u32)t << s));
#pragma unroll
    for (int j = 0; j < G; j++) {
#pragma unroll
      for (int t = 0; t < N; t++) {
        if (t & (1 << j)) continue;
        fr_t y = e[t + (1 << j)];
        if (step_has_mul<G, FIRST>(j, t)) {         // w^0 = 1 (FIRST, t mod 2^j == 0): nothing to multiply
          const int nx = step_next_mul<G, FIRST>(j, t);
          mul_tw(y, cur, nx >= 0 ? tw_at(nx >> 8, nx & 255) : master);
        }
        fr_t u, v;
        frl_add(u, e[t], y);
        frl_sub(v, e[t], y);
        e[t] = u;
        e[t + (1 << j)] = v;
      }
    }
#pragma unroll
    for (int t = 0; t < N; t++) tile_st(p0, p1, pos0 + ((u32)t << s), e[t]);
  }
}

// ---- where is tile t of a pass? ----------------------------------------------------------------------------------
// non-last: element(row j, col c) at base + j*M + c            (M = n >> (s + r))
// last:     element(row j, col c) at base + c*colstride + j    (rows contiguous)
struct TileGeo {
  u64 base, in_row_stride, in_col_stride;
  u64 out_base, out_row_stride, out_col_stride;
  u32 jp0;   // first column's j' (non-last passes)
};
__device__ __forceinline__ TileGeo tile_geo(const NttPass &a, u64 t) {
  const u64 n = (u64)1 << a.log_n;
  u64 base, in_row_stride, in_col_stride;
  u32 jp0 = 0;
  u64 out_base = 0, out_row_stride = 1, out_col_stride = 1;
  if (!a.is_last) {
    const u32 logM = a.log_n - a.s - a.r;
    const u64 tiles_per_block = ((u64)1 << logM) >> a.log_c;
    const u64 kprefix = t / tiles_per_block;
    jp0 = (u32)((t % tiles_per_block) << a.log_c);
    base = (kprefix << (a.log_n - a.s)) + jp0;
    in_row_stride = (u64)1 << logM;
    in_col_stride = 1;
    out_base = base; out_row_stride = in_row_stride; out_col_stride = 1;
  } else if (a.L == 1) {
    base = 0; in_row_stride = 1; in_col_stride = 0;
    out_base = 0; out_row_stride = 1; out_col_stride = 0;
  } else {
    // columns = C consecutive values of the FIRST digit k0; the middle digit (L == 3) is fixed by `mid`
    const u32 groups = (1u << a.r0) >> a.log_c;          // tiles per middle value
    const u64 mid = t / groups;
    const u32 k00 = (u32)(t % groups) << a.log_c;
    const u64 M0 = n >> a.r0;
    base = (u64)k00 * M0 + (mid << a.r);
    in_row_stride = 1; in_col_stride = M0;
    // output index = k0 + R0*(k1 + R1*k2): `mid` is k1 (L == 3), the rows of this pass are the last digit
    out_base = k00 + (mid << a.r0);
    out_row_stride = (u64)1 << (a.log_n - a.r);  // last digit is the most significant
    out_col_stride = 1;
  }

  TileGeo g;
  g.base = base; g.in_row_stride = in_row_stride; g.in_col_stride = in_col_stride;
  g.out_base = out_base; g.out_row_stride = out_row_stride; g.out_col_stride = out_col_stride; g.jp0 = jp0;
  return g;
}

// ONE: the instantiation for sizes with one-level tables (tw1 / pre1 / post1).  Their entries stream from HBM (0.2 GB per
// table and pass), so the chain "multiply while the next entry loads" of the two-level path - whose entries sit in L2 -
// left a third of the wave cycles waiting (profiles/archive/r4_final_pmc_g2_pairs_and_fft.json: SQ_WAIT_ANY 32 %): all eight
// entries of a thread are loaded at once, beside the data, before the first product.
template <bool ONE>
__global__ __launch_bounds__(NttCfg<ONE>::TH, NttCfg<ONE>::WAVES_PER_SIMD) void ntt_pass_kernel(NttPass a) {
  constexpr int TH = NttCfg<ONE>::TH, GMAX = NttCfg<ONE>::GMAX;
  __shared__ uint4 plane0[NTT_PLANE];
  __shared__ uint4 plane1[NTT_PLANE];
  const u32 R = 1u << a.r, C = 1u << a.log_c;
  const u32 tid = threadIdx.x;
  const fr_t *vin = blockIdx.y == 0 ? a.in : a.in_y[blockIdx.y - 1];
  fr_t *vout = blockIdx.y == 0 ? a.out : a.out_y[blockIdx.y - 1];
  const u32 n_mask = (a.log_n >= 32) ? 0xffffffffu : ((1u << a.log_n) - 1);

  // ---- which tile (XCD-aware: blocks b, b+8, b+16, ... run on one XCD and get consecutive tiles) ------------
  u64 t = blockIdx.x;
  if (a.xcd_swizzle) {
    const u32 per_xcd = gridDim.x >> 3;
    t = (u64)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  }
  const TileGeo tg = tile_geo(a, t);
  const u64 base = tg.base, in_row_stride = tg.in_row_stride, in_col_stride = tg.in_col_stride;
  const u32 jp0 = tg.jp0;
  const u64 out_base = tg.out_base, out_row_stride = tg.out_row_stride, out_col_stride = tg.out_col_stride;
  const u64 tile_first = t << NTT_LOG_TILE;   // one-level tables are in TILE order: entry t * 2048 + e belongs to element e of tile t

  // ---- load (bit-reversed rows), optional pre-multiplication ------------------------------
  const u32 total = R << a.log_c;
  const u32 lb_mask = (1u << a.lb) - 1;
  // All of a thread's loads are issued before anything else happens: a loop of "load, wait, multiply, store to LDS"
  // exposed the full memory latency once per element (8 times per tile, a third of the kernel's time).
  constexpr int PER = NTT_TILE / TH;   // elements per thread
  {
    fr_t v[PER];
    u64 gi[PER];
    u32 slot[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
      const u32 e = tid + (u32)i * TH;
      u32 row, col;
      if (!a.is_last) { row = e >> a.log_c; col = e & (C - 1); }   // consecutive lanes -> consecutive columns
      else { col = e >> a.r; row = e & (R - 1); }                  // consecutive lanes -> consecutive rows
      gi[i] = base + row * in_row_stride + col * in_col_stride;
      const u32 rrow = a.r ? (__brev(row) >> (32 - a.r)) : 0;
      slot[i] = (col << a.r) + rrow;
      if (e < total) v[i] = ld_fr(vin + gi[i]);
    }
    if (ONE && a.pre1) {   // one product per element; entry of element e = tid + i * 256: lanes read consecutive entries
      MReg w[PER];
#pragma unroll
      for (int i = 0; i < PER; i++) w[i] = tw_load1(a.pre1, tile_first + tid + i * TH);
#pragma unroll
      for (int i = 0; i < PER; i++) mul_w(v[i], w[i]);
    } else if (!ONE && a.pre_lo) {   // input element g times pre_hi[g >> lb] * pre_lo[g & mask]: one chain of products (fr_mul_tw)
      const BTw *ph[PER], *pl[PER];
#pragma unroll
      for (int i = 0; i < PER; i++) {
        const bool live = tid + (u32)i * TH < total;
        ph[i] = a.pre_hi + (live ? (u32)(gi[i] >> a.lb) : 0u);
        pl[i] = a.pre_lo + (live ? ((u32)gi[i] & lb_mask) : 0u);
      }
      TwReg cur = tw_load(ph[0]);
#pragma unroll
      for (int i = 0; i < PER; i++) {
        if (tid + (u32)i * TH < total) {
          mul_tw(v[i], cur, pl[i]);
          mul_tw(v[i], cur, i + 1 < PER ? ph[i + 1] : a.pre_hi);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < PER; i++) {
      if (tid + (u32)i * TH < total) tile_st(plane0, plane1, slot[i], v[i]);
    }
  }
  // ---- the r DIT stages: radix-8 steps, then what is left (4 = 2 + 2 rather than 3 + 1) ---------------------
  {
    u32 s = 0, left = a.r;
    bool first = true, prev_local = false;
    u32 prev_g = 0;
    while (left) {
      const u32 g = GMAX == 2 ? (left >= 2 ? 2 : 1) : (left == 4) ? 2 : (left >= 3 ? 3 : left);
      // A step whose tasks are one per thread over a full tile keeps every wavefront inside its own block of 2048 / waves
      // consecutive LDS positions as long as the task's span 2^(s+g) fits the block (the top bits of a position are then
      // the top bits of the task index, i.e. the wavefront): between two such steps of the same shape nothing crosses a
      // wavefront, and a wavefront's LDS operations execute in order - no s_barrier, only a compiler fence.
      const bool local = total == (u32)NTT_TILE && (total >> g) == (u32)TH &&
                         (u32)NTT_LOG_TILE >= g + s + (u32)NttCfg<ONE>::LOG_WAVES;
      if (local && prev_local && g == prev_g) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      else __syncthreads();
      if (GMAX == 3 && g == 3) { if (first) ntt_step<3, true, TH>(plane0, plane1, tid, total, a.r, s, a.master); else ntt_step<3, false, TH>(plane0, plane1, tid, total, a.r, s, a.master); }
      else if (g == 2) { if (first) ntt_step<2, true, TH>(plane0, plane1, tid, total, a.r, s, a.master); else ntt_step<2, false, TH>(plane0, plane1, tid, total, a.r, s, a.master); }
      else { if (first) ntt_step<1, true, TH>(plane0, plane1, tid, total, a.r, s, a.master); else ntt_step<1, false, TH>(plane0, plane1, tid, total, a.r, s, a.master); }
      s += g; left -= g; first = false;
      prev_local = local; prev_g = g;
    }
  }
  __syncthreads();
  // ---- store: inter-pass twiddle (non-last) or final scaling + digit-reversed position ------
  // every product first, every store last: stores count in vmcnt and each out-of-line product starts with
  // s_waitcnt vmcnt(0), so a store followed by the next element's product waited for the store to land
  {
    fr_t v[PER];
    u64 gi[PER];
    if constexpr (ONE) {
      const BTw *one_level = !a.is_last ? a.tw1 : a.post1;
      MReg w[PER];
      if (one_level) {
#pragma unroll
        for (int i = 0; i < PER; i++) w[i] = tw_load1(one_level, tile_first + tid + i * TH);
      }
#pragma unroll
      for (int i = 0; i < PER; i++) {
        const u32 e = tid + (u32)i * TH;
        const u32 row = e >> a.log_c, col = e & (C - 1);   // consecutive lanes -> consecutive columns
        v[i] = tile_ld(plane0, plane1, (col << a.r) + row);
        gi[i] = out_base + row * out_row_stride + col * out_col_stride;
      }
      if (one_level) {
#pragma unroll
        for (int i = 0; i < PER; i++) mul_w(v[i], w[i]);
      }
    } else {
    const BTw *ph[PER], *pl[PER];
    // the two-level table a pass multiplies its outputs with: inter-pass twiddles w_n^ex (not the last pass; ex = 0
    // gives hi[0] * lo[0] = 1 * 1, exact in Montgomery form), or the coset / 1/n scaling of the last pass
    const BTw *hi_tab = !a.is_last ? a.tw_hi : a.post_hi, *lo_tab = !a.is_last ? a.tw_lo : a.post_lo;
#pragma unroll
    for (int i = 0; i < PER; i++) {
      const u32 e = tid + (u32)i * TH;
      const bool live = e < total;
      const u32 row = e >> a.log_c, col = e & (C - 1);   // consecutive lanes -> consecutive columns
      if (live) v[i] = tile_ld(plane0, plane1, (col << a.r) + row);
      const u64 g = out_base + row * out_row_stride + col * out_col_stride;
      gi[i] = g;
      u32 ex = 0;
      if (live) ex = !a.is_last ? (u32)((((u64)(jp0 + col) * row) << a.s) & n_mask) : (u32)g;
      ph[i] = hi_tab ? hi_tab + (ex >> a.lb) : nullptr;   // no table: plain fft / the 1/n of ifft below
      pl[i] = lo_tab ? lo_tab + (ex & lb_mask) : nullptr;
    }
    if (hi_tab) {
      TwReg cur = tw_load(ph[0]);
#pragma unroll
      for (int i = 0; i < PER; i++) {
        if (tid + (u32)i * TH < total) {
          mul_tw(v[i], cur, pl[i]);
          mul_tw(v[i], cur, i + 1 < PER ? ph[i + 1] : hi_tab);
        }
      }
    } else if (a.is_last && a.post_const) {
      TwReg cur = tw_load(a.post_const);
#pragma unroll
      for (int i = 0; i < PER; i++) {
        if (tid + (u32)i * TH < total) mul_tw(v[i], cur, a.post_const);
      }
    }
    }   // two-level tables
#pragma unroll
    for (int i = 0; i < PER; i++) {
      const u32 e = tid + (u32)i * TH;
      if (e < total) {
        if (a.is_last) frl_canon(v[i]);   // what leaves the transform is the canonical field element
        st_fr(vout + gi[i], v[i]);
      }
    }
  }
}

// out[i] = scale * g^i  (i < n).  Each thread owns 16 consecutive powers: g^(16 t) from the
// table g^(2^k) (k < 32) then 15 multiplications by g.
struct PowTable {
  fr_t p2[32];  // g^(2^k)
  fr_t scale;
};
__global__ void gen_powers_kernel(fr_t *out, u64 n, PowTable tab, int mul_into) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 i0 = t * 16;
  if (i0 >= n) return;
  fr_t acc = tab.scale;
  for (int k = 0; k < 32; k++)
    if ((i0 >> k) & 1) fe_mul(acc, acc, tab.p2[k]);
  for (u64 i = i0; i < n && i < i0 + 16; i++) {
    if (mul_into) { fr_t v = ld_fr(out + i); fe_mul(v, v, acc); st_fr(out + i, v); }
    else st_fr(out + i, acc);
    fe_mul(acc, acc, tab.p2[0]);
  }
}
// the same powers as B-form table entries
__global__ void gen_btw_kernel(BTw *out, u64 n, PowTable tab) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 i0 = t * 16;
  if (i0 >= n) return;
  fr_t acc = tab.scale;
  for (int k = 0; k < 32; k++)
    if ((i0 >> k) & 1) fe_mul(acc, acc, tab.p2[k]);
  for (u64 i = i0; i < n && i < i0 + 16; i++) {
    BTw w;
    fe_to_bform<FrParams>(w.l, acc);
    uint4 *q = reinterpret_cast<uint4 *>(out + i);
    q[0] = make_uint4(w.l[0], w.l[1], w.l[2], w.l[3]);
    q[1] = make_uint4(w.l[4], w.l[5], w.l[6], w.l[7]);
    q[2] = make_uint4(w.l[8], 0u, 0u, 0u);
    fe_mul(acc, acc, tab.p2[0]);
  }
}

// one-level table of a pass in tile order (ntt_pass_kernel): entry t * 2048 + e = scale * g^x with
//   kind 0  x = the inter-pass twiddle exponent of element e in the STORE phase of a non-last pass (`ex`),
//   kind 1  x = the index of the element the LOAD phase reads as its element e (coset factors g^i of the first pass),
//   kind 2  x = the index the STORE phase of the last pass writes its element e to (7^-i of icoset_fft)
__global__ __launch_bounds__(256) void gen_tile_table_kernel(BTw *out, NttPass a, PowTable tab, int kind) {
  const u64 t = blockIdx.x;
  const TileGeo tg = tile_geo(a, t);
  const u32 R = 1u << a.r, C = 1u << a.log_c;
  const u32 n_mask = (a.log_n >= 32) ? 0xffffffffu : ((1u << a.log_n) - 1);
  for (u32 e = threadIdx.x; e < (u32)NTT_TILE; e += 256) {
    u32 x;
    if (kind == 1) {
      u32 row, col;
      if (!a.is_last) { row = e >> a.log_c; col = e & (C - 1); } else { col = e >> a.r; row = e & (R - 1); }
      x = (u32)(tg.base + row * tg.in_row_stride + col * tg.in_col_stride);
    } else {
      const u32 row = e >> a.log_c, col = e & (C - 1);
      x = kind == 0 ? (u32)((((u64)(tg.jp0 + col) * row) << a.s) & n_mask)
                    : (u32)(tg.out_base + row * tg.out_row_stride + col * tg.out_col_stride);
    }
    fr_t acc = tab.scale;
    for (int k = 0; k < 32; k++)
      if ((x >> k) & 1) fe_mul(acc, acc, tab.p2[k]);
    st_fr(reinterpret_cast<fr_t *>(out) + (t << NTT_LOG_TILE) + e, acc);
  }
}
// ---- element-wise domain ops -----------------------------------------------------------------
__global__ void fr_mul_assign_kernel(fr_t *a, const fr_t *b, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    fr_t x = ld_fr(a + i), y = ld_fr(b + i);
    fe_mul(x, x, y);
    st_fr(a + i, x);
  }
}
__global__ void fr_sub_assign_kernel(fr_t *a, const fr_t *b, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    fr_t x = ld_fr(a + i), y = ld_fr(b + i);
    fe_sub(x, x, y);
    st_fr(a + i, x);
  }
}
__global__ void fr_scale_kernel(fr_t *a, fr_t k, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    fr_t x = ld_fr(a + i);
    fe_mul(x, x, k);
    st_fr(a + i, x);
  }
}
// a = (a*b - c) * zinv : mul_assign + sub_assign + divide_by_z_on_coset in one pass
__global__ void fr_quotient_kernel(fr_t *a, const fr_t *b, const fr_t *c, fr_t zinv, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    fr_t x = ld_fr(a + i), y = ld_fr(b + i), z = ld_fr(c + i);
    fe_mul(x, x, y);
    fe_sub(x, x, z);
    fe_mul(x, x, zinv);
    st_fr(a + i, x);
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static fr_t fr_from_u64_host(u64 v) {
  fr_t c;
  fe_zero(c);
  c.l[0] = (u32)v;
  c.l[1] = (u32)(v >> 32);
  fr_t m;
  fe_to_mont(m, c);
  return m;
}
static fr_t fr_root_of_unity_host() {  // 7^((q-1)/2^32), canonical limbs (ff::PrimeField::ROOT_OF_UNITY)
  static const u32 rou[8] = {0x439f0d2bu, 0x3829971fu, 0x8c2280b9u, 0xb6368350u,
                             0x22c813b4u, 0xd09b6819u, 0xdfe81f20u, 0x16a2a19eu};
  fr_t c, m;
  for (int i = 0; i < 8; i++) c.l[i] = rou[i];
  fe_to_mont(m, c);
  return m;
}
fr_t fr_domain_omega_host(uint32_t log_n) {  // domain.rs:62-66
  fr_t w = fr_root_of_unity_host();
  for (uint32_t i = log_n; i < 32; i++) fe_sqr(w, w);
  return w;
}
static fr_t fr_pow_u64_host(const fr_t &a, u64 e) {
  u32 el[2] = {(u32)e, (u32)(e >> 32)};
  fr_t r;
  fe_pow(r, a, el, 2);
  return r;
}

static fr_t zinv_host(uint32_t log_n) {  // domain.rs:129-140: (7^m - 1)^-1
  fr_t g = fr_from_u64_host(7), z = fr_pow_u64_host(g, (u64)1 << log_n), one, zi;
  fe_one(one);
  fe_sub(z, z, one);
  fe_inv(zi, z);
  return zi;
}

static PowTable make_pow_table(const fr_t &g, const fr_t &scale) {
  PowTable tab;
  tab.p2[0] = g;
  for (int k = 1; k < 32; k++) fe_sqr(tab.p2[k], tab.p2[k - 1]);
  tab.scale = scale;
  return tab;
}
static int launch_gen_powers(fr_t *out, u64 n, const fr_t &g, const fr_t &scale, int mul_into, hipStream_t st) {
  const PowTable tab = make_pow_table(g, scale);
  u64 threads = (n + 15) / 16;
  u32 blocks = (u32)((threads + 255) / 256);
  hipLaunchKernelGGL(gen_powers_kernel, dim3(blocks), dim3(256), 0, st, out, n, tab, mul_into);
  BH_HIP_CHECK(hipGetLastError());
  return BH_OK;
}
// allocates and fills a B-form table of scale * g^i, i < count
static int make_btw_table(BTw **out, u64 count, const fr_t &g, const fr_t &scale, hipStream_t st) {
  BH_HIP_CHECK(hipMalloc((void **)out, count * sizeof(BTw)));
  const PowTable tab = make_pow_table(g, scale);
  const u64 threads = (count + 15) / 16;
  hipLaunchKernelGGL(gen_btw_kernel, dim3((u32)((threads + 255) / 256)), dim3(256), 0, st, *out, count, tab);
  BH_HIP_CHECK(hipGetLastError());
  return BH_OK;
}

void fft_tables_free(FftTables &t) {
  BTw **all[] = {&t.tw_lo[0], &t.tw_lo[1], &t.tw_hi[0], &t.tw_hi[1], &t.coset_lo, &t.coset_hi, &t.icoset_lo, &t.icoset_hi,
                 &t.minv_dev, &t.tw1[0][0], &t.tw1[0][1], &t.tw1[1][0], &t.tw1[1][1], &t.coset1, &t.icoset1};
  for (BTw **p : all) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
}
void fft_master_free(Context &c) {
  for (int d = 0; d < 2; d++) {
    if (c.fft_master[d]) (void)hipFree(c.fft_master[d]);
    c.fft_master[d] = nullptr;
  }
}

// split log_n into 1-3 passes of <= 11 bits, as evenly as possible (largest first)
static void plan_passes(uint32_t log_n, uint32_t *r, uint32_t *L) {
  const uint32_t l = log_n <= (uint32_t)NTT_MAX_R ? 1 : (log_n + NTT_MAX_R - 1) / NTT_MAX_R;
  *L = l;
  const uint32_t q = log_n / l, rem = log_n % l;
  for (uint32_t i = 0; i < l; i++) r[i] = q + (i < rem ? 1 : 0);
}

// the geometry fields of pass p (what tile_geo and the index arithmetic of ntt_pass_kernel read); s = bits consumed by
// the passes before it
static NttPass pass_geometry(uint32_t log_n, const uint32_t *r, uint32_t L, uint32_t p, uint32_t s) {
  NttPass a = NttPass();   // (value-initialised: every pointer null)
  const bool last = (p == L - 1);
  a.log_n = log_n;
  a.s = s;
  a.r = r[p];
  a.is_last = last ? 1 : 0;
  a.r0 = r[0];
  a.L = L;
  a.r1 = r[1];
  // tile columns: as many as fit 2048 elements, bounded by the extent of the column dimension
  uint32_t log_c = NTT_LOG_TILE - r[p];
  if (L == 1) log_c = 0;
  else if (last) { if (log_c > r[0]) log_c = r[0]; }
  else { uint32_t logM = log_n - s - r[p]; if (log_c > logM) log_c = logM; }
  a.log_c = log_c;
  const u64 tiles = ((u64)1 << log_n) >> (r[p] + log_c);
  a.xcd_swizzle = (tiles >= 64 && (tiles & 7) == 0) ? 1 : 0;
  return a;
}

// internal return code of get_tables: `*evict_need` bytes of OTHER sizes' tables have to go first (fft_evict), then call again
constexpr int BH_FFT_EVICT = -1000;
// the complete one-level set of a size: an inter-pass twiddle table per non-last pass and direction, 7^i, 7^-i
static size_t one_level_set_bytes(uint32_t log_n, uint32_t L) { return (size_t)(2 * (L - 1) + 2) * ((size_t)NTT_ONE_STRIDE << log_n); }

static int get_tables(Context &c, uint32_t log_n, bool inverse, bool need_tw, bool need_coset, bool need_icoset,
                      hipStream_t st, FftTables *out, const BTw **master, size_t *evict_need) {
  std::lock_guard<std::mutex> g(c.fft_mu);
  fr_t one;
  fe_one(one);
  const int dir = inverse ? 1 : 0;
  auto it = c.fft_tables.find(log_n);
  FftTables t = it != c.fft_tables.end() ? it->second : FftTables();
  const u64 n = (u64)1 << log_n;
  uint32_t pr[3] = {0, 0, 0}, pL = 1;
  plan_passes(log_n, pr, &pL);
  // One-level tables (n entries each: 0.13 GB at 2^22, 0.5 GB at 2^24) where the size's COMPLETE set fits the cache's
  // budget - so that dropping the other sizes always makes room for whichever of its tables is asked for next, and the
  // kind of tables a size uses never changes while it is cached (the inverse twiddle table carries the 1/n: the two kinds
  // are never mixed within a size).  A size that holds two-level tables (over budget, a failed allocation) keeps them.
  static const bool one_level_on = [] { const char *e = getenv("BELLMAN_HIP_FFT_ONE_LEVEL"); return !(e && *e == '0'); }();
  const bool has_two_level = t.tw_lo[0] || t.tw_lo[1] || t.coset_lo || t.icoset_lo;
  bool one_level = one_level_on && log_n >= 12 && log_n <= 24 && !t.one_level_failed && !has_two_level &&
                   (t.one_level || one_level_set_bytes(log_n, pL) <= c.fft_table_budget);
  const size_t entry_bytes = (size_t)NTT_ONE_STRIDE << log_n;
  const size_t others = c.fft_table_bytes - (it != c.fft_tables.end() ? it->second.bytes : 0);
  if (one_level) {
    size_t need = 0;
    if (need_tw && !t.tw1[dir][0]) need += (size_t)(pL - 1) * entry_bytes;
    if (need_coset && !t.coset1) need += entry_bytes;
    if (need_icoset && !t.icoset1) need += entry_bytes;
    if (need && c.fft_table_bytes + need > c.fft_table_budget) {
      if (others) { *evict_need = need; return BH_FFT_EVICT; }   // (nothing built yet: nothing to undo)
      if (!t.one_level) one_level = false;   // the budget was lowered under this call: two-level tables
      // (a size that already runs on one-level tables completes its set: at most its own set over the new budget)
    }
  }
  // Everything new is built into locals and PUBLISHED into the context only after every allocation and launch of
  // this call succeeded (and the generating stream drained): a failed hipMalloc / launch leaves the cache exactly as
  // it was, so a retry regenerates instead of running a kernel with half a table pair (null *_hi pointer).
  std::vector<BTw *> fresh;   // freed on failure
  size_t fresh_bytes = 0;
  auto fail = [&](int rc) {
    (void)hipStreamSynchronize(st);
    for (BTw *p : fresh) (void)hipFree(p);
    return rc;
  };
  auto make = [&](BTw **dst, u64 count, const fr_t &base, const fr_t &scale) -> int {
    BTw *p = nullptr;
    int rc = make_btw_table(&p, count, base, scale, st);
    if (p) { fresh.push_back(p); fresh_bytes += count * sizeof(BTw); }
    if (rc == BH_OK) *dst = p;
    return rc;
  };
  BTw *new_master = nullptr;
  if (!c.fft_master[dir]) {   // the in-tile twiddles (96 KiB, shared by every size: not part of the budget)
    fr_t w = fr_domain_omega_host(NTT_LOG_TILE);
    if (inverse) fe_inv(w, w);
    int rc = make(&new_master, 1024, w, one);
    if (rc) return fail(rc);
    fresh_bytes -= 1024 * sizeof(BTw);
  }
  if (!t.init) {
    t.lb = (log_n + 1) / 2;
    fr_t nn = fr_from_u64_host(n);
    fe_inv(t.minv, nn);
    t.zinv = zinv_host(log_n);   // ~800 host field products: computed once per domain size, not per proof
    int rc = make(&t.minv_dev, 1, one, t.minv);   // the single entry 1/n
    if (rc) return fail(rc);
    t.init = true;
  }
  if (one_level) {
    const size_t mark = fresh.size(), mark_bytes = fresh_bytes;
    FftTables t1 = t;
    bool ok = true;
    // entry t * 2048 + e of a table belongs to element e of tile t of its pass (gen_tile_table_kernel)
    auto tile_table = [&](BTw **dst, uint32_t p, uint32_t s_bits, int kind, const fr_t &base, const fr_t &scale) {
      BTw *tab_p = nullptr;
      if (hipMalloc((void **)&tab_p, entry_bytes) != hipSuccess) return false;
      fresh.push_back(tab_p);
      fresh_bytes += entry_bytes;
      const NttPass geo = pass_geometry(log_n, pr, pL, p, s_bits);
      hipLaunchKernelGGL(gen_tile_table_kernel, dim3((u32)(n >> NTT_LOG_TILE)), dim3(256), 0, st, tab_p, geo,
                         make_pow_table(base, scale), kind);
      if (hipGetLastError() != hipSuccess) return false;
      *dst = tab_p;
      return true;
    };
    if (need_tw && !t1.tw1[dir][0]) {
      fr_t w = fr_domain_omega_host(log_n);
      if (inverse) fe_inv(w, w);
      uint32_t s_bits = 0;
      for (uint32_t p = 0; p + 1 < pL && ok; p++) {
        ok = tile_table(&t1.tw1[dir][p], p, s_bits, 0, w, (inverse && p == 0) ? t.minv : one);   // the 1/n of ifft / icoset_fft rides here
        s_bits += pr[p];
      }
    }
    if (ok && need_coset && !t1.coset1) ok = tile_table(&t1.coset1, 0, 0, 1, fr_from_u64_host(7), one);
    if (ok && need_icoset && !t1.icoset1) {
      fr_t ginv;
      fe_inv(ginv, fr_from_u64_host(7));
      uint32_t s_last = 0;
      for (uint32_t p = 0; p + 1 < pL; p++) s_last += pr[p];
      ok = tile_table(&t1.icoset1, pL - 1, s_last, 2, ginv, one);
    }
    // (the size counts as running on one-level tables only once it HOLDS one: a call that built none - a cached 1/Z, a
    // single-pass request - must not let later calls skip the budget gate above)
    if (ok) { t = t1; t.one_level = t.one_level || t1.tw1[0][0] || t1.tw1[1][0] || t1.coset1 || t1.icoset1; }
    else {
      (void)hipStreamSynchronize(st);
      (void)hipGetLastError();
      if (t.one_level || others) {
        // The size already runs on one-level tables (its inverse twiddles carry the 1/n): it must not fall back to the
        // two-level kind half way (a forward call whose allocation failed after an inverse call had built its
        // tables ran the one-level kernel with null table pointers) - make room by dropping the other sizes, and if
        // there is nothing to drop the call fails.  A size without tables yet also tries that before it gives up.
        for (BTw *p : fresh) (void)hipFree(p);
        if (others) { *evict_need = (size_t)-1; return BH_FFT_EVICT; }
        return BH_ERR_HIP;
      }
      for (size_t i = mark; i < fresh.size(); i++) (void)hipFree(fresh[i]);
      fresh.resize(mark);
      fresh_bytes = mark_bytes;
      t.one_level_failed = true;   // this size stays on the two-level tables for good
      one_level = false;
    }
  }
  const u64 n_lo = (u64)1 << t.lb, n_hi = (u64)1 << (log_n - t.lb);
  auto two_level = [&](BTw **lo, BTw **hi, const fr_t &base, const fr_t &scale) -> int {
    BTw *l = nullptr, *h = nullptr;
    int rc = make(&l, n_lo, base, scale);
    if (rc) return rc;
    const fr_t step = fr_pow_u64_host(base, n_lo);
    rc = make(&h, n_hi, step, one);
    if (rc) return rc;
    *lo = l; *hi = h;   // a pair is set together or not at all
    return BH_OK;
  };
  if (!one_level && need_tw && !t.tw_lo[dir]) {
    fr_t w = fr_domain_omega_host(log_n);
    if (inverse) fe_inv(w, w);
    int rc = two_level(&t.tw_lo[dir], &t.tw_hi[dir], w, one);
    if (rc) return fail(rc);
  }
  if (!one_level && need_coset && !t.coset_lo) {
    int rc = two_level(&t.coset_lo, &t.coset_hi, fr_from_u64_host(7), one);   // MULTIPLICATIVE_GENERATOR
    if (rc) return fail(rc);
  }
  if (!one_level && need_icoset && !t.icoset_lo) {
    fr_t ginv;
    fe_inv(ginv, fr_from_u64_host(7));
    int rc = two_level(&t.icoset_lo, &t.icoset_hi, ginv, t.minv);   // 7^-i / n
    if (rc) return fail(rc);
  }
  // tables are generated on `st`; later users may be on other streams
  if (!fresh.empty() && hipStreamSynchronize(st) != hipSuccess) return fail(BH_ERR_HIP);
  if (new_master) c.fft_master[dir] = new_master;
  t.bytes += fresh_bytes;
  t.last_use = ++c.fft_tick;
  c.fft_table_bytes += fresh_bytes;
  c.fft_tables[log_n] = t;
  *out = t;
  *master = c.fft_master[dir];
  return BH_OK;
}

// drops the least recently used sizes other than `keep_log_n` until `need` more bytes fit the budget ((size_t)-1: all of
// them).  The caller holds fft_use_mu EXCLUSIVELY: no transform is between its get_tables and its last launch; the device is
// drained before a free.
static void fft_evict_locked(Context &c, uint32_t keep_log_n, size_t need) {
  (void)hipDeviceSynchronize();
  std::lock_guard<std::mutex> g(c.fft_mu);
  for (;;) {
    if (need != (size_t)-1 && c.fft_table_bytes + need <= c.fft_table_budget) break;
    auto victim = c.fft_tables.end();
    for (auto i = c.fft_tables.begin(); i != c.fft_tables.end(); ++i)
      if (i->first != keep_log_n && i->second.bytes && (victim == c.fft_tables.end() || i->second.last_use < victim->second.last_use)) victim = i;
    if (victim == c.fft_tables.end()) break;
    c.fft_table_bytes -= victim->second.bytes;
    fft_tables_free(victim->second);
    c.fft_tables.erase(victim);
  }
}
// What a transform holds on the table cache from get_tables to its last launch: the use lock shared - or, after an eviction,
// exclusively.
struct TableUse {
  std::shared_lock<std::shared_mutex> shared;
  std::unique_lock<std::shared_mutex> exclusive;
};
// get_tables with the cache's eviction; on BH_OK `use` holds fft_use_mu - keep it until the last launch that reads the
// tables has been enqueued.  [r6, ADVICE r5] The eviction path reaches a guaranteed outcome: it takes the use lock
// EXCLUSIVELY and keeps it through the rebuild and the caller's launches, so that no other host thread can take the room
// back between "evicted" and "built" (round 5 retried four times under the shared lock and returned BH_ERR_HIP when two
// threads whose table sets do not fit the budget together kept evicting each other - a spurious failure that the
// resident-domain Rust path turns into a panic).  With every other size gone get_tables never asks for an eviction
// again: it builds the set, completes a set it already runs on, or falls back to two-level tables.
static int acquire_tables(Context &c, uint32_t log_n, bool inverse, bool need_tw, bool need_coset, bool need_icoset, hipStream_t st,
                          FftTables *out, const BTw **master, TableUse &use) {
  use.shared = std::shared_lock<std::shared_mutex>(c.fft_use_mu);
  size_t need = 0;
  int rc = get_tables(c, log_n, inverse, need_tw, need_coset, need_icoset, st, out, master, &need);
  if (rc != BH_FFT_EVICT) { if (rc != BH_OK) use.shared.unlock(); return rc; }
  use.shared.unlock();
  use.exclusive = std::unique_lock<std::shared_mutex>(c.fft_use_mu);
  for (int attempt = 0; attempt < 2 && rc == BH_FFT_EVICT; attempt++) {   // (second round: everything else goes)
    fft_evict_locked(c, log_n, attempt ? (size_t)-1 : need);
    rc = get_tables(c, log_n, inverse, need_tw, need_coset, need_icoset, st, out, master, &need);
  }
  if (rc == BH_FFT_EVICT) rc = BH_ERR_HIP;   // (unreachable: nothing is left to evict)
  if (rc != BH_OK) use.exclusive.unlock();
  return rc;
}

// data: device, 2^log_n Montgomery Fr, in place.  scratch: device, same size (ping-pong for the
// digit-reversing last pass; may be null when log_n <= NTT_MAX_R).
// `more`: up to two further vectors transformed in place by the same launch - only for single-pass sizes (no scratch
// vector to share); a small proof's three iFFTs / coset FFTs are one launch each instead of three latency-bound ones
static int ntt_run_batch(Context &c, fr_t *data, fr_t *scratch, uint32_t log_n, int mode, hipStream_t st, fr_t *const *more, u32 n_more);
int ntt_run(Context &c, fr_t *data, fr_t *scratch, uint32_t log_n, int mode, hipStream_t st) {
  return ntt_run_batch(c, data, scratch, log_n, mode, st, nullptr, 0);
}
static int ntt_run_batch(Context &c, fr_t *data, fr_t *scratch, uint32_t log_n, int mode, hipStream_t st, fr_t *const *more, u32 n_more) {
  if (log_n >= 32) return BH_ERR_DEGREE_TOO_LARGE;
  const bool inverse = (mode == BH_IFFT || mode == BH_ICOSET_FFT);
  uint32_t r[3] = {0, 0, 0}, L = 1;
  plan_passes(log_n, r, &L);
  FftTables tab;
  const BTw *master = nullptr;
  TableUse use;   // the tables stay cached until this call's launches are enqueued
  int rc = acquire_tables(c, log_n, inverse, L > 1, mode == BH_COSET_FFT, mode == BH_ICOSET_FFT, st, &tab, &master, use);
  if (rc) return rc;
  // one kind of tables per size: the one-level kernel multiplies by whatever its table pointers say, so every table
  // this call needs has to be there (get_tables guarantees it; a null pointer here would be a silently wrong transform)
  if (tab.one_level && ((L > 1 && !tab.tw1[inverse ? 1 : 0][0]) || (mode == BH_COSET_FFT && !tab.coset1) ||
                        (mode == BH_ICOSET_FFT && !tab.icoset1)))
    return BH_ERR_HIP;
  uint32_t s = 0;
  for (uint32_t p = 0; p < L; p++) {
    NttPass a = pass_geometry(log_n, r, L, p, s);
    const bool last = (p == L - 1);
    // ping-pong without a copy: pass 0 data -> scratch, middle passes in place in scratch,
    // last pass scratch -> data (digit-reversing scatter)
    a.in = (p == 0) ? data : scratch;
    a.out = (last) ? data : scratch;
    for (u32 y = 0; y < 2; y++) { a.in_y[y] = y < n_more ? more[y] : nullptr; a.out_y[y] = y < n_more ? more[y] : nullptr; }
    a.master = master;
    a.tw_lo = tab.tw_lo[inverse ? 1 : 0];
    a.tw_hi = tab.tw_hi[inverse ? 1 : 0];
    const bool pre = (p == 0 && mode == BH_COSET_FFT), post = (last && mode == BH_ICOSET_FFT);
    a.pre_lo = pre ? tab.coset_lo : nullptr;
    a.pre_hi = pre ? tab.coset_hi : nullptr;
    a.post_lo = post ? tab.icoset_lo : nullptr;
    a.post_hi = post ? tab.icoset_hi : nullptr;
    // one-level tables (L >= 2): the inverse twiddle table of pass 0 carries the 1/n, so ifft has no scaling product
    a.tw1 = (tab.one_level && !last) ? tab.tw1[inverse ? 1 : 0][p] : nullptr;
    a.pre1 = (tab.one_level && pre) ? tab.coset1 : nullptr;
    a.post1 = (tab.one_level && post) ? tab.icoset1 : nullptr;
    a.post_const = (last && mode == BH_IFFT && !tab.one_level) ? tab.minv_dev : nullptr;
    a.lb = tab.lb;
    const u64 tiles = ((u64)1 << log_n) >> (r[p] + a.log_c);
    if (n_more && L != 1) return BH_ERR_INVALID_ARG;
    if (tab.one_level) hipLaunchKernelGGL(ntt_pass_kernel<true>, dim3((u32)tiles, 1 + n_more), dim3(NttCfg<true>::TH), 0, st, a);
    else hipLaunchKernelGGL(ntt_pass_kernel<false>, dim3((u32)tiles, 1 + n_more), dim3(NttCfg<false>::TH), 0, st, a);
    BH_HIP_CHECK(hipGetLastError());
    s += r[p];
  }
  return BH_OK;
}

static u32 ew_blocks(Context &c, u64 n) {
  u64 b = (n + 255) / 256;
  u64 cap = (u64)c.num_cus * 8;
  return (u32)(b < cap ? (b ? b : 1) : cap);
}

int fr_mul_assign(Context &c, fr_t *a, const fr_t *b, u64 n, hipStream_t st) {
  hipLaunchKernelGGL(fr_mul_assign_kernel, dim3(ew_blocks(c, n)), dim3(256), 0, st, a, b, n);
  BH_HIP_CHECK(hipGetLastError());
  return BH_OK;
}
int fr_sub_assign(Context &c, fr_t *a, const fr_t *b, u64 n, hipStream_t st) {
  hipLaunchKernelGGL(fr_sub_assign_kernel, dim3(ew_blocks(c, n)), dim3(256), 0, st, a, b, n);
  BH_HIP_CHECK(hipGetLastError());
  return BH_OK;
}
static int cached_zinv(Context &c, uint32_t log_n, hipStream_t st, fr_t *out) {
  FftTables tab;
  const BTw *master = nullptr;
  TableUse use;
  int rc = acquire_tables(c, log_n, false, false, false, false, st, &tab, &master, use);
  if (rc) return rc;
  *out = tab.zinv;
  return BH_OK;
}
int fr_divide_by_z(Context &c, fr_t *a, uint32_t log_n, hipStream_t st) {
  u64 n = (u64)1 << log_n;
  fr_t zinv;
  int rc = cached_zinv(c, log_n, st, &zinv);
  if (rc) return rc;
  hipLaunchKernelGGL(fr_scale_kernel, dim3(ew_blocks(c, n)), dim3(256), 0, st, a, zinv, n);
  BH_HIP_CHECK(hipGetLastError());
  return BH_OK;
}
int fr_distribute_powers(Context &c, fr_t *a, u64 n, const fr_t &g, hipStream_t st) {
  fr_t one;
  fe_one(one);
  return launch_gen_powers(a, n, g, one, 1, st);
}
// out[i] = scale * g^i (generator.rs:249-263 powers of tau, with the h-query factor folded in)
int fr_gen_powers(Context &c, fr_t *out, u64 n, const fr_t &g, const fr_t &scale, hipStream_t st) {
  (void)c;
  if (!n) return BH_OK;
  return launch_gen_powers(out, n, g, scale, 0, st);
}
// prover.rs:221-240 on device-resident, already padded a,b,c; result (m entries) in a
int h_poly_dev(Context &c, fr_t *a, fr_t *b, fr_t *cc, fr_t *scratch, uint32_t log_n, hipStream_t st) {
  int rc;
  fr_t *v[3] = {a, b, cc};
  if (log_n <= NTT_MAX_R) {   // single-pass sizes: a, b and c in one launch per transform
    if ((rc = ntt_run_batch(c, a, scratch, log_n, BH_IFFT, st, v + 1, 2))) return rc;
    if ((rc = ntt_run_batch(c, a, scratch, log_n, BH_COSET_FFT, st, v + 1, 2))) return rc;
  } else {
    for (int i = 0; i < 3; i++) {
      if ((rc = ntt_run(c, v[i], scratch, log_n, BH_IFFT, st))) return rc;
      if ((rc = ntt_run(c, v[i], scratch, log_n, BH_COSET_FFT, st))) return rc;
    }
  }
  u64 n = (u64)1 << log_n;
  fr_t zinv;
  if ((rc = cached_zinv(c, log_n, st, &zinv))) return rc;
  hipLaunchKernelGGL(fr_quotient_kernel, dim3(ew_blocks(c, n)), dim3(256), 0, st, a, b, cc, zinv, n);
  BH_HIP_CHECK(hipGetLastError());
  return ntt_run(c, a, scratch, log_n, BH_ICOSET_FFT, st);
}

}  // namespace bh
