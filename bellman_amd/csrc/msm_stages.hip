// Pippenger bucket multi-scalar multiplication over BLS12-381 G1 / G2 for gfx950.
//
// Replaces bellman's `multiexp` / `multiexp_inner` (src/multiexp.rs:210-332).  What is kept
// is the CONTRACT (Appendix A items 1-9 of SURVEY.md): the sum over dense entries of
// s_i * B[skip + rank_i], the Source/QueryDensity error semantics, asynchronous issue with a
// waiter.  What is not kept is the CPU schedule (one rayon task per window, serial bucket
// fill): the result is a group element, so the order of additions is unobservable.
//
// Device pipeline (all on the job's stream, no host round trip until the very end):
//   1. digits      one thread per scalar: density rank -> base index; the scalar is recoded into
//                  W = ceil(256/c) SIGNED c-bit digits (carry between windows) and written as
//                  (|digit| << 32 | sign << 31 | base) pairs, window-major.       msm_stages.hip
//   2. sort        stable LSD radix sort of every window's pairs by |digit|, 8 bits per pass;
//                  ranking inside a tile uses wavefront ballots (match-any) + popcounts; the tile is sorted
//                  into LDS and copied out in runs.  One bucket set (window tables): the first pass drops
//                  the zero digits.
//   3. chunks      the sorted stream of every window (zero digits skipped) is cut into equal
//                  chunks of K entries: every lane performs exactly K mixed additions whatever
//                  the bucket-size distribution.
//   4. accumulate  one lane per chunk: gather affine bases (L2 / Infinity-Cache resident: the
//                  96 MiB base table fits the 256 MiB MALL), negate for negative digits and
//                  XYZZ mixed-add; buckets straddling chunk borders are folded by the merge
//                  kernels (serial walk for short runs, one workgroup per long run).  msm_ec.cuh
//   5. reduce      sum_i (i+1)*B_i without a serial running sum: split i = hi*2^l + lo, take row
//                  sums over lo and column sums over hi (shuffle-tree reductions), then per-bit
//                  sums of those vectors and the plain window totals -> W*c points.
//   6. tail        result = sum_w 2^(c*w) (sum_p 2^p U[w][p] + T[w]): a 256-step double-and-add,
//                  inherently serial -> host, 64-bit limbs (host_fp.hpp).
#include <stdlib.h>

#include <algorithm>
#include <cmath>

#include "msm_scalar.cuh"
#include "msm_types.hpp"

namespace bh {

// ============================================================================================
// exclusive scan (u32), in place
// ============================================================================================
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_PER = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_PER;

__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_kernel(u32 *data, u32 *block_sums, u64 n) {
  __shared__ u32 wsum[SCAN_THREADS / 64];
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)tid * SCAN_PER;
  u32 v[SCAN_PER];
  u32 run = 0;
#pragma unroll
  for (int k = 0; k < SCAN_PER; k++) {
    u32 t = (base + k < n) ? data[base + k] : 0;
    v[k] = run;
    run += t;
  }
  u32 x = run;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    u32 y = __shfl_up(x, off);
    if (lane >= (u32)off) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  u32 woff = 0;
  for (u32 w = 0; w < wave; w++) woff += wsum[w];
  const u32 toff = woff + x - run;
#pragma unroll
  for (int k = 0; k < SCAN_PER; k++)
    if (base + k < n) data[base + k] = v[k] + toff;
  if (tid == SCAN_THREADS - 1) block_sums[blockIdx.x] = toff + run;
}
__global__ void scan_add_kernel(u32 *data, const u32 *block_off, u64 n) {
  const u64 i = (u64)blockIdx.x * SCAN_TILE + threadIdx.x;
  const u32 off = block_off[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_PER; k++) {
    u64 j = i + (u64)k * SCAN_THREADS;
    if (j < n) data[j] += off;
  }
}
size_t scan_tmp_elems(u64 n) {
  size_t tot = 0;
  while (n > 1) {
    n = (n + SCAN_TILE - 1) / SCAN_TILE;
    tot += (n + 63) & ~(size_t)63;
    if (n == 1) break;
  }
  return tot + 64;
}
static int exclusive_scan_u32(u32 *data, u64 n, u32 *tmp, hipStream_t st) {
  if (n == 0) return BH_OK;
  u64 blocks = (n + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(scan_tile_kernel, dim3((u32)blocks), dim3(SCAN_THREADS), 0, st, data, tmp, n);
  BH_HIP_CHECK(hipGetLastError());
  if (blocks > 1) {
    int rc = exclusive_scan_u32(tmp, blocks, tmp + ((blocks + 63) & ~(u64)63), st);
    if (rc) return rc;
    hipLaunchKernelGGL(scan_add_kernel, dim3((u32)blocks), dim3(SCAN_THREADS), 0, st, data, tmp, n);
    BH_HIP_CHECK(hipGetLastError());
  }
  return BH_OK;
}

// ============================================================================================
// 1. digits
// ============================================================================================
__global__ void density_popc_kernel(const u64 *words, u32 *out, u64 nwords) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nwords) out[i] = (u32)__popcll(words[i]);
}

__global__ void msm_digits_kernel(const void *scalars, int fmt, u32 n, const u64 *density,
                                  const u32 *word_prefix, u64 skip, u64 n_bases, u32 c, u32 W,
                                  u64 base_stride, u64 *pairs, ErrFlags *err) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool dense = true;
  u64 k = skip + i;
  if (density) {
    const u64 word = density[i >> 6];
    dense = (word >> (i & 63)) & 1;
    k = skip + word_prefix[i >> 6] + __popcll(word & (((u64)1 << (i & 63)) - 1));
  }
  bool live = dense;
  if (dense && k >= n_bases) {   // every dense entry checks EOF first, whatever its scalar
    atomicOr(&err->eof, 1u);
    live = false;
  }
  fr_t s;
  if (live) load_scalar(scalars, i, fmt, s);
  // signed-digit recoding, low to high with carry; pair = |d| << 32 | sign << 31 | base index
  const u32 half = 1u << (c - 1);
  u32 carry = 0;
  for (u32 w = 0; w < W; w++) {
    u32 v = (live ? extract_bits(s, w * c, c) : 0) + carry;
    u32 neg = 0;
    carry = 0;
    if (v > half) { v = (1u << c) - v; neg = (v != 0); carry = 1; }
    // with a window table digit w of base k adds row w of the table: 2^(c*w) P_k
    pairs[(u64)w * n + i] = ((u64)v << 32) | ((u64)neg << 31) | ((u32)(k + w * base_stride) & 0x7fffffffu);
  }
}

// ============================================================================================
// 2. radix sort (per window region of n pairs, keyed on 8 bits of the digit per pass)
// ============================================================================================
constexpr int SORT_THREADS = 256;
constexpr int SORT_ROUNDS = 16;
constexpr int SORT_TILE = SORT_THREADS * SORT_ROUNDS;

// [r6] Plans with ONE bucket set (window tables) drop the zero digits in the first pass instead of sorting them to the front:
// `drop_zeros` - the pass neither counts nor moves entries whose digit is 0; `live_from` (later passes) - only the
// n - *live_from entries the first pass kept exist; the last pass writes them to the END of the array (`out_shift`), so that
// what follows sees exactly what it saw before: *zstart entries to skip, then the sorted non-zero digits.  Boolean-heavy
// scalars are mostly zero digits (12 of the 13 of a 0 / 1 scalar): a 90 %-boolean 2^20-term multiexp sorted 13.6 M entries to
// accumulate 1.8 M of them.
__global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(const u64 *pairs, u32 *counts, u32 n,
                                                                u32 shift, u32 num_tiles, u32 drop_zeros, const u32 *live_from) {
  __shared__ u32 hist[256];
  const u32 tid = threadIdx.x, tile = blockIdx.x, w = blockIdx.y;
  hist[tid] = 0;
  __syncthreads();
  const u64 *src = pairs + (u64)w * n;
  const u32 nn = live_from ? n - *live_from : n;
  if (tile * SORT_TILE >= nn) {   // nothing left for this tile (later passes of a vector that was mostly zero digits)
    counts[((u64)w * 256 + tid) * num_tiles + tile] = 0;
    return;
  }
#pragma unroll 4
  for (int r = 0; r < SORT_ROUNDS; r++) {
    u32 idx = tile * SORT_TILE + r * SORT_THREADS + tid;
    if (idx < nn) {
      const u64 key = src[idx];
      if (!(drop_zeros && (u32)(key >> 32) == 0)) atomicAdd(&hist[(u32)(key >> shift) & 0xff], 1u);
    }
  }
  __syncthreads();
  counts[((u64)w * 256 + tid) * num_tiles + tile] = hist[tid];
  if (drop_zeros && tile == 0 && w == 0 && tid == 0) counts[(u64)gridDim.y * 256 * num_tiles] = 0;   // the scan's total slot
}
// after the first pass's scan (one bucket set): the entries it keeps, as the count of those it drops
__global__ void sort_live_kernel(const u32 *scanned_total, u32 *zstart, u32 n) { zstart[0] = n - *scanned_total; }

// STAGED [r6]: the tile is first sorted into LDS (same stable ballot ranking, positions relative to the tile) and then copied
// out, so that neighbouring lanes write neighbouring entries of a bin's run instead of one 8-byte store per bin and round: a
// timing-only build with coalesced writes put the scattered stores at 25-40 % of the whole sort (0.44 -> 0.33 ms at 2^20,
// 1.80 -> 1.09 at 2^22: profiles/r6_call52_scatter_writes_upper_bound.txt).
template <bool STAGED>
__global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(const u64 *pairs_in, u64 *pairs_out,
                                                                   const u32 *offsets, u32 n, u32 shift,
                                                                   u32 num_tiles, u32 drop_zeros, const u32 *live_from,
                                                                   const u32 *out_shift) {
  __shared__ u32 base[256];                              // next position of every bin: global, or (STAGED) inside the tile
  __shared__ u32 wcnt[SORT_THREADS / 64][256];
  __shared__ u32 gbase[STAGED ? 256 : 1];                // STAGED: global position of a bin's run minus its position in the tile
  __shared__ u32 wsum[STAGED ? SORT_THREADS / 64 : 1];
  __shared__ u64 stage[STAGED ? SORT_TILE : 1];          // STAGED: the tile, sorted (32 KB)
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tile = blockIdx.x, w = blockIdx.y;
  if (live_from && tile * SORT_TILE >= n - *live_from) return;   // (uniform over the workgroup: before any barrier)
  const u32 goff = offsets[((u64)w * 256 + tid) * num_tiles + tile];   // global position (all windows)
  base[tid] = STAGED ? 0u : goff;
#pragma unroll
  for (int v = 0; v < SORT_THREADS / 64; v++) wcnt[v][tid] = 0;
  __syncthreads();
  const u64 *src = pairs_in + (u64)w * n;
  const u32 nn = live_from ? n - *live_from : n;
  if (out_shift) pairs_out += *out_shift;
  const u64 lt_mask = ((u64)1 << lane) - 1;
  // all of the thread's keys are loaded before the ranking rounds: one exposed memory latency per tile instead of one
  // per round (the rounds themselves are ballots, LDS and barriers)
  u64 keys[SORT_ROUNDS];
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; r++) {
    const u32 idx = tile * SORT_TILE + r * SORT_THREADS + tid;
    keys[r] = idx < nn ? src[idx] : 0;
  }
  u32 tile_total = 0;
  if constexpr (STAGED) {
    // the tile's histogram, then its exclusive prefix: where every bin's run starts inside the sorted tile
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; r++) {
      const u32 idx = tile * SORT_TILE + r * SORT_THREADS + tid;
      const u64 key = keys[r];
      if (idx < nn && !(drop_zeros && (u32)(key >> 32) == 0)) atomicAdd(&base[(u32)(key >> shift) & 0xff], 1u);
    }
    __syncthreads();
    const u32 cnt = base[tid];
    u32 x = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const u32 y = __shfl_up(x, off);
      if (lane >= (u32)off) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    u32 woff = 0;
#pragma unroll
    for (int v = 0; v < SORT_THREADS / 64; v++) {
      if ((u32)v < wave) woff += wsum[v];
      tile_total += wsum[v];
    }
    const u32 lstart = woff + x - cnt;
    base[tid] = lstart;
    gbase[tid] = goff - lstart;
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; r++) {
    const u32 idx = tile * SORT_TILE + r * SORT_THREADS + tid;
    const u64 key = keys[r];
    const bool valid = idx < nn && !(drop_zeros && (u32)(key >> 32) == 0);
    const u32 bin = (u32)(key >> shift) & 0xff;
    // wavefront match-any over the 8-bit bin: lanes with equal bins
    u64 mask = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const u64 bal = __ballot((bin >> b) & 1);
      mask &= ((bin >> b) & 1) ? bal : ~bal;
    }
    const u32 rank = (u32)__popcll(mask & lt_mask);
    if (valid && rank == 0) wcnt[wave][bin] = (u32)__popcll(mask);
    __syncthreads();
    if (valid) {
      u32 off = base[bin] + rank;
      for (u32 v = 0; v < wave; v++) off += wcnt[v][bin];
      if constexpr (STAGED) stage[off] = key; else pairs_out[off] = key;
    }
    __syncthreads();
    u32 tot = 0;
#pragma unroll
    for (int v = 0; v < SORT_THREADS / 64; v++) { tot += wcnt[v][tid]; wcnt[v][tid] = 0; }
    base[tid] += tot;
    __syncthreads();
  }
  if constexpr (STAGED) {
    for (u32 j = tid; j < tile_total; j += SORT_THREADS) {
      const u64 key = stage[j];
      pairs_out[gbase[(u32)(key >> shift) & 0xff] + j] = key;
    }
  }
}

// ============================================================================================
// 3. first non-zero digit position per window (zero digits sort to the front and are skipped)
// ============================================================================================
__global__ void window_zero_count_kernel(const u64 *pairs, u32 *zstart, u32 n, u32 W) {
  const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= W) return;
  const u64 *src = pairs + (u64)w * n;
  u32 lo = 0, hi = n;   // first index whose digit != 0
  while (lo < hi) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if ((u32)(src[mid] >> 32) == 0) lo = mid + 1; else hi = mid;
  }
  zstart[w] = lo;
}

static u32 ilog2(u64 v) { u32 r = 0; while (v >>= 1) r++; return r; }

MsmPlan make_plan(u64 n, unsigned forced_c, unsigned forced_chunk, bool g2) {
  MsmPlan p;
  p.n = (u32)n;
  // Window size from the MI355X sweep (profiles/archive/r1_tune_small_sizes.txt, r1_tune_c_K_sweep.txt):
  // sizes whose top window is not a sliver (256 = 32*8 = 16*16, 20*13 leaves 9 bits) avoid a
  // few huge buckets; larger c trades bucket-reduction work against accumulation passes.
  const u32 lg = ilog2(n ? n : 1);
  // re-swept after the field-arithmetic changes (profiles/archive/r1_tune_small_sizes.txt, second table): the
  // cheaper additions moved every boundary down by 2-3 powers of two
  // G2 keeps c = 13 up to 2^17: its bucket reduction is two G2 additions per bucket and throughput-bound, 16 windows
  // of 2^15 buckets cost ~2 ms whatever n is - more than the 25 % extra accumulation of 20 windows below 2^18
  // (profiles/archive/r2_call4_*: 2^16 2.4 vs 3.0 ms, 2^17 3.5 vs 3.8, 2^18 5.4 = 5.4).  From 2^24 terms on the accumulation
  // saved by 13 windows of 20 bits outweighs the 6.8 M-bucket reduction (profiles/archive/r2_call8_c20.txt, 2^24: accumulate
  // 47.6 -> 37.9 ms for +5.2 ms of reduction and a third sort pass; 2^23: 28.6 vs 29.4 ms, no gain).
  int c = g2 ? (lg <= 12 ? 8 : lg <= 17 ? 13 : 16) : (lg <= 11 ? 8 : lg <= 14 ? 13 : lg <= 23 ? 16 : 20);
  if (forced_c) c = (int)std::min(24u, std::max(2u, forced_c));
  p.c = (u32)c;
  // Signed c-bit digits d in [-(2^(c-1)-1), 2^(c-1)]: bucket index |d|-1 < 2^(c-1), the sign is
  // applied to the base (y -> -y) when it is loaded.  ceil(256/c) windows always leave room for
  // the final carry (the top window holds < 2^(c-1) because scalars are < 2^255).
  p.W = (256 + p.c - 1) / p.c;
  p.nb = 1u << (p.c - 1);
  p.NB = p.W * p.nb;
  p.lo_bits = (p.c - 1) / 2;
  p.hi_bits = (p.c - 1) - p.lo_bits;
  p.num_tiles = (p.n + SORT_TILE - 1) / SORT_TILE;
  // K entries per lane: 32 once the chip is full, fewer for small problems so that the serial chain
  // per lane shrinks instead of leaving SIMDs idle (same sweep)
  p.chunk = forced_chunk ? forced_chunk
                         : g2 ? (lg <= 11 ? 8 : lg <= 15 ? 16 : lg <= 17 ? 32 : 64)   // G2: fewer, costlier partials
                              : (lg <= 11 ? 8 : lg <= 17 ? 16 : lg <= 19 ? 32 : 64);   // [r5] 2^20: 64 (below)
  // [r5] G1 at 2^20 terms: K = 64 instead of 32 - 2^18 chunks instead of 2^19, i.e. half the head / tail partials the merge
  // launch folds (merge + reduce 0.86 -> 0.69 ms) for the same accumulation (2.51 vs 2.56 ms): 3.72 -> 3.62 ms of device time,
  // same process, alternating (profiles/archive/r5_call4_glv_and_chunk_ab.txt); the launch is exactly two wavefronts per SIMD
  // never let a typical bucket span many chunks: the chunk merge is serial per bucket
  // (c = 20: twice the average run - every chunk partial is a 192-byte record written, read and merged, and there
  // are 6.8 M buckets to merge into; 2^24: reduce 8.4 -> 6.1 ms)
  if (!forced_chunk) p.chunk = (u32)std::max<u64>(p.chunk, (n >> (p.c - 1)) << (p.c >= 20 ? 1 : 0));
  p.chunks_per_window = (p.n + p.chunk - 1) / p.chunk;
  p.sort_passes = (p.c + 7) / 8;
  p.nd = p.n; p.Wd = p.W; p.base_stride = 0;
  return p;
}

// Window bits a table is built for.  All W = ceil(256/c) digits of a scalar land in ONE bucket set, so the sorted
// stream has W*n entries over 2^(c-1) buckets: c is chosen so that a bucket holds a handful of entries (runs that
// span a few chunks at most), which is 3-4 bits more than the classic plan uses for the same n.
unsigned table_window_bits(u64 n_bases, bool g2) {
  // only window sizes whose top window is not a sliver (260 = 20 x 13, 256 = 16 x 16): a sliver puts the top digit of
  // EVERY scalar into a handful of buckets, i.e. runs of n / 8 entries (profiles/archive/r2_call2_sizes_and_table_sweeps.txt:
  // c = 12 and 14 are 1.5-2x slower than 13 at 2^10-2^14)
  const u32 lg = ilog2(n_bases ? n_bases : 1);
  // 260 = 20 x 13, 256 = 16 x 16, 260 = 13 x 20.  G2 stays at 16 above 2^15: its bucket reduction is the expensive
  // part.  13 again around 2^15-2^16, where the 2^15-bucket set of c = 16 is reduced by a launch that no longer
  // fills the chip (profiles/archive/r2_call15_table_bits.txt: G1 2^15 0.79 vs 0.90 ms, 2^16 0.96 vs 1.03; G2 2^15 1.64 vs 1.74)
  // tiny G2 vectors: 32 rows of 8 bits - 128 buckets instead of 4096 mostly empty ones, a shorter reduction chain
  // (profiles/archive/r2_call19_table_bits_tiny.txt: 2^9 0.65 vs 0.80 ms, 2^10 0.72 vs 0.81, 2^11 0.85 vs 0.91; G1: no difference)
  // [r4] re-swept with the lane-pair G2 accumulation and the fused G1 formula (profiles/archive/r4_call13_*): G2 2^15 takes the
  // 8-bit table (1.43 vs 1.53-1.54 ms with 13 / 16 bits); G1 tables now reach 2^18 (16 bits: 1.06 / 1.47 ms at 2^17 / 2^18)
  // [r4, call 16] G1 2^19-2^22 (not built automatically - api.hip auto_table_max_log2 - but on request; the tables of
  // 2^19 points and more are kept at a 128-byte record stride): 16 bits up to 2^20 (wall 2.21 / 3.95 ms against 2.72 / 4.15
  // with 20 bits), 20 bits above (2^21 6.95 vs 7.47 ms with 16 bits; 2^22 12.3 vs 14.2) -
  // profiles/archive/r4_call16_g1_tables_2p19_2p22.txt
  // [r6] re-swept after the merges of the bucket runs moved to lane pairs / one fused launch and "big run" became relative to
  // the average run (profiles/r6_call20_table_bits_final.txt; the two sweeps before it, r6_call16 / r6_call17, were taken with a
  // fixed threshold that sent half of a tiny table's runs down the long path).  The cheaper merges favour MORE partials per
  // bucket: 26 rows of 10 bits for G1 2^11 ... 2^14 (0.44 / 0.46 / 0.50 / 0.57 ms against 0.47 (13 bits) / 0.52 / 0.55 / 0.58
  // with 16), 13 bits for G1 2^15 ... 2^18 (2^16 0.71 against 0.87, 2^17 0.93 against 1.03, 2^18 1.38 against 1.42 with 16);
  // G2 keeps its 8-bit rows up to 2^12 (0.80 / 0.97 against 1.02 / 1.07 with 16), takes 10 bits at 2^13 and is back on 16 bits
  // at 2^15 (1.49 against 1.90 ms with the 8-bit rows round 4 chose for it)
  // [r6, later] 20-bit rows for G1 2^19 ... 2^24 - 13 rows instead of the classic plan's 16 windows, i.e. 19 % fewer additions, into
  // ONE set of 2^19 buckets - became the default once that set's reduction stopped costing what the rows save (two-stage row /
  // column sums, bit sums over the selected half, msm_ec.cuh): 2^19 2.01 ms against 2.40 classic and 2.21 with 16-bit rows,
  // 2^20 3.33 against 3.72-3.81 and 3.88, 2^21 6.12 against 6.90, 2^22 11.3 against 12.9 (profiles/r6_call33_tables_after_sums.txt,
  // r6_call34_table_bits_mid.txt: 13 bits still win up to 2^18 - 1.37 against 1.38 there)
  // G2 keeps its 16-bit rows.  20-bit rows (bh_bases_precompute(.., 20), or BELLMAN_HIP_G2_TABLE20_FROM=<log2 of the first size that
  // takes them>) win on UNIFORM scalars from 2^20 points - 2^20 9.2-9.4 ms against 9.9-10.1, 2^21 16.6 against 19.1 (2^19: 5.7-5.9
  // against 5.4) - but the reduction of their 2^19-bucket set is ~2 ms on lane triples against 0.6 for 2^15 buckets whatever the
  // scalars are, and the b_g2 query of a real proof meets a boolean-heavy witness: 90 % booleans at 2^20 points 3.9 ms against 2.5
  // (profiles/r6_call40_*, r6_call42_*, r6_call45_drop_zeros.txt against r6_final_boolean_mix.txt).  Not the default.
  static const u32 g2_from = [] { const char *e = getenv("BELLMAN_HIP_G2_TABLE20_FROM"); long v = e && *e ? strtol(e, nullptr, 10) : 99; return (u32)(v < 0 ? 0 : v); }();
  // [r6] tiny G2 vectors (up to 2^10 points): 13-bit rows like G1 - 20 n entries over 4096 buckets qualify for the one-launch path
  // (msm_small_fill_kernel: at most 16 entries per bucket on average), which the 8-bit rows' 128 buckets never did: 2^8 0.45 against
  // 0.60 ms, 2^9 0.45 against 0.51, 2^10 0.48 against 0.54 - the b_g2 multiexp was the critical path of a MiMC-322 proof; 2^12: 10 bits
  // (0.70 against 0.78).  (Explicit 10- to 12-bit tables over 2^8 ... 2^9 points used to take that path too and were pathologically slow
  // in it - 2-9 ms, bucket lists overflowing under a sliver top row: profiles/r6_call54_tiny_g2_bits.txt; msm_enqueue now declines
  // the path for such plans, r6_call56_small_path_guard.txt.)
  if (g2) return lg <= 10 ? 13 : lg == 11 ? 8 : lg <= 13 ? 10 : lg >= g2_from ? 20 : 16;
  // (2^25 points and more - never automatic, 94 GB for 2^26 points - take 24-bit rows: 11 of them into 2^23 buckets; 2^25 75.1 ms
  // against 83.5 classic and 82.5 / 77.5 with 20- / 22-bit rows, 2^26 145.1 against 162.0: profiles/r6_call46_g1_tables_2p25_2p26.txt)
  if (lg <= 10) return 13;
  return lg <= 14 ? 10 : lg <= 18 ? 13 : lg <= 24 ? 20 : 24;
}

MsmPlan make_table_plan(u64 n, const WindowTable &t, unsigned forced_chunk, bool g2, int num_cus) {
  MsmPlan p;
  p.c = t.c;
  p.Wd = (256 + p.c - 1) / p.c;            // == t.W
  p.nd = (u32)n;
  p.base_stride = t.stride;
  p.n = (u32)((u64)p.Wd * n);              // every digit of every scalar is an entry of the one window
  p.W = 1;
  p.nb = 1u << (p.c - 1);
  p.NB = p.nb;
  p.lo_bits = (p.c - 1) / 2;
  p.hi_bits = (p.c - 1) - p.lo_bits;
  p.num_tiles = (p.n + SORT_TILE - 1) / SORT_TILE;
  // K: at least the average bucket, so that a typical run touches two chunks (one partial to fold) - shorter chunks
  // turn EVERY bucket into a multi-chunk run and the merge into the dominant cost (profiles/archive/r2_call3_*)
  const u32 lg = ilog2(p.n ? p.n : 1);
  const u32 base_k = lg <= 20 ? 8 : lg <= 22 ? 16 : g2 ? 64 : 32;
  const u64 avg = (u64)p.n >> (p.c - 1);
  // ... but never so long that the chip runs out of lanes (one wavefront per SIMD for the register-heavy
  // one-lane-per-point G2 accumulation, two otherwise); runs of 2-4 chunks are still folded by their owner lane
  // (x BELLMAN_HIP_TABLE_OVERSUB, default 1.  A launch that fills the chip EXACTLY - one wavefront per SIMD for the
  // one-lane-per-point G2 kernel - is the fastest alone, but when other jobs' accumulations hold SIMDs at its start the
  // workgroups that find no slot wait for the first round to END: 10.2 ms instead of 5.5 inside a proof,
  // profiles/archive/r3_call2_proof_timeline.txt.  Shorter chunks for several rounds bound that tail but multiply the partials
  // the merge has to fold - reduce 1.0 -> 2.0 -> 3.1 ms for 2 / 4 rounds, profiles/archive/r3_call3_oversub.txt; the
  // accumulation chain of common.hpp removes the cause instead.)
  static const u64 oversub = [] { const char *e = getenv("BELLMAN_HIP_TABLE_OVERSUB"); long v = e && *e ? strtol(e, nullptr, 10) : 1; return (u64)(v < 1 ? 1 : v > 64 ? 64 : v); }();
  const u64 lanes_min = (u64)num_cus * 4 * 64 * (g2 ? 1 : 2) * oversub;
  u64 k = std::max<u64>(base_k, avg);
  k = std::min<u64>(k, std::max<u64>(base_k, (u64)p.n / lanes_min));
  // [r6] G1 tables with MANY buckets (20-bit rows over 2^19 ... 2^22 points: the average bucket holds 13-104 entries, the chip's
  // lanes 52-416 each): chunks of a whole number of chip-filling rounds instead of several rounds of average-bucket chunks - the
  // accumulation takes the same time (2.29 against 2.30-2.33 ms at 2^20) and the chunk merge folds a quarter of the partials
  // (0.21 -> 0.05 ms; profiles/r6_call29_table20_chunks.txt, r6_call30_*).  BELLMAN_HIP_TABLE_ONE_ROUND=0: the rule above
  static const bool one_round = [] { const char *e = getenv("BELLMAN_HIP_TABLE_ONE_ROUND"); return !(e && *e == '0'); }();
  // ... up to 128 entries per lane; beyond, R launches' worth of equal chunks (one round of 208 / 416 entries ran 8 % slower than
  // four of 52 / 104 at 2^21 / 2^22 - 4.58 against 4.24 ms, 9.33 against 8.59 - for 0.2 ms less merging:
  // profiles/r6_call33_tables_after_sums.txt).  And never ONE round: a launch that fills the chip exactly is as fast as any
  // when it runs alone (2^20: 3.31-3.33 ms with one round, 3.29-3.35 with two), but inside a proof the merge and sum kernels of
  // the job before hold some SIMDs when it starts, and the workgroups that find no slot run as a second round of full-length
  // chunks - the first G1 accumulation of a 2^20-constraint proof took 3.75 ms against 2.3 stand-alone; with two rounds the
  // proof's device part is 18.2-18.4 ms against 19.3-19.6 with one and 20.1 without these tables
  // (profiles/r6_call36_proof_timeline_*.txt, r6_call37_table_rounds.txt).  BELLMAN_HIP_TABLE_ROUNDS=n: at least n rounds
  static const u64 min_rounds = [] { const char *e = getenv("BELLMAN_HIP_TABLE_ROUNDS"); long v = e && *e ? strtol(e, nullptr, 10) : 2; return (u64)(v < 1 ? 1 : v > 16 ? 16 : v); }();
  if (one_round && !g2 && (u64)p.n / lanes_min > k) {
    const u64 per = ((u64)p.n + lanes_min - 1) / lanes_min, rounds = std::max<u64>(min_rounds, (per + 127) / 128);
    k = std::max<u64>(k, ((u64)p.n + lanes_min * rounds - 1) / (lanes_min * rounds));
  }
  // [r6] G2 tables with many buckets (20-bit rows): the accumulation runs on lane PAIRS, 2 x 32 workers per SIMD; whole rounds
  // of at most 64 entries (2^20 points: four rounds of 52 - 6.86 ms against 7.51 with the 64 the rule above gives, which is
  // 3.25 rounds; profiles/r6_call39_g2_table_bits_triples.txt)
  if (one_round && g2 && p.c >= 18 && !forced_chunk) {
    const u64 workers = (u64)num_cus * 4 * 2 * 32;
    const u64 per = ((u64)p.n + workers - 1) / workers, rounds = std::max<u64>(2, (per + 63) / 64);
    k = std::max<u64>(8, ((u64)p.n + workers * rounds - 1) / (workers * rounds));
  }
  p.chunk = forced_chunk ? forced_chunk : (u32)k;
  p.chunks_per_window = (p.n + p.chunk - 1) / p.chunk;
  p.sort_passes = (p.c + 7) / 8;
  return p;
}


int msm_run_stages(const MsmPlan &p, const MsmBuffers &b, const void *scalars_dev, int fmt, const u64 *density_dev,
                   u64 skip, u64 n_bases, hipStream_t st, const u64 **sorted_out) {
  const u64 n = p.nd;   // scalars (the density bitmap is indexed by scalar)
  const u64 ncounts = (u64)p.W * 256 * p.num_tiles;
  const u64 nwords = (n + 63) / 64;
  if (density_dev) {
    hipLaunchKernelGGL(density_popc_kernel, dim3((u32)((nwords + 255) / 256)), dim3(256), 0, st, density_dev,
                       b.word_prefix, nwords);
    BH_HIP_CHECK(hipGetLastError());
    int rc = exclusive_scan_u32(b.word_prefix, nwords, b.scan_tmp, st);
    if (rc) return rc;
  }
  // 1. digits
  hipLaunchKernelGGL(msm_digits_kernel, dim3((p.nd + 255) / 256), dim3(256), 0, st, scalars_dev, fmt, p.nd,
                     density_dev, b.word_prefix, skip, n_bases, p.c, p.Wd, p.base_stride, b.pairs_a, b.err);
  BH_HIP_CHECK(hipGetLastError());
  // 2. sort by digit, 8 bits per pass; one bucket set: zero digits dropped by the first pass (BELLMAN_HIP_SORT_DROP_ZEROS=0:
  // sorted to the front like the classic plan's)
  static const bool drop_on = [] { const char *e = getenv("BELLMAN_HIP_SORT_DROP_ZEROS"); return !(e && *e == '0'); }();
  const bool drop = drop_on && p.W == 1;
  // tiles sorted in LDS and copied out (sort_scatter_kernel<true>).  BELLMAN_HIP_SORT_STAGED=0: one store per entry and round
  static const bool staged_on = [] { const char *e = getenv("BELLMAN_HIP_SORT_STAGED"); return !(e && *e == '0'); }();
  u64 *src = b.pairs_a, *dst = b.pairs_b;
  for (u32 pass = 0; pass < p.sort_passes; pass++) {
    const u32 shift = 32 + 8 * pass;
    const bool first = pass == 0, last = pass + 1 == p.sort_passes;
    const u32 *live_from = (drop && !first) ? b.zstart : nullptr;
    hipLaunchKernelGGL(sort_hist_kernel, dim3(p.num_tiles, p.W), dim3(SORT_THREADS), 0, st, src, b.counts, p.n,
                       shift, p.num_tiles, (drop && first) ? 1u : 0u, live_from);
    BH_HIP_CHECK(hipGetLastError());
    int rc = exclusive_scan_u32(b.counts, ncounts + ((drop && first) ? 1 : 0), b.scan_tmp, st);
    if (rc) return rc;
    if (drop && first) {
      hipLaunchKernelGGL(sort_live_kernel, dim3(1), dim3(1), 0, st, b.counts + ncounts, b.zstart, p.n);
      BH_HIP_CHECK(hipGetLastError());
    }
    if (staged_on)
      hipLaunchKernelGGL(sort_scatter_kernel<true>, dim3(p.num_tiles, p.W), dim3(SORT_THREADS), 0, st, src, dst, b.counts,
                         p.n, shift, p.num_tiles, (drop && first) ? 1u : 0u, live_from, (drop && last) ? b.zstart : nullptr);
    else
      hipLaunchKernelGGL(sort_scatter_kernel<false>, dim3(p.num_tiles, p.W), dim3(SORT_THREADS), 0, st, src, dst, b.counts,
                         p.n, shift, p.num_tiles, (drop && first) ? 1u : 0u, live_from, (drop && last) ? b.zstart : nullptr);
    BH_HIP_CHECK(hipGetLastError());
    std::swap(src, dst);
  }
  *sorted_out = src;
  // 3. where the non-zero digits start in every window
  if (!drop) {
    hipLaunchKernelGGL(window_zero_count_kernel, dim3((p.W + 63) / 64), dim3(64), 0, st, src, b.zstart, p.n, p.W);
    BH_HIP_CHECK(hipGetLastError());
  }
  return BH_OK;
}

}  // namespace bh
