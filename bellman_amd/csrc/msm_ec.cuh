// Curve-dependent part of the MSM pipeline (bucket accumulation, reductions, host tail),
// written once over the field-ops bundle and instantiated for G1 (msm_g1.hip) and G2
// (msm_g2.hip) in separate translation units so they compile in parallel.
#pragma once
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>
#include <array>
#include <cmath>

#include "fp2pair.cuh"
#include "host_fp.hpp"
#include "msm_scalar.cuh"
#include "msm_types.hpp"

namespace bh {

// ---------------------------------------------------------------------------------------------------------
// Logical workers.  A kernel below is written for "workers": one thread when a lane holds a whole group
// element (F::LANES == 1: G1, single-lane G2), one lane TRIPLE for the K3 form of G2 (fp2k3.cuh), where a
// wavefront carries PER_WAVE triples - 21, or 16 where shuffle trees want a power of two - or one lane PAIR
// (fp2pair.cuh, 32 per wavefront).  All lanes of a worker see the same worker index and take the same branches
// (every predicate of the curve code is uniform over the worker's lanes).
// ---------------------------------------------------------------------------------------------------------
template <class F>
constexpr u32 default_per_wave() { return F::LANES == 3 ? 21u : F::LANES == 2 ? 32u : 64u; }
template <class F>
constexpr u32 tree_per_wave() { return F::LANES == 3 ? 16u : F::LANES == 2 ? 32u : 64u; }
template <class F>
constexpr u32 workers_per_block(u32 threads, u32 per_wave) { return F::LANES == 1 ? threads : (threads / 64u) * per_wave; }

// which of its worker's lanes a thread is (0 = the lane that speaks for the worker)
template <class F>
__device__ __forceinline__ u32 worker_role() {
  if constexpr (F::LANES == 1) return 0u;
  else if constexpr (F::LANES == 2) return pair_role();
  else return k3_role();
}
// false for lanes that carry no worker (lane 63 of a K3 wavefront, lanes beyond PER_WAVE triples)
template <class F>
__device__ __forceinline__ bool worker_index(u32 per_wave, u32 &in_block, u32 &global) {
  if constexpr (F::LANES == 1) {
    in_block = threadIdx.x;
    global = blockIdx.x * blockDim.x + threadIdx.x;
    return true;
  } else {
    const u32 t = F::LANES == 3 ? k3_triple(threadIdx.x & 63u) : (threadIdx.x & 63u) >> 1;
    const u32 wave = threadIdx.x >> 6, waves_per_block = blockDim.x >> 6;
    in_block = wave * per_wave + t;
    global = (blockIdx.x * waves_per_block + wave) * per_wave + t;
    return t < per_wave;
  }
}

// Only launched when both EOF and an identity were seen: decides which error the reference
// would report (the highest window's, i.e. the first failing element of the top window;
// multiexp.rs:295-300).  c_ref = reference window size, top window = bits [lo_ref, 256).
template <class F>
__global__ void msm_err_resolve_kernel(const void *scalars, int fmt, u32 n, const u64 *density,
                                       const u32 *word_prefix, u64 skip, u64 n_bases,
                                       const Affine<F> *bases, u32 base_stride, u32 lo_ref, ErrFlags *err) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 k = skip + i;
  if (density) {
    const u64 word = density[i >> 6];
    if (!((word >> (i & 63)) & 1)) return;
    k = skip + word_prefix[i >> 6] + __popcll(word & (((u64)1 << (i & 63)) - 1));
  }
  if (k >= n_bases) return;   // at/after the first EOF entry (k is monotone in i)
  fr_t s;
  load_scalar(scalars, i, fmt, s);
  bool top_nonzero = false;
  for (u32 b = lo_ref; b < 256; b += 16) top_nonzero |= extract_bits(s, b, (256 - b) < 16 ? (256 - b) : 16) != 0;
  if (!top_nonzero) return;
  // (the records the bucket accumulation of this job read: `base_stride` bytes apart)
  if (aff_is_identity(*reinterpret_cast<const Affine<F> *>(reinterpret_cast<const char *>(bases) + (size_t)k * base_stride)))
    atomicOr(&err->ident_top, 1u);
}

// ============================================================================================
// 4. bucket accumulation over equal chunks of the sorted stream
// ============================================================================================
// Lane l of window w owns sorted entries [z_w + l*K, z_w + (l+1)*K).  Buckets that lie completely
// inside the chunk are written straight to pts[bucket]; the (at most two) buckets shared with
// the neighbouring chunks go to head[l] / tail[l] and are folded by msm_merge_chunks_kernel.
struct ChunkView {
  u32 begin, end;       // entry range inside the window
  u32 d_first, d_last;  // digits of the first / last entry
  bool head_partial;    // first bucket started in an earlier chunk
  bool tail_partial;    // last bucket continues in the next chunk
};
// The plan's K is sized for n entries per window, but zero digits (z of them, sorted to the front) are skipped: sparse
// density maps - create_proof's b_g1 / b_g2 queries use about half of the aux variables - and small scalars leave a
// large part of the window empty, and with the plan's K the upper part of the launch would then have nothing to do
// (measured: the 2^19 dense entries of a 2^20-term G2 multiexp ran on HALF of the chip's lanes, 11 ms instead of 5.5).
// Every kernel that walks chunks therefore derives the window's effective chunk length from its zero count, so that
// the n - z live entries are spread over all chunks_per_window lanes again; never below 8 (or the plan's K if smaller)
// so that runs do not shatter into many partials.
__device__ __forceinline__ u32 effective_chunk(u32 n, u32 z, u32 chunks_per_window, u32 K) {
  const u32 live = n - z;
  u32 k = (u32)(((u64)live + chunks_per_window - 1) / chunks_per_window);
  const u32 floor_k = K < 8u ? K : 8u;
  if (k < floor_k) k = floor_k;
  return k < K ? k : K;
}
__device__ __forceinline__ bool chunk_view(const u64 *src, u32 n, u32 z, u32 lane, u32 K, ChunkView &v) {
  const u64 b = (u64)z + (u64)lane * K;
  if (b >= n) return false;
  v.begin = (u32)b;
  v.end = (u32)min((u64)n, b + K);
  v.d_first = (u32)(src[v.begin] >> 32);
  v.d_last = (u32)(src[v.end - 1] >> 32);
  // (never look below z: with one bucket set the sort drops the zero digits and what lies there is not an entry)
  v.head_partial = v.begin > z && (u32)(src[v.begin - 1] >> 32) == v.d_first;
  v.tail_partial = v.end < n && (u32)(src[v.end] >> 32) == v.d_last;
  return true;
}

// LDS_ACC: the running bucket sum lives in the lane's LDS slot instead of 48/96 VGPRs - for G2 this
// is the difference between spilling at one wave per SIMD and fitting two.
// (measured and removed in round 6: a variant that reads one word of the NEXT entry's record right before the iteration's
// last, inline product, so that a window table too big for the Infinity Cache is gathered through the L2 - no difference, 2.88
// against 2.90 ms over a 2 GB table: profiles/r6_call28_touch.txt)
template <class F, bool LDS_ACC>
__global__ __launch_bounds__(128) void msm_accumulate_kernel(const u64 *pairs, const u32 *zstart,
                                                             const Affine<typename F::Mem> *bases,
                                                             XYZZ<typename F::Mem> *pts, XYZZ<typename F::Mem> *head,
                                                             XYZZ<typename F::Mem> *tail, u32 n, u32 c, u32 K,
                                                             u32 chunks_per_window, ErrFlags *err, u32 base_stride) {
  static_assert(!LDS_ACC || F::LANES == 1, "the LDS accumulator belongs to the single-lane kernels");
  static_assert(F::LANES == 1 || F::LANES == 2 || F::LANES == 3, "one lane, a lane pair or a lane triple per group element");
  const u32 w = blockIdx.y;
  u32 in_block, lane;
  if (!worker_index<F>(default_per_wave<F>(), in_block, lane)) return;
  const u64 *src = pairs + (u64)w * n;
  ChunkView v;
  const u32 z = zstart[w];
  K = effective_chunk(n, z, chunks_per_window, K);
  if (lane >= chunks_per_window || !chunk_view(src, n, z, lane, K, v)) return;
  XYZZ<typename F::Mem> *bucket = pts + ((u64)w << (c - 1)) - 1;   // bucket[d], d = |digit| in [1, 2^(c-1)]
  const u64 slot = (u64)w * chunks_per_window + lane;
  __shared__ XYZZ<F> lds_acc[LDS_ACC ? 128 : 1];
  XYZZ<F> reg_acc;
  XYZZ<F> &acc = LDS_ACC ? lds_acc[threadIdx.x] : reg_acc;
  xyzz_set_identity(acc);
  u32 cur = v.d_first;
  bool saw_identity = false;
  u32 adds = 0;   // additions into a non-empty accumulator (the rest of the chunk's entries are copies): bh_msm_wait_stats
  // Software pipeline (one lane per G2 point only: the kernels that fit two wavefronts per SIMD would lose the second
  // one to the extra live registers, and the second wavefront already covers their loads): the entry two steps
  // ahead and the base point one step ahead are loaded by `prefetch`, which xyzz_madd calls right before its last
  // (inline) product - the only stretch of an iteration without an out-of-line call, i.e. without a forced
  // s_waitcnt vmcnt(0) (ec.cuh).  With one resident wavefront per SIMD the two dependent loads of an iteration were
  // 21 % of the kernel's time (profiles/archive/r2_call8_pmc_g2_accumulate.json).
  constexpr bool PIPELINED = F::LANES == 1 && F::WORDS == 24;
  // record `idx` of the vector the accumulation gathers from: every variant reads at the caller's record stride
  auto base_at = [&](u32 idx) {
    return reinterpret_cast<const Affine<typename F::Mem> *>(reinterpret_cast<const char *>(bases) + (size_t)idx * base_stride);
  };
  if constexpr (PIPELINED) {
    u64 e = src[v.begin];
    u64 e1 = v.begin + 1 < v.end ? src[v.begin + 1] : 0;
    Affine<F> q;
    load_affine<F>(q, base_at((u32)e & 0x7fffffffu));
    for (u32 p = v.begin; p < v.end; p++) {
      const u32 d = (u32)(e >> 32);
      if (d != cur) {   // bucket `cur` ends inside this chunk
        store_xyzz<F>((cur == v.d_first && v.head_partial) ? &head[slot] : &bucket[cur], acc);
        xyzz_set_identity(acc);
        cur = d;
      }
      Affine<F> qn = q;
      u64 e2 = 0;
      auto prefetch = [&]() {
        if (p + 1 < v.end) load_affine<F>(qn, base_at((u32)e1 & 0x7fffffffu));
        if (p + 2 < v.end) e2 = src[p + 2];
      };
      if (aff_is_identity(q)) {
        saw_identity = true;
        prefetch();
      } else {
        if ((u32)e >> 31) F::neg(q.y, q.y);   // negative digit: add -P
        adds += xyzz_madd(acc, q, prefetch) ? 1u : 0u;
      }
      q = qn;
      e = e1;
      e1 = e2;
    }
  } else {
    for (u32 p = v.begin; p < v.end; p++) {
      const u64 e = src[p];
      const u32 d = (u32)(e >> 32);
      if (d != cur) {   // bucket `cur` ends inside this chunk
        store_xyzz<F>((cur == v.d_first && v.head_partial) ? &head[slot] : &bucket[cur], acc);
        xyzz_set_identity(acc);
        cur = d;
      }
      Affine<F> q;
      load_affine<F>(q, base_at((u32)e & 0x7fffffffu));
      if (aff_is_identity(q)) { saw_identity = true; continue; }
      if ((u32)e >> 31) F::neg(q.y, q.y);   // negative digit: add -P
      adds += xyzz_madd(acc, q) ? 1u : 0u;
    }
  }
  store_xyzz<F>((cur == v.d_first && v.head_partial) ? &head[slot] : v.tail_partial ? &tail[slot] : &bucket[cur], acc);
  if (saw_identity) atomicOr(&err->ident, 1u);
  // (one address for the whole launch: the compiler's atomic optimizer folds a wavefront's adds into one atomic)
  if (worker_role<F>() == 0) {
    atomicAdd(&err->madds, (unsigned long long)adds);
    if (lane == 0) atomicAdd(&err->zeros, (unsigned long long)z);
  }
}

// ============================================================================================
// 4'. small multiexps over bases with a window table: digits + bucket accumulation in ONE launch, no sort
// ============================================================================================
// create_proof on a small circuit (MiMC-322: multiexps of 322-1023 terms) is bound by the number of launches - a job
// was ~17 of them, ~8 us of host time each.  For up to SMALL_MAX_SCALARS scalars whose Wd digits all go to the one
// bucket set of a window table, every workgroup recodes ALL scalars into signed digits in its LDS (a few scalars per
// thread, redundantly per workgroup) and then each worker owns one bucket and scans the digit table for its entries:
// density prefix, digits, radix sort, zero count, accumulation and the three merge kernels become one launch.
// (profiles/archive/r3_call7_small_fused.txt)
template <class F>
__device__ __forceinline__ void group_reduce_points(XYZZ<F> &acc, u32 G, u32 sub);   // defined with the merge kernels
constexpr u32 SMALL_MAX_SCALARS = 2048, SMALL_MAX_ENTRIES = 20480 + 1024, SMALL_MAX_PER_BUCKET = 16, SMALL_LIST_CAP = 24;
constexpr u32 SMALL_THREADS = 256, SMALL_LANES_PER_BUCKET = 4;
// dynamic LDS of the kernel: base indices, digit table, per-bucket entry lists of the workgroup's buckets
inline size_t small_fill_lds_bytes(u32 nd, u32 n_entries) {
  return (size_t)nd * 4 + (((size_t)n_entries * 2 + 3) & ~size_t(3)) + SMALL_THREADS * 4 + (size_t)SMALL_THREADS * SMALL_LIST_CAP * 2;
}
template <class F>
__global__ __launch_bounds__(SMALL_THREADS) void msm_small_fill_kernel(const void *scalars, int fmt, u32 nd, const u64 *density,
                                                             u32 *word_prefix_out, u64 skip, u64 n_bases, u32 c, u32 Wd,
                                                             u64 base_stride, const Affine<typename F::Mem> *table,
                                                             XYZZ<typename F::Mem> *pts, ErrFlags *err) {
  extern __shared__ u32 small_lds[];
  const u32 n_entries = nd * Wd;
  u32 *kidx = small_lds;                                   // [nd] base index of scalar i, ~0u = contributes nothing
  short *dig = reinterpret_cast<short *>(small_lds + nd);  // [Wd][nd] signed digits
  u32 *cnt = small_lds + nd + ((n_entries * 2 + 3) >> 2);  // [SMALL_THREADS] entries seen per bucket of this workgroup
  unsigned short *lists = reinterpret_cast<unsigned short *>(cnt + SMALL_THREADS);   // [SMALL_THREADS][SMALL_LIST_CAP] entry ids
  const u32 nwords = (nd + 63) / 64;
  cnt[threadIdx.x] = 0;
  // (a) every workgroup recodes all scalars (a few per thread)
  for (u32 i = threadIdx.x; i < nd; i += blockDim.x) {
    bool dense = true;
    u64 k = skip + i;
    if (density) {
      const u64 word = density[i >> 6];
      dense = (word >> (i & 63)) & 1;
      u32 before = 0;
      for (u32 w = 0; w < (i >> 6); w++) before += (u32)__popcll(density[w]);   // <= 32 words
      k = skip + before + __popcll(word & (((u64)1 << (i & 63)) - 1));
    }
    bool live = dense;
    if (dense && k >= n_bases) {   // every dense entry checks EOF first, whatever its scalar (multiexp.rs:55-61,74-80)
      if (blockIdx.x == 0) atomicOr(&err->eof, 1u);
      live = false;
    }
    fr_t s;
    if (live) load_scalar(scalars, i, fmt, s);
    const u32 half = 1u << (c - 1);
    u32 carry = 0;
    for (u32 w = 0; w < Wd; w++) {
      u32 v = (live ? extract_bits(s, w * c, c) : 0) + carry;
      carry = 0;
      int d = (int)v;
      if (v > half) { d = (int)v - (int)(1u << c); carry = 1; }
      dig[w * nd + i] = (short)d;
    }
    kidx[i] = live ? (u32)k : 0xffffffffu;
  }
  if (blockIdx.x == 0 && density && word_prefix_out)   // the (rare) error-resolution pass wants the word prefix
    for (u32 t = threadIdx.x; t <= nwords; t += blockDim.x) {
      u32 before = 0;
      for (u32 w = 0; w < t && w < nwords; w++) before += (u32)__popcll(density[w]);
      word_prefix_out[t] = before;
    }
  __syncthreads();
  // (b) the entries of this workgroup's buckets, one list per bucket (order within a list does not matter)
  constexpr u32 PW = tree_per_wave<F>();                        // workers per wavefront (a power of two)
  const u32 wpb = workers_per_block<F>(SMALL_THREADS, PW);
  const u32 bpb = wpb / SMALL_LANES_PER_BUCKET;                 // buckets per workgroup
  const u32 bucket_lo = blockIdx.x * bpb;   // buckets [bucket_lo, bucket_lo + bpb), bucket index = |digit| - 1
  for (u32 e = threadIdx.x; e < n_entries; e += blockDim.x) {
    const int d = dig[e];
    if (d == 0) continue;
    const u32 bkt = (u32)(d < 0 ? -d : d) - 1;
    if (bkt < bucket_lo || bkt >= bucket_lo + bpb) continue;
    const u32 slot = atomicAdd(&cnt[bkt - bucket_lo], 1u);
    if (slot < SMALL_LIST_CAP) lists[(bkt - bucket_lo) * SMALL_LIST_CAP + slot] = (unsigned short)e;
  }
  __syncthreads();
  // (c) SMALL_LANES_PER_BUCKET workers per bucket: every worker adds every 4th entry of the list, then a shuffle tree.
  // (One worker per bucket made the launch as long as the FULLEST bucket of a wavefront - 12 entries where the average
  // is 5 - at 15 us per addition; the chip is nearly empty during these launches, so lanes are free.)
  u32 in_block, worker;
  const bool has_worker = worker_index<F>(PW, in_block, worker);
  const u32 nb = 1u << (c - 1);
  const u32 local = in_block / SMALL_LANES_PER_BUCKET, sub = in_block % SMALL_LANES_PER_BUCKET;
  const u32 bkt = bucket_lo + local;
  const bool active = has_worker && local < bpb && bkt < nb;
  const u32 total = active ? cnt[local] : 0;
  XYZZ<F> acc;
  xyzz_set_identity(acc);
  bool saw_identity = false;
  auto add_entry = [&](u32 e) {
    const u32 w = e / nd, i = e - w * nd;
    const u32 k = kidx[i];
    if (k == 0xffffffffu) return;   // (cannot happen: such scalars have all-zero digits)
    Affine<F> q;
    load_affine<F>(q, table + ((u64)w * base_stride + k));
    if (aff_is_identity(q)) { saw_identity = true; return; }
    if (dig[e] < 0) F::neg(q.y, q.y);
    xyzz_madd(acc, q);
  };
  if (total <= SMALL_LIST_CAP) {
    for (u32 t = sub; t < total; t += SMALL_LANES_PER_BUCKET) add_entry(lists[local * SMALL_LIST_CAP + t]);
  } else {
    // a bucket fuller than its list (many equal scalars - boolean witnesses put every 1 into bucket 1): scan the table
    const int bucket = (int)bkt + 1;
    u32 seen = 0;
    for (u32 e = 0; e < n_entries; e++) {
      const int d = dig[e];
      if (d == bucket || d == -bucket) {
        if (seen % SMALL_LANES_PER_BUCKET == sub) add_entry(e);
        seen++;
      }
    }
  }
  group_reduce_points<F>(acc, SMALL_LANES_PER_BUCKET, sub);   // every lane of the wavefront takes part in the shuffles
  if (active && sub == 0) store_xyzz<F>(&pts[bkt], acc);
  if (saw_identity) atomicOr(&err->ident, 1u);
}

// Last chunk that holds a head partial of the run with digit d, given that the run continues from chunk `lane`
// into chunk lane + 1: the last chunk whose FIRST entry has digit d.  Galloping + binary search over the chunk
// starts (8-byte probes of the sorted stream): one or two probes for the usual two-chunk run.
__device__ __forceinline__ u32 run_last_chunk(const u64 *src, u32 n, u32 z, u32 K, u32 lane, u32 d) {
  const u32 nchunks = (u32)(((u64)(n - z) + K - 1) / K);
  u32 lo = lane + 1, step = 1, hi;
  for (;;) {
    hi = lo + step;
    if (hi >= nchunks) { hi = nchunks; break; }
    if ((u32)(src[(u64)z + (u64)hi * K] >> 32) != d) break;
    lo = hi;
    step <<= 1;
  }
  while (hi - lo > 1) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if ((u32)(src[(u64)z + (u64)mid * K] >> 32) == d) lo = mid; else hi = mid;
  }
  return lo;
}

// Folds the partial sums of buckets that straddle chunk boundaries.  The lane whose chunk holds the run's first
// entry owns it: it finds the run's last chunk, folds tail[l0] + head[l0+1 .. l1] itself when that is at most
// `walk` additions, and queues longer runs - medium ones for the G-workers-per-run part of msm_merge_tail_kernel, runs of
// more than `big_chunks` chunks cut into workgroup-sized pieces for its long part - so that no lane ever executes a long
// serial chain of point additions (~20 us per link).
// wavefronts per SIMD the register allocation aims at (hipcc reads the second launch bound that way): two for every
// bundle whose lane state allows it - the lane-triple instantiation sat at 256 VGPRs + 2 AGPRs, one wavefront per SIMD
// for two registers (profiles/archive/r4_call8.txt: G2 reduce 1.0 -> 0.91-0.94 ms)
template <class F>
constexpr int merge_waves_per_simd() { return (F::LANES == 1 && F::WORDS == 24) ? 1 : 2; }
template <class F>
__global__ __launch_bounds__(128, merge_waves_per_simd<F>()) void msm_merge_chunks_kernel(const u64 *pairs, const u32 *zstart,
                                                               XYZZ<typename F::Mem> *pts,
                                                               const XYZZ<typename F::Mem> *head,
                                                               const XYZZ<typename F::Mem> *tail, u32 n, u32 c, u32 K,
                                                               u32 chunks_per_window, u32 walk, LongRun *long_runs,
                                                               u32 max_long, BigRun *big_runs, u32 max_big, u32 big_chunks,
                                                               u32 piece, ErrFlags *err) {
  const u32 w = blockIdx.y;
  u32 in_block, lane;
  if (!worker_index<F>(default_per_wave<F>(), in_block, lane)) return;
  const u64 *src = pairs + (u64)w * n;
  const u32 z = zstart[w];
  K = effective_chunk(n, z, chunks_per_window, K);   // the same rule as the accumulation
  ChunkView v;
  if (lane >= chunks_per_window || !chunk_view(src, n, z, lane, K, v)) return;
  if (!v.tail_partial) return;
  if (v.head_partial && v.d_first == v.d_last) return;   // a middle piece of a long bucket
  const u32 d = v.d_last;
  const u32 last = run_last_chunk(src, n, z, K, lane, d);
  if (last - lane > walk) {
    if (worker_role<F>() == 0) {   // one entry per worker
      if (last - lane > big_chunks) {   // cut into workgroup-sized pieces (msm_merge_tail_kernel, long part)
        const u32 slot = atomicAdd(&err->nbig, 1u);
        const u32 np = (last - lane + piece) / piece;   // ceil(partials / piece), partials = last - lane + 1
        const u32 p0 = atomicAdd(&err->npieces, np);
        if (slot < max_big) { BigRun br = {w, lane, d, last, p0, np, 0u, 0u}; big_runs[slot] = br; }
      } else {
        const u32 slot = atomicAdd(&err->nlong, 1u);
        if (slot < max_long) { LongRun lr = {w, lane, d, last}; long_runs[slot] = lr; }
      }
    }
    return;
  }
  const u64 slot0 = (u64)w * chunks_per_window;
  XYZZ<F> acc;
  load_xyzz<F>(acc, tail + slot0 + lane);
  for (u32 j = lane + 1; j <= last; j++) {
    XYZZ<F> o, r;
    load_xyzz<F>(o, head + slot0 + j);
    xyzz_add(r, acc, o);
    acc = r;
  }
  store_xyzz<F>(&pts[((u64)w << (c - 1)) + d - 1], acc);
}

// shuffle-based tree reduction of per-worker points over groups of G consecutive workers of one wavefront;
// `sub` = the worker's index inside its group
template <class F>
__device__ __forceinline__ void group_reduce_points(XYZZ<F> &acc, u32 G, u32 sub) {
  constexpr int NW = sizeof(XYZZ<F>) / 4;
  for (u32 off = G >> 1; off >= 1; off >>= 1) {
    XYZZ<F> o;
    u32 *dst = reinterpret_cast<u32 *>(&o);
    const u32 *srcw = reinterpret_cast<const u32 *>(&acc);
#pragma unroll
    for (int i = 0; i < NW; i++) dst[i] = __shfl_down(srcw[i], off * F::LANES);
    if (sub < off) {
      XYZZ<F> r;
      xyzz_add(r, acc, o);
      acc = r;
    }
  }
}

// ============================================================================================
// 5. reductions: G lanes per output point (serial partial sums, then a shuffle tree)
// ============================================================================================
// the j-th index with bit k set, j = 0, 1, ...  (SUM_BITS walks exactly the selected half of its vector: striding over ALL
// indices left the workers whose own index has bit k clear - k < log2 G - with nothing to add and the others with twice the
// chain; [r6] 13 -> 9 levels for the bit sums of a 2^15-bucket window, 37 -> 21 for a 2^19-bucket set)
__device__ __forceinline__ u32 nth_with_bit(u32 j, u32 k) { return ((j >> k) << (k + 1)) | (1u << k) | (j & ((1u << k) - 1u)); }
// out[g] = sum of a set of in[] points chosen by the mode:
//   SUM_STRIDED: g = (outer, inner): elements in[(outer << group_shift) + inner*istride + t*stride], t < count
//   SUM_BITS   : g = (outer, k):     elements in[(outer << group_shift) + i], i < count, bit k of i set
template <class F>
struct SumJob {
  const XYZZ<typename F::Mem> *in;
  XYZZ<typename F::Mem> *out;
  SumDesc d;
  u32 nblocks;   // one-wavefront blocks assigned to this job
};
// up to three independent reductions per launch (rows + columns, then the bit-sum sets and the
// plain window totals): they are latency-bound, so sharing a launch lets the hardware overlap them.
template <class F>
struct SumJobs {
  SumJob<F> j[3];
};
// A block carries 64 workers: one wavefront of single lanes, or FOUR wavefronts of 16 lane triples (K3) - so that a
// K3 reduction can also put up to 64 workers on one output (the tree then crosses wavefronts through LDS); with 16
// the 256-element column sums of a window table's 2^15 buckets were 16 + 4 additions deep.
template <class F>
constexpr u32 sum_block_threads() { return F::LANES == 3 ? 256u : F::LANES == 2 ? 128u : 64u; }
template <class F>
__global__ __launch_bounds__(sum_block_threads<F>(), F::LANES == 3 ? 2 : 1) void msm_sum_kernel(SumJobs<F> jobs) {
  u32 blk = blockIdx.x;
  u32 which = 0;
  if (blk >= jobs.j[0].nblocks) { blk -= jobs.j[0].nblocks; which = 1; if (blk >= jobs.j[1].nblocks) { blk -= jobs.j[1].nblocks; which = 2; } }
  const SumDesc d = jobs.j[which].d;
  const XYZZ<typename F::Mem> *in = jobs.j[which].in;
  XYZZ<typename F::Mem> *out = jobs.j[which].out;
  // (lane triples: 16 per wavefront where shuffle trees want a power of two - but all 21 when every worker sums alone, the
  // first stage of a two-stage sum [r6]: WPB == 84 then, and the host counts its blocks accordingly)
  constexpr u32 NWAVES = sum_block_threads<F>() / 64;
  const u32 PW = (F::LANES == 3 && d.lanes == 1) ? default_per_wave<F>() : tree_per_wave<F>(), WPB = PW * NWAVES;   // WPB == 64
  u32 t, gid;
  const bool live = worker_index<F>(PW, t, gid);   // t = worker inside the block
  const u32 G = d.lanes;
  const u32 g = (blk * WPB + t) / G;
  const u32 sub = t & (G - 1);
  // single-lane G2: partial sums live in LDS slots (an XYZZ<Fp2> accumulator is 96 VGPRs) and the tree reads
  // the partner's slot directly; G1 and K3-form G2 keep registers + shuffles
  constexpr bool LDS_ACC = (F::LANES == 1 && F::WORDS == 24);
  __shared__ XYZZ<F> lds_acc[LDS_ACC ? 64 : 1];
  __shared__ XYZZ<F> wave_part[NWAVES > 1 ? NWAVES : 1][F::LANES];
  XYZZ<F> reg_acc;
  XYZZ<F> &acc = LDS_ACC ? lds_acc[threadIdx.x] : reg_acc;
  xyzz_set_identity(acc);
  if (live && g < d.groups) {
    const u32 gg = g / d.splits, len = d.count / d.splits, k0 = (g % d.splits) * len;   // (splits == 1: the whole group)
    const u32 outer = gg / d.inner, in_idx = gg % d.inner;
    const XYZZ<typename F::Mem> *base = in + ((u64)outer << d.group_shift);
    if (d.mode == SUM_STRIDED) {
      for (u32 k = k0 + sub; k < k0 + len; k += G) {
        XYZZ<F> o, r;
        load_xyzz<F>(o, base + (u64)in_idx * d.istride + (u64)k * d.stride);
        xyzz_add(r, acc, o);
        acc = r;
      }
    } else {  // SUM_BITS: in_idx = bit position
      for (u32 j = sub; j < (d.count >> 1); j += G) {
        XYZZ<F> o, r;
        load_xyzz<F>(o, base + nth_with_bit(j, in_idx));
        xyzz_add(r, acc, o);
        acc = r;
      }
    }
  }
  if (LDS_ACC) {
    for (u32 off = G >> 1; off >= 1; off >>= 1) {
      __syncthreads();
      if (sub < off) {
        XYZZ<F> r;
        xyzz_add(r, lds_acc[threadIdx.x], lds_acc[threadIdx.x + off]);
        lds_acc[threadIdx.x] = r;
      }
    }
  } else if (NWAVES == 1 || G <= PW) {
    group_reduce_points<F>(acc, G, sub);
  } else {
    // the group spans G / PW wavefronts: tree inside each wavefront, partials through LDS, tree over the partials
    const u32 wave = threadIdx.x >> 6, t_in_wave = t - wave * PW, role = worker_role<F>();
    group_reduce_points<F>(acc, PW, t_in_wave);
    if (live && t_in_wave == 0) wave_part[wave][role] = acc;
    __syncthreads();
    const u32 wpg = G / PW;                       // wavefronts per group (2 or 4); the group's first wavefront folds
    if ((wave & (wpg - 1)) == 0) {
      if (live && t_in_wave < wpg) acc = wave_part[wave + t_in_wave][role]; else xyzz_set_identity(acc);
      group_reduce_points<F>(acc, wpg, t_in_wave);
    }
  }
  if (live && sub == 0 && g < d.groups) store_xyzz<F>(&out[g], acc);
}

// ============================================================================================
// 5'. G1 point additions on lane PAIRS ("K2") for the latency-bound reductions
// ============================================================================================
// A reduction launch of a small or medium job uses a fraction of the chip and is a chain of dependent point additions
// (~14 deep for 4096 buckets), each 8.2 K instructions = 20 us on a wavefront that has its SIMD to itself.  Lanes are free
// there, so a point is split over two neighbouring lanes - the even lane holds (X, ZZ), the odd lane (Y, ZZZ) - and the
// general addition add-2008-s runs as SEVEN lane-local product slots instead of fourteen products:
//     slot   even lane (x side)        odd lane (y side)
//      1     U1 = X1 ZZ2               S1 = Y1 ZZZ2
//      2     U2 = X2 ZZ1               S2 = Y2 ZZZ1            then  P = U2 - U1 | R = S2 - S1
//      3     PP = P^2                  RR = R^2                exchange: x gets RR, y gets PP
//      4     PPP = P PP                ZT = ZZZ1 ZZZ2          exchange: y gets PPP
//      5     Q = U1 PP                 S1P = S1 PPP            x: X3 = RR - PPP - 2Q, D = Q - X3; exchange: y gets D
//      6     T = ZZ1 ZZ2               ZZZ3 = ZT PPP
//      7     ZZ3 = T PP                Y3 = R D - S1P
// (three 12-word DPP exchanges, quad_perm [1,0,3,2]); an addition has about half the latency of the one-lane form, and a
// worker holds 24 words of state instead of 48.  The rare equal-points case falls back to the one-lane doubling,
// computed redundantly by both lanes.  Only the reduction kernels of G1 use it, and only for launches that leave at
// least half of the SIMDs empty (msm_enqueue).
struct HalfPt {
  fp_t u, v;   // even lane: X, ZZ    odd lane: Y, ZZZ
};
__device__ __forceinline__ u32 k2_role() { return k3_lane() & 1u; }
__device__ __forceinline__ fp_t k2_swap(const fp_t &x) {   // the value held by the other lane of my pair
  fp_t r;
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)x.l[i], 0xB1, 0xf, 0xf, false);
  return r;
}
__device__ __forceinline__ bool k2_flag_from(bool mine, u32 want_role) {   // the predicate as the `want_role` lane of my pair sees it
  const u32 lane = k3_lane();
  const u64 m = __ballot(mine);
  return (m >> ((lane & ~1u) | want_role)) & 1;
}
__device__ __forceinline__ fp_t k2_sel(bool odd, const fp_t &even_v, const fp_t &odd_v) {
  fp_t r;
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = odd ? odd_v.l[i] : even_v.l[i];
  return r;
}
__device__ __forceinline__ void k2_load(HalfPt &h, const XYZZ<FpOps> *p) {
  const bool odd = k2_role();
  h.u = odd ? p->y : p->x;
  h.v = odd ? p->zzz : p->zz;
}
__device__ __forceinline__ void k2_store(XYZZ<FpOps> *p, const HalfPt &h) {
  if (k2_role()) { p->y = h.u; p->zzz = h.v; } else { p->x = h.u; p->zz = h.v; }
}
__device__ __forceinline__ void k2_set_identity(HalfPt &h) { fe_zero(h.u); fe_zero(h.v); }
__device__ __forceinline__ bool k2_is_identity(const HalfPt &h) { return k2_flag_from(fpl_is_zero(h.v), 0u); }   // ZZ == 0
// r = a + b; r may alias a.  Every lane of a pair takes the same branches.
__device__ __forceinline__ void k2_add(HalfPt &r, const HalfPt &a, const HalfPt &b) {
  const bool odd = k2_role();
  if (k2_is_identity(a)) { r = b; return; }
  if (k2_is_identity(b)) { r = a; return; }
  fp_t t1 = fp_mul_call(a.u, b.v);          // U1 | S1
  fp_t t2 = fp_mul_call(b.u, a.v);          // U2 | S2
  fp_t d;
  fpl_sub(d, t2, t1);                       // P | R
  const bool dz = fpl_is_zero(d);
  if (k2_flag_from(dz, 0u)) {               // P == 0: the same x coordinate
    if (k2_flag_from(dz, 1u)) {             // ... and R == 0: the same point - one-lane doubling, both lanes redundantly
      XYZZ<FpOps> full, dbl;
      const fp_t ou = k2_swap(a.u), ov = k2_swap(a.v);
      full.x = odd ? ou : a.u; full.y = odd ? a.u : ou;
      full.zz = odd ? ov : a.v; full.zzz = odd ? a.v : ov;
      xyzz_dbl(dbl, full);
      r.u = odd ? dbl.y : dbl.x;
      r.v = odd ? dbl.zzz : dbl.zz;
    } else {
      k2_set_identity(r);                   // opposite points
    }
    return;
  }
  const fp_t sq = fp_sqr_call(d);           // PP | RR
  const fp_t osq = k2_swap(sq);             // x lane: RR    y lane: PP
  const fp_t pp = odd ? osq : sq;           // PP in both lanes
  // slot 4: P PP | ZZZ1 ZZZ2
  const fp_t m4 = fp_mul_call(k2_sel(odd, d, a.v), k2_sel(odd, sq, b.v));
  const fp_t om4 = k2_swap(m4);
  const fp_t ppp = odd ? om4 : m4;          // PPP in both lanes
  // slot 5: U1 PP | S1 PPP
  const fp_t m5 = fp_mul_call(t1, k2_sel(odd, pp, ppp));        // Q | S1P
  fp_t x3, dq;
  fpl_sub(x3, osq, ppp);                    // x lane: RR - PPP (the y lane computes garbage it never uses)
  fpl_sub(x3, x3, m5);
  fpl_sub(x3, x3, m5);                      // X3 = RR - PPP - 2Q
  fpl_sub(dq, m5, x3);                      // Q - X3
  const fp_t odq = k2_swap(dq);             // y lane: Q - X3
  // slot 6: ZZ1 ZZ2 | ZT PPP
  const fp_t m6 = fp_mul_call(k2_sel(odd, a.v, m4), k2_sel(odd, b.v, ppp));   // T | ZZZ3
  // slot 7: T PP | R (Q - X3)
  const fp_t m7 = fp_mul_call(k2_sel(odd, m6, d), k2_sel(odd, pp, odq));      // ZZ3 | R (Q - X3)
  fp_t y3;
  fpl_sub(y3, m7, m5);                      // y lane: Y3 = R (Q - X3) - S1P
  r.u = odd ? y3 : x3;
  r.v = odd ? m6 : m7;
}
// shuffle tree over groups of G consecutive lane pairs of one wavefront (G a power of two <= 32)
__device__ __forceinline__ void k2_group_reduce(HalfPt &acc, u32 G, u32 sub) {
  for (u32 off = G >> 1; off >= 1; off >>= 1) {
    HalfPt o;
#pragma unroll
    for (int i = 0; i < 12; i++) {
      o.u.l[i] = __shfl_down(acc.u.l[i], off * 2);
      o.v.l[i] = __shfl_down(acc.v.l[i], off * 2);
    }
    if (sub < off) k2_add(acc, acc, o);   // pairs take the branch together (sub is a property of the pair)
  }
}
// worker `sub` of the G that share output g adds up its share of the output's elements
__device__ __forceinline__ void k2_partial_sum(HalfPt &acc, const SumDesc &d, const XYZZ<FpOps> *in, u32 g, u32 sub, u32 G) {
  const u32 gg = g / d.splits, len = d.count / d.splits, k0 = (g % d.splits) * len;   // (splits == 1: the whole group)
  const u32 outer = gg / d.inner, in_idx = gg % d.inner;
  const XYZZ<FpOps> *base = in + ((u64)outer << d.group_shift);
  if (d.mode == SUM_STRIDED) {
    for (u32 k = k0 + sub; k < k0 + len; k += G) {
      HalfPt o;
      k2_load(o, base + (u64)in_idx * d.istride + (u64)k * d.stride);
      k2_add(acc, acc, o);
    }
  } else {  // SUM_BITS: in_idx = bit position
    for (u32 j = sub; j < (d.count >> 1); j += G) {
      HalfPt o;
      k2_load(o, base + nth_with_bit(j, in_idx));
      k2_add(acc, acc, o);
    }
  }
}
// msm_sum_kernel for G1 with lane-pair workers: one wavefront = 32 workers per block
template <class FK>   // always FpOps: a template only so that both translation units may see the definition
__global__ __launch_bounds__(64) void msm_sum_k2_kernel(SumJobs<FK> jobs) {
  u32 blk = blockIdx.x;
  u32 which = 0;
  if (blk >= jobs.j[0].nblocks) { blk -= jobs.j[0].nblocks; which = 1; if (blk >= jobs.j[1].nblocks) { blk -= jobs.j[1].nblocks; which = 2; } }
  const SumDesc d = jobs.j[which].d;
  const XYZZ<FpOps> *in = jobs.j[which].in;
  XYZZ<FpOps> *out = jobs.j[which].out;
  const u32 t = threadIdx.x >> 1;           // worker inside the block
  const u32 G = d.lanes;                    // workers per output (<= 32)
  const u32 g = (blk * 32 + t) / G;
  const u32 sub = t & (G - 1);
  HalfPt acc;
  k2_set_identity(acc);
  if (g < d.groups) {
    k2_partial_sum(acc, d, in, g, sub, G);
  }
  k2_group_reduce(acc, G, sub);
  if (sub == 0 && g < d.groups) k2_store(&out[g], acc);
}
// The same with ONE output per workgroup of NW = blockDim.x / 64 <= 4 wavefronts (G = 32 NW workers): tree inside each wavefront,
// the NW partial sums through LDS, tree over them in the first wavefront.  For the handful of long sums a big single bucket
// set ends in (the 20-bit window table of a 2^20-point G1 query: 10 + 9 bit sums and a total over 512 selected points
// each were 16 serial additions + 5 levels on one wavefront; 2 + 5 + 3 here).
template <class FK>
__global__ __launch_bounds__(256) void msm_sum_k2_wide_kernel(SumJobs<FK> jobs) {
  u32 blk = blockIdx.x;
  u32 which = 0;
  if (blk >= jobs.j[0].nblocks) { blk -= jobs.j[0].nblocks; which = 1; if (blk >= jobs.j[1].nblocks) { blk -= jobs.j[1].nblocks; which = 2; } }
  const SumDesc d = jobs.j[which].d;
  const XYZZ<FpOps> *in = jobs.j[which].in;
  XYZZ<FpOps> *out = jobs.j[which].out;
  __shared__ HalfPt wave_part[4][2];
  const u32 NW = blockDim.x >> 6, wave = threadIdx.x >> 6, t_in_wave = (threadIdx.x & 63u) >> 1, role = k2_role();
  const u32 g = blk;                        // one output per workgroup
  HalfPt acc;
  k2_set_identity(acc);
  if (g < d.groups) k2_partial_sum(acc, d, in, g, wave * 32 + t_in_wave, NW * 32);
  k2_group_reduce(acc, 32, t_in_wave);
  if (t_in_wave == 0) wave_part[wave][role] = acc;
  __syncthreads();
  if (wave == 0) {
    if (t_in_wave < NW) acc = wave_part[t_in_wave][role]; else k2_set_identity(acc);
    k2_group_reduce(acc, NW, t_in_wave);
    if (t_in_wave == 0 && g < d.groups) k2_store(&out[g], acc);
  }
}

// ============================================================================================
// 5''. G2 point additions on lane SEXTETS ("K6") for the latency-bound merges [r6]
// ============================================================================================
// The x / y split of K2 on top of the lane triples of K3 (fp2k3.cuh): a G2 point is held by two neighbouring triples -
// the even triple (X, ZZ), the odd triple (Y, ZZZ), each an Fp2 element in (c0, c1, c0 + c1) form - and the general
// addition runs as SEVEN product slots of one lane-local Fp product each instead of fourteen (the table of 5' above,
// with Fp2 products).  A level of a merge tree costs ~20 us instead of the ~37 us of the lane-triple form: what the big
// bucket runs of boolean-heavy G2 queries and the medium runs of small window tables are made of (a 90 %-boolean G2
// query of 2^19 points: 14 levels, 0.58 ms in lane triples - profiles/r6_call5_*).  Eight workers per wavefront (48
// lanes: a power of two for the shuffle trees).  The side exchanges are ds_bpermute moves by +-3 lanes.
constexpr u32 K6_PER_WAVE = 8;
__device__ __forceinline__ u32 k6_lane_in_worker() { const u32 l = k3_lane(); return l - 6u * ((l * 43u) >> 8); }   // lane % 6
__device__ __forceinline__ u32 k6_side() { return k6_lane_in_worker() >= 3u ? 1u : 0u; }   // 0: (X, ZZ)   1: (Y, ZZZ)
__device__ __forceinline__ fp_t k6_swap(const fp_t &x) {   // the same role's value in the other triple of my sextet
  const int src = (int)k3_lane() + (k6_side() ? -3 : 3);
  fp_t r;
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = (u32)__shfl((int)x.l[i], src);
  return r;
}
// the (triple-uniform) predicate as the `want_side` triple of my sextet sees it
__device__ __forceinline__ bool k6_flag_from(bool mine, u32 want_side) {
  const u32 lane = k3_lane();
  const u64 m = __ballot(mine);
  return (m >> (lane - k6_lane_in_worker() + 3u * want_side)) & 1;
}
__device__ __forceinline__ void k6_load(HalfPt &h, const XYZZ<Fp2Ops> *p) {
  const bool y = k6_side();
  Fp2K3Ops::load(h.u, y ? &p->y : &p->x);
  Fp2K3Ops::load(h.v, y ? &p->zzz : &p->zz);
}
__device__ __forceinline__ void k6_store(XYZZ<Fp2Ops> *p, const HalfPt &h) {
  const bool y = k6_side();
  Fp2K3Ops::store(y ? &p->y : &p->x, h.u);
  Fp2K3Ops::store(y ? &p->zzz : &p->zz, h.v);
}
__device__ __forceinline__ bool k6_is_identity(const HalfPt &h) { return k6_flag_from(Fp2K3Ops::is_zero(h.v), 0u); }   // ZZ == 0
// r = a + b; r may alias a.  Every lane of a sextet takes the same branches.
__device__ __forceinline__ void k6_add(HalfPt &r, const HalfPt &a, const HalfPt &b) {
  typedef Fp2K3Ops F;
  const bool y = k6_side();
  if (k6_is_identity(a)) { r = b; return; }
  if (k6_is_identity(b)) { r = a; return; }
  fp_t t1, t2, d;
  F::mul(t1, a.u, b.v);                     // U1 | S1
  F::mul(t2, b.u, a.v);                     // U2 | S2
  F::sub(d, t2, t1);                        // P | R
  const bool dz = F::is_zero(d);
  if (k6_flag_from(dz, 0u)) {               // P == 0: the same x coordinate
    if (k6_flag_from(dz, 1u)) {             // ... and R == 0: the same point - the lane-triple doubling, both triples redundantly
      XYZZ<F> full, dbl;
      const fp_t ou = k6_swap(a.u), ov = k6_swap(a.v);
      full.x = y ? ou : a.u; full.y = y ? a.u : ou;
      full.zz = y ? ov : a.v; full.zzz = y ? a.v : ov;
      xyzz_dbl(dbl, full);
      r.u = y ? dbl.y : dbl.x;
      r.v = y ? dbl.zzz : dbl.zz;
    } else {
      fe_zero(r.u); fe_zero(r.v);           // opposite points
    }
    return;
  }
  fp_t sq, m4, m5, m6, m7;
  F::sqr(sq, d);                            // PP | RR
  const fp_t osq = k6_swap(sq);             // x side: RR    y side: PP
  const fp_t pp = y ? osq : sq;             // PP on both sides
  F::mul(m4, k2_sel(y, d, a.v), k2_sel(y, sq, b.v));        // P PP | ZZZ1 ZZZ2
  const fp_t om4 = k6_swap(m4);
  const fp_t ppp = y ? om4 : m4;            // PPP on both sides
  F::mul(m5, t1, k2_sel(y, pp, ppp));       // Q | S1P
  fp_t x3, dq, y3;
  F::sub(x3, osq, ppp);                     // x side: RR - PPP (the y side computes garbage it never uses)
  F::sub(x3, x3, m5);
  F::sub(x3, x3, m5);                       // X3 = RR - PPP - 2Q
  F::sub(dq, m5, x3);                       // Q - X3
  const fp_t odq = k6_swap(dq);             // y side: Q - X3
  F::mul(m6, k2_sel(y, a.v, m4), k2_sel(y, b.v, ppp));      // T | ZZZ3
  F::mul(m7, k2_sel(y, m6, d), k2_sel(y, pp, odq));         // ZZ3 | R (Q - X3)
  F::sub(y3, m7, m5);                       // y side: Y3 = R (Q - X3) - S1P
  r.u = y ? y3 : x3;
  r.v = y ? m6 : m7;
}
// shuffle tree over groups of G consecutive sextets of one wavefront (G a power of two <= 8)
__device__ __forceinline__ void k6_group_reduce(HalfPt &acc, u32 G, u32 sub) {
  for (u32 off = G >> 1; off >= 1; off >>= 1) {
    HalfPt o;
#pragma unroll
    for (int i = 0; i < 12; i++) {
      o.u.l[i] = __shfl_down(acc.u.l[i], off * 6);
      o.v.l[i] = __shfl_down(acc.v.l[i], off * 6);
    }
    if (sub < off) k6_add(acc, acc, o);   // sextets take the branch together
  }
}

// msm_sum_kernel for G2 with lane-sextet workers: 256 threads = four wavefronts of 8 sextets = 32 workers per block; a
// group of G <= 8 workers folds inside its wavefront, 16 or 32 through LDS.  For the launches that leave the chip mostly
// empty (the bucket sets of small window tables: a MiMC-322 proof's G2 job spent 2 x 136 us here in lane triples).
template <class FK>   // always Fp2K3Ops: a template only so that both translation units may see the definition
__global__ __launch_bounds__(256) void msm_sum_k6_kernel(SumJobs<FK> jobs) {
  u32 blk = blockIdx.x;
  u32 which = 0;
  if (blk >= jobs.j[0].nblocks) { blk -= jobs.j[0].nblocks; which = 1; if (blk >= jobs.j[1].nblocks) { blk -= jobs.j[1].nblocks; which = 2; } }
  const SumDesc d = jobs.j[which].d;
  const XYZZ<Fp2Ops> *in = jobs.j[which].in;
  XYZZ<Fp2Ops> *out = jobs.j[which].out;
  constexpr u32 PW = K6_PER_WAVE, WPB = 4 * PW;   // 32 workers per block
  __shared__ HalfPt wave_part[4][6];
  const u32 lane = k3_lane(), t_in_wave = (lane * 43u) >> 8, wave = threadIdx.x >> 6, role = k6_lane_in_worker();
  const bool live = t_in_wave < PW;
  const u32 t = wave * PW + t_in_wave;      // worker inside the block
  const u32 G = d.lanes;                    // workers per output (a power of two <= 32)
  const u32 g = (blk * WPB + t) / G;
  const u32 sub = t & (G - 1);
  HalfPt acc;
  fe_zero(acc.u); fe_zero(acc.v);
  if (live && g < d.groups) {
    const u32 gg = g / d.splits, len = d.count / d.splits, k0 = (g % d.splits) * len;   // (splits == 1: the whole group)
    const u32 outer = gg / d.inner, in_idx = gg % d.inner;
    const XYZZ<Fp2Ops> *base = in + ((u64)outer << d.group_shift);
    if (d.mode == SUM_STRIDED) {
      for (u32 k = k0 + sub; k < k0 + len; k += G) {
        HalfPt o;
        k6_load(o, base + (u64)in_idx * d.istride + (u64)k * d.stride);
        k6_add(acc, acc, o);
      }
    } else {  // SUM_BITS: in_idx = bit position
      for (u32 j = sub; j < (d.count >> 1); j += G) {
        HalfPt o;
        k6_load(o, base + nth_with_bit(j, in_idx));
        k6_add(acc, acc, o);
      }
    }
  }
  if (G <= PW) {
    k6_group_reduce(acc, G, sub);
  } else {
    // the group spans G / 8 wavefronts: tree inside each wavefront, partials through LDS, tree over the partials
    k6_group_reduce(acc, PW, t_in_wave);
    if (live && t_in_wave == 0) wave_part[wave][role] = acc;
    __syncthreads();
    const u32 wpg = G / PW;                      // wavefronts per group (2 or 4); the group's first wavefront folds
    if ((wave & (wpg - 1)) == 0) {
      if (live && t_in_wave < wpg) acc = wave_part[wave + t_in_wave][role]; else { fe_zero(acc.u); fe_zero(acc.v); }
      k6_group_reduce(acc, wpg, t_in_wave);
    }
  }
  if (live && sub == 0 && g < d.groups) k6_store(&out[g], acc);
}

// ============================================================================================
// 4''. very long bucket runs: workgroup-sized pieces, the last workgroup to finish folds the piece results
// ============================================================================================
// What a run of L partials costs is the DEPTH of its addition tree times the latency of a point addition (~19 us on a
// wavefront that has its SIMD to itself, ~10 us on lane pairs), not the number of additions: round 5 gave every such run
// ONE workgroup of 512 workers - ceil(L / 512) serial additions + 9 tree levels at two wavefronts per SIMD (37 us a level):
// 0.59 ms for the 5 461 partials of bucket 1 of a 50 %-boolean 2^20-term multiexp, 7.5 ms for an all-ones vector over a
// window table, 4.2 ms for a 90 %-boolean G2 query of 2^19 points (profiles/r6_call2_boolean_kernels.txt).  Now a run is cut
// into pieces of 2 x (workers of a 256-thread workgroup) partials - one serial addition, then the tree, four wavefronts on
// four SIMDs - spread over the chip, and the workgroup that finishes a run's last piece (a counter in the run's record)
// folds the piece results the same way: depth ~ log2 L + 2.
// A worker is a lane / lane triple of bundle F holding an XYZZ point, or - G1 - a lane PAIR holding half a point (K2 above).
template <class F>
struct XyzzWorker {
  typedef XYZZ<F> Pt;
  typedef typename F::Mem Mem;
  static constexpr u32 PER_WAVE = tree_per_wave<F>(), LANES = F::LANES;
  __device__ __forceinline__ static bool index(u32 &in_block) { u32 g; return worker_index<F>(PER_WAVE, in_block, g); }
  __device__ __forceinline__ static u32 role() { return worker_role<F>(); }
  __device__ __forceinline__ static void identity(Pt &p) { xyzz_set_identity(p); }
  __device__ __forceinline__ static void load(Pt &p, const XYZZ<Mem> *m) { load_xyzz<F>(p, m); }
  __device__ __forceinline__ static void store(XYZZ<Mem> *m, const Pt &p) { store_xyzz<F>(m, p); }
  __device__ __forceinline__ static void add(Pt &acc, const Pt &o) { Pt r; xyzz_add(r, acc, o); acc = r; }
  __device__ __forceinline__ static void tree(Pt &acc, u32 G, u32 sub) { group_reduce_points<F>(acc, G, sub); }
};
struct K2Worker {
  typedef HalfPt Pt;
  typedef FpOps Mem;
  static constexpr u32 PER_WAVE = 32, LANES = 2;
  __device__ __forceinline__ static bool index(u32 &in_block) { in_block = threadIdx.x >> 1; return true; }
  __device__ __forceinline__ static u32 role() { return k2_role(); }
  __device__ __forceinline__ static void identity(Pt &p) { k2_set_identity(p); }
  __device__ __forceinline__ static void load(Pt &p, const XYZZ<FpOps> *m) { k2_load(p, m); }
  __device__ __forceinline__ static void store(XYZZ<FpOps> *m, const Pt &p) { k2_store(m, p); }
  __device__ __forceinline__ static void add(Pt &acc, const Pt &o) { k2_add(acc, acc, o); }
  __device__ __forceinline__ static void tree(Pt &acc, u32 G, u32 sub) { k2_group_reduce(acc, G, sub); }
};
struct K6Worker {   // G2 on lane sextets (5'' above)
  typedef HalfPt Pt;
  typedef Fp2Ops Mem;
  static constexpr u32 PER_WAVE = K6_PER_WAVE, LANES = 6;
  __device__ __forceinline__ static bool index(u32 &in_block) {
    const u32 t = (k3_lane() * 43u) >> 8;   // sextet inside the wavefront
    in_block = (threadIdx.x >> 6) * PER_WAVE + t;
    return t < PER_WAVE;
  }
  __device__ __forceinline__ static u32 role() { return k6_lane_in_worker(); }
  __device__ __forceinline__ static void identity(Pt &p) { fe_zero(p.u); fe_zero(p.v); }
  __device__ __forceinline__ static void load(Pt &p, const XYZZ<Fp2Ops> *m) { k6_load(p, m); }
  __device__ __forceinline__ static void store(XYZZ<Fp2Ops> *m, const Pt &p) { k6_store(m, p); }
  __device__ __forceinline__ static void add(Pt &acc, const Pt &o) { k6_add(acc, acc, o); }
  __device__ __forceinline__ static void tree(Pt &acc, u32 G, u32 sub) { k6_group_reduce(acc, G, sub); }
};
constexpr u32 LONG_THREADS = 256;
template <class WK>
constexpr u32 long_workers() { return (LONG_THREADS / 64) * WK::PER_WAVE; }
template <class WK>
constexpr u32 long_piece() { return 2 * long_workers<WK>(); }   // partials per piece
// sum over the workgroup's workers -> worker 0 (every thread of the workgroup calls it)
template <class WK>
__device__ __forceinline__ void long_block_sum(typename WK::Pt &acc, bool live, u32 wid, typename WK::Pt (*wave_part)[WK::LANES]) {
  constexpr u32 PW = WK::PER_WAVE, NWAVES = LONG_THREADS / 64;
  const u32 wave = threadIdx.x >> 6, t_in_wave = wid - wave * PW, role = WK::role();
  if (live) {
    WK::tree(acc, PW, t_in_wave);
    if (t_in_wave == 0) wave_part[wave][role] = acc;
  }
  __syncthreads();
  if (wave == 0 && live) {
    if (t_in_wave < NWAVES) acc = wave_part[t_in_wave][role]; else WK::identity(acc);
    WK::tree(acc, NWAVES, t_in_wave);
  }
  __syncthreads();
}
// Medium runs: tail[l0] + sum of head[l0+1 .. l1] by G workers per run (G a power of two chosen by the host, PER_WAVE / G
// runs per wavefront): every worker folds every G-th partial, then a shuffle tree.  The case this is for: the top window
// of the 13-bit plans and the buckets of window-table plans, a few times fuller than a chunk (8-32 chunks).  Wavefront
// `wave` of `nwaves` (no barrier in here: the wavefronts of a workgroup proceed independently).
template <class WK>
__device__ __forceinline__ void merge_medium_runs(XYZZ<typename WK::Mem> *pts, const XYZZ<typename WK::Mem> *head,
                                                  const XYZZ<typename WK::Mem> *tail, u32 c, u32 chunks_per_window,
                                                  const LongRun *runs, u32 nruns, u32 G, u32 wave, u32 nwaves) {
  typedef typename WK::Pt Pt;
  constexpr u32 PW = WK::PER_WAVE;
  u32 wid;
  const bool live = WK::index(wid);
  const u32 t = wid - (threadIdx.x >> 6) * PW;   // worker inside the wavefront
  const u32 per_wave = PW / G, r = t / G, k = t & (G - 1);
  for (u32 base = wave * per_wave; base < nruns; base += nwaves * per_wave) {
    const u32 e = base + r;
    const bool valid = live && e < nruns;
    LongRun lr = {0, 0, 0, 0};
    if (valid) lr = runs[e];
    const u64 slot0 = (u64)lr.w * chunks_per_window;
    Pt acc;
    WK::identity(acc);
    if (valid) {
      if (k == 0) WK::load(acc, tail + slot0 + lr.lane);
      for (u32 j = lr.lane + 1 + k; j <= lr.last; j += G) {
        Pt o;
        WK::load(o, head + slot0 + j);
        WK::add(acc, o);
      }
    }
    WK::tree(acc, G, k);   // (every lane of the wavefront takes part in the shuffles)
    if (valid && k == 0) WK::store(&pts[((u64)lr.w << (c - 1)) + lr.d - 1], acc);
  }
}
// Big runs, piece by piece: workgroup `blk` of `nblk`.
template <class WK>
__device__ __forceinline__ void merge_big_runs(XYZZ<typename WK::Mem> *pts, const XYZZ<typename WK::Mem> *head,
                                               const XYZZ<typename WK::Mem> *tail, u32 c, u32 chunks_per_window,
                                               BigRun *runs, u32 nruns, XYZZ<typename WK::Mem> *piece_out, u32 total,
                                               u32 blk, u32 nblk, void *lds_part, u32 *sh_run, u32 *sh_last) {
  typedef typename WK::Pt Pt;
  constexpr u32 NWORK = long_workers<WK>(), PIECE = long_piece<WK>();
  Pt(*wave_part)[WK::LANES] = reinterpret_cast<Pt(*)[WK::LANES]>(lds_part);
  u32 wid;
  const bool live = WK::index(wid);   // idle lanes stay for the barriers
  for (u32 piece = blk; piece < total; piece += nblk) {
    if (threadIdx.x == 0) *sh_run = 0xffffffffu;
    __syncthreads();
    for (u32 e = threadIdx.x; e < nruns; e += LONG_THREADS)
      if (piece - runs[e].piece0 < runs[e].npieces) *sh_run = e;   // exactly one run owns the piece
    __syncthreads();
    const u32 e = *sh_run;
    if (e == 0xffffffffu) continue;   // (uniform over the workgroup)
    const u32 rw = runs[e].w, rlane = runs[e].lane, rd = runs[e].d, rlast = runs[e].last, p0 = runs[e].piece0, np = runs[e].npieces;
    const u64 slot0 = (u64)rw * chunks_per_window;
    XYZZ<typename WK::Mem> *out = &pts[((u64)rw << (c - 1)) + rd - 1];
    // partial t of the run: t == 0 the tail partial of its first chunk, then the head partials of the chunks that follow
    const u32 t0 = (piece - p0) * PIECE, nparts = rlast - rlane + 1;
    const u32 t1 = t0 + PIECE < nparts ? t0 + PIECE : nparts;
    Pt acc;
    WK::identity(acc);
    if (live)
      for (u32 t = t0 + wid; t < t1; t += NWORK) {
        Pt o;
        WK::load(o, t == 0 ? tail + slot0 + rlane : head + slot0 + rlane + t);
        WK::add(acc, o);
      }
    long_block_sum<WK>(acc, live, wid, wave_part);
    if (np == 1) {
      if (live && wid == 0) WK::store(out, acc);
      continue;
    }
    if (live && wid == 0) WK::store(&piece_out[piece], acc);
    __threadfence();   // the piece result is visible device-wide before the counter moves
    __syncthreads();
    if (threadIdx.x == 0) *sh_last = atomicAdd(&runs[e].done, 1u) == np - 1 ? 1u : 0u;
    __syncthreads();
    if (*sh_last) {   // every other piece of the run has been published: fold them
      __threadfence();
      WK::identity(acc);
      if (live)
        for (u32 q = wid; q < np; q += NWORK) {
          Pt o;
          WK::load(o, &piece_out[p0 + q]);
          WK::add(acc, o);
        }
      long_block_sum<WK>(acc, live, wid, wave_part);
      if (live && wid == 0) WK::store(out, acc);
    }
  }
}
// The two parts as launches of their own (one wavefront per workgroup for the medium runs): what the tiny G2 tables run.
// The fused kernel below needs 256-thread workgroups for its long part, and the lane-triple instantiation then allocates
// 256 VGPRs + 47 AGPRs where this one fits 256 + 0 - the medium runs of a window table over 2^10 G2 points (128 runs of
// 32 partials, the whole job) took 351 us fused against ~230 us here (profiles/r6_call14_small_jobs.txt,
// r6_call15_tail_split_ab.txt).
template <class WK>
__global__ __launch_bounds__(64, WK::LANES == 1 && sizeof(typename WK::Pt) > 200 ? 1 : 2) void msm_merge_runs_kernel(
    XYZZ<typename WK::Mem> *pts, const XYZZ<typename WK::Mem> *head, const XYZZ<typename WK::Mem> *tail, u32 c,
    u32 chunks_per_window, const LongRun *runs, u32 max_runs, u32 G, const ErrFlags *err) {
  u32 nruns = err->nlong;
  if (nruns > max_runs) nruns = max_runs;
  merge_medium_runs<WK>(pts, head, tail, c, chunks_per_window, runs, nruns, G, blockIdx.x, gridDim.x);
}
template <class WK>
__global__ __launch_bounds__(LONG_THREADS) void msm_merge_long_kernel(XYZZ<typename WK::Mem> *pts,
                                                                      const XYZZ<typename WK::Mem> *head,
                                                                      const XYZZ<typename WK::Mem> *tail, u32 c,
                                                                      u32 chunks_per_window, BigRun *big_runs, u32 max_big,
                                                                      XYZZ<typename WK::Mem> *piece_out, u32 max_pieces,
                                                                      const ErrFlags *err) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_part[sizeof(typename WK::Pt) * (LONG_THREADS / 64) * WK::LANES];
  __shared__ u32 sh_run, sh_last;
  u32 nbig = err->nbig, total = err->npieces;
  if (nbig > max_big) nbig = max_big;
  if (total > max_pieces) total = max_pieces;
  merge_big_runs<WK>(pts, head, tail, c, chunks_per_window, big_runs, nbig, piece_out, total, blockIdx.x, gridDim.x, lds_part,
                     &sh_run, &sh_last);
}
// ONE launch for both: workgroups [0, run_blocks) fold the medium runs (a wavefront each at a time, worker policy WKR),
// the rest the pieces of the big runs (policy WKL) - side by side, not one after the other: a window table over 2^16
// points has 4 096 medium runs AND ~115 big ones (the top row's 8-bit digits), 130 us + 170 us as two launches
// (profiles/r6_call7_k2_ab.txt).
template <class WKR, class WKL>
__global__ __launch_bounds__(LONG_THREADS) void msm_merge_tail_kernel(XYZZ<typename WKL::Mem> *pts,
                                                                      const XYZZ<typename WKL::Mem> *head,
                                                                      const XYZZ<typename WKL::Mem> *tail, u32 c,
                                                                      u32 chunks_per_window, const LongRun *runs, u32 max_runs,
                                                                      u32 G, u32 run_blocks, BigRun *big_runs, u32 max_big,
                                                                      XYZZ<typename WKL::Mem> *piece_out, u32 max_pieces,
                                                                      const ErrFlags *err) {
  static_assert(std::is_same<typename WKR::Mem, typename WKL::Mem>::value, "both parts work on the same records");
  __shared__ __attribute__((aligned(16))) unsigned char lds_part[sizeof(typename WKL::Pt) * (LONG_THREADS / 64) * WKL::LANES];
  __shared__ u32 sh_run, sh_last;
  if (blockIdx.x < run_blocks) {
    u32 nruns = err->nlong;
    if (nruns > max_runs) nruns = max_runs;   // (cannot happen: max_long is an upper bound, msm_enqueue)
    merge_medium_runs<WKR>(pts, head, tail, c, chunks_per_window, runs, nruns, G,
                           blockIdx.x * (LONG_THREADS / 64) + (threadIdx.x >> 6), run_blocks * (LONG_THREADS / 64));
  } else {
    u32 nbig = err->nbig, total = err->npieces;
    if (nbig > max_big) nbig = max_big;
    if (total > max_pieces) total = max_pieces;
    merge_big_runs<WKL>(pts, head, tail, c, chunks_per_window, big_runs, nbig, piece_out, total, blockIdx.x - run_blocks,
                        gridDim.x - run_blocks, lds_part, &sh_run, &sh_last);
  }
}

// Resident wavefronts per SIMD the lane cost model assumes for the reduction kernels: [0] G1, [1] G2 one lane per
// point, [2] G2 lane triples.  One each: a second resident wavefront does NOT interleave for free in these mad-bound
// chains (profiles/archive/r2_call8_slots.txt: G1 2^17-2^20 reduce 0.73-0.99 ms with 1, 0.92-1.19 ms with 2; G2 within noise).
// BELLMAN_HIP_SUM_SLOTS="a,b,c" overrides for sweeps.
inline double sum_slot_factor(int kind) {
  static const std::array<double, 3> f = [] {
    std::array<double, 3> v = {1.0, 1.0, 1.0};
    if (const char *e = getenv("BELLMAN_HIP_SUM_SLOTS")) sscanf(e, "%lf,%lf,%lf", &v[0], &v[1], &v[2]);
    return v;
  }();
  return f[kind];
}
// ============================================================================================
// fixed-base scalar multiplication (fixture generation) and test hooks
// ============================================================================================
// out[i] = [s_i] base (generator.rs:271-296,398-421 use a windowed fixed-base table on the CPU; so does this).
// [r4] Table T[j][d - 1] = d * 2^(8j) * base, j < 32, d = 1 .. 255 (affine; 0.8 MB for G1, 1.6 MB for G2, L2 resident): a
// scalar is its 32 bytes, a multiplication is at most 32 mixed additions and no doubling at all - 320 field products
// and one inversion instead of the 255 doublings + ~127 additions (3 600 products) of the double-and-add ladder this
// replaces.  generate_parameters for a 2^20-constraint circuit is 4 x 2^20 G1 + 2^20 G2 such multiplications.
constexpr u32 FB_ROWS = 32, FB_COLS = 255;
// step 1: the row bases 2^(8j) * base (one lane per row: 8j doublings, one inversion)
template <class F>
__global__ void __launch_bounds__(64) fixed_base_rows_kernel(Affine<F> base, Affine<F> *table) {
  const u32 j = threadIdx.x;
  if (j >= FB_ROWS) return;
  Affine<F> p = base;
  if (j && !aff_is_identity(base)) {
    XYZZ<F> acc, t;
    xyzz_dbl_affine(acc, base);
    for (u32 k = 1; k < 8 * j; k++) { xyzz_dbl(t, acc); acc = t; }
    xyzz_to_affine(p, acc);
  }
  table[(size_t)j * FB_COLS] = p;
}
// step 2: d * (row base) for d = 2 .. 255 by double-and-add over the 8 bits of d (one lane per entry)
template <class F>
__global__ void __launch_bounds__(128) fixed_base_table_kernel(Affine<F> *table) {
  const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= FB_ROWS * FB_COLS) return;
  const u32 j = e / FB_COLS, d = e % FB_COLS + 1;
  if (d == 1) return;
  const Affine<F> p = table[(size_t)j * FB_COLS];
  Affine<F> r = p;
  if (!aff_is_identity(p)) {
    XYZZ<F> acc, t;
    xyzz_set_identity(acc);
    for (int b = 7; b >= 0; b--) {
      xyzz_dbl(t, acc);
      acc = t;
      if ((d >> b) & 1) xyzz_madd(acc, p);
    }
    xyzz_to_affine(r, acc);
  }
  table[e] = r;
}
template <class F>
__global__ void __launch_bounds__(128) fixed_base_mul_kernel(const Affine<F> *table, const void *scalars, int fmt, u64 n, Affine<F> *out) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fr_t s;
  load_scalar(scalars, i, fmt, s);
  XYZZ<F> acc;
  xyzz_set_identity(acc);
  for (u32 j = 0; j < FB_ROWS; j++) {
    const u32 d = (s.l[j >> 2] >> ((j & 3) * 8)) & 255u;
    if (!d) continue;
    const Affine<F> q = table[(size_t)j * FB_COLS + d - 1];
    if (!aff_is_identity(q)) xyzz_madd(acc, q);   // (an identity base gives an all-identity table)
  }
  Affine<F> r;
  xyzz_to_affine(r, acc);
  out[i] = r;
}
template <class F>
__global__ void __launch_bounds__(128) point_add_kernel(Affine<F> *r, const Affine<F> *a, const Affine<F> *b, u64 n) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  XYZZ<F> x, y, z;
  xyzz_from_affine(x, a[i]);
  xyzz_from_affine(y, b[i]);
  xyzz_add(z, x, y);
  Affine<F> o;
  xyzz_to_affine(o, z);
  r[i] = o;
}
// from_uncompressed's two group-membership tests (bls12_381: is_on_curve & is_torsion_free) for
// points the decode kernel already accepted: y^2 = x^3 + b, then [q]P = O by plain double-and-add
// (q is the same for every lane, so the loop does not diverge).  One-time CRS loading work.
template <class F>
__global__ void __launch_bounds__(64) point_check_kernel(const Affine<F> *pts, u64 n, u32 *status) {
  typedef typename F::T T;
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 st = status[i];
  if (st & (PT_INVALID_MASK | PT_IS_INF)) return;
  const Affine<F> p = pts[i];
  T lhs, rhs, b;
  F::sqr(lhs, p.y);
  F::sqr(rhs, p.x);
  F::mul(rhs, rhs, p.x);
  F::curve_b(b);
  F::add(rhs, rhs, b);
  if (!F::eq(lhs, rhs)) { status[i] = st | PT_OFF_CURVE; return; }
  XYZZ<F> acc;
  xyzz_set_identity(acc);
  for (int bit = 254; bit >= 0; bit--) {
    XYZZ<F> t;
    xyzz_dbl(t, acc);
    acc = t;
    if ((FrParams::mod(bit >> 5) >> (bit & 31)) & 1) xyzz_madd(acc, p);
  }
  if (!xyzz_is_identity(acc)) status[i] = st | PT_NOT_IN_SUBGROUP;
}
// Window table of a registered base vector: row j holds 2^(c*j) P_i in affine form, so that digit j of
// a scalar can go to the same bucket set as digit 0 (one bucket reduction instead of W, no Horner
// over windows).  One lane per base walks the rows: c doublings and one inversion per row.
template <class F>
__global__ void __launch_bounds__(128) window_table_kernel(Affine<F> *table, u64 n, u32 c, u32 W) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<F> p = table[i];
  for (u32 j = 1; j < W; j++) {
    if (!aff_is_identity(p)) {
      XYZZ<F> acc, t;
      xyzz_dbl_affine(acc, p);
      for (u32 k = 1; k < c; k++) { xyzz_dbl(t, acc); acc = t; }
      xyzz_to_affine(p, acc);
    }
    table[(u64)j * n + i] = p;
  }
}
template <class F>
static int window_table_t(void *table_dev, u64 n, u32 c, u32 W, hipStream_t st) {
  const u32 blocks = (u32)((n + 127) / 128);
  if (!blocks) return BH_OK;
  hipLaunchKernelGGL(window_table_kernel<F>, dim3(blocks), dim3(128), 0, st, (Affine<F> *)table_dev, n, c, W);
  BH_HIP_CHECK(hipGetLastError());
  return BH_OK;
}
template <class F>
static int points_check_t(const void *pts_dev, u64 n, u32 *status_dev, hipStream_t st) {
  const u32 blocks = (u32)((n + 63) / 64);
  if (!blocks) return BH_OK;
  hipLaunchKernelGGL(point_check_kernel<F>, dim3(blocks), dim3(64), 0, st, (const Affine<F> *)pts_dev, n, status_dev);
  BH_HIP_CHECK(hipGetLastError());
  return BH_OK;
}
// ============================================================================================
// host orchestration
// ============================================================================================
// F = the ops bundle the accumulation computes with (FpOps for G1; Fp2K3Ops, or single-lane Fp2Ops, for G2), FR the
// bundle of the merge and reduction kernels; records in memory are Affine / XYZZ over F::Mem == FR::Mem, so the two
// can differ: a big G2 job accumulates one lane per point (throughput) and reduces a window table's 2^15 buckets in
// lane triples (latency).
template <class F, class FR>
static int msm_enqueue(MsmJobImpl &job, const void *bases_dev, u64 n_bases, u64 skip, const void *scalars_dev,
                       u64 n, int fmt, const u64 *density_dev, const MsmOpts &opts, const WindowTable *table) {
  typedef typename F::Mem M;
  static_assert(sizeof(typename FR::Mem::T) == sizeof(typename M::T), "both bundles work on the same records");
  typedef XYZZ<M> Pt;
  constexpr bool G2 = (M::WORDS == 24);
  Context &c = *job.ctx;
  hipStream_t st = job.stream;
  // a window table is used when it exists, nobody forces another window size, and its row indices
  // fit the 31-bit base field of a pair
  const bool use_table = table && !(opts.flags & BH_MSM_NO_TABLE) && (opts.c == 0 || opts.c == table->c) &&
                         (u64)table->W * table->stride < ((u64)1 << 31) && (u64)table->W * n < ((u64)1 << 32);
  if (opts.padded_table && G2) return BH_ERR_INVALID_ARG;   // only G1 tables are laid out at a 128-byte stride
  const MsmPlan p = use_table ? make_table_plan(n, *table, opts.chunk, G2, c.num_cus)
                              : make_plan(n, opts.c, opts.chunk, G2);
  // running bucket sum in LDS: only meaningful for the single-lane G2 kernel (its default), or when forced
  const bool lds_acc = F::LANES == 1 && ((opts.flags & BH_MSM_ACC_LDS) ? true : (opts.flags & BH_MSM_ACC_REGISTERS) ? false : G2);
  job.plan = p;
  if ((u64)p.Wd * n >= ((u64)1 << 32)) return BH_ERR_INVALID_ARG;  // pair positions are 32-bit
  if (n_bases >= ((u64)1 << 31)) return BH_ERR_INVALID_ARG;        // base index shares its word with the sign bit

  // ---- one workspace block per job, carved into its buffers (one pool round trip, one memset) ----------
  const u64 npairs = (u64)p.Wd * n;
  const u64 ncounts = (u64)p.W * 256 * p.num_tiles;
  const u64 nslots = (u64)p.W * p.chunks_per_window;
  // the owner lane of a run folds at most `walk` following chunks itself; longer runs are queued for
  // msm_merge_runs_kernel (G workers per run), the longest of those for the workgroup kernel
  const u32 walk = 4;
  // G workers per queued run: 8-16 chunk runs become 1-2 serial additions + 3 tree levels; when the AVERAGE run is
  // much longer than that (window tables over tiny vectors: 32 n entries in 128 buckets, chunks of 8) more workers
  // shorten the chain as long as the launch still fits the chip
  // [r6] G1: the same on lane PAIRS (K2: half the latency of an addition for twice the lanes) when that is cheaper by
  // the same model - levels x latency of a level x how far the launch overfills the chip.  BELLMAN_HIP_LONG_K2=0: one lane
  // per point everywhere (A/B)
  static const bool long_k2 = [] { const char *e = getenv("BELLMAN_HIP_LONG_K2"); return !(e && *e == '0'); }();
  // [r6] G2 in lane triples: the same on lane SEXTETS (K6Worker)
  constexpr bool G1_PAIRS = std::is_same<FR, FpOps>::value, G2_SEXTETS = std::is_same<FR, Fp2K3Ops>::value;
  constexpr bool PAIRS_POSSIBLE = G1_PAIRS || G2_SEXTETS;   // "pairs" below: the half-point worker of the group
  typedef typename std::conditional<G1_PAIRS, K2Worker, typename std::conditional<G2_SEXTETS, K6Worker, XyzzWorker<FR>>::type>::type HalfWorker;
  constexpr double LEVEL_US = G2_SEXTETS ? 37.0 : 19.0, HALF_LEVEL_US = G2_SEXTETS ? 20.0 : 10.5;   // a tree level, by worker kind
  u32 run_lanes = 8;
  bool runs_on_pairs = false;
  {
    const double avg_chunks = (double)p.n / (double)p.nb / (double)p.chunk;   // per window: n sorted entries, nb buckets
    double best_cost = 1e30;
    auto sweep = [&](u32 per_wave, double level_us, bool pairs) {
      for (u32 g = 8, lg = 3; g <= per_wave; g <<= 1, lg++) {
        const double steps = std::ceil(std::max(1.0, avg_chunks) / g) + lg;
        const double waves = (double)p.NB * g / (double)per_wave;
        const double cost = steps * level_us * std::max(1.0, waves / ((double)c.num_cus * 4));
        if (cost < best_cost) { best_cost = cost; run_lanes = g; runs_on_pairs = pairs; }
      }
    };
    sweep(tree_per_wave<FR>(), LEVEL_US, false);
    if (PAIRS_POSSIBLE && long_k2) sweep(HalfWorker::PER_WAVE, HALF_LEVEL_US, true);
    // (the model takes every bucket for a queued run - true of window-table plans; where a typical bucket fits a chunk only
    // the few outliers are queued and the chip has the lanes)
    if (PAIRS_POSSIBLE && long_k2 && avg_chunks <= 1.0) { run_lanes = 8; runs_on_pairs = true; }
  }
  const u32 max_long = (u32)(nslots / (walk + 1) + 1);
  // runs of more than big_chunks chunks - more than four serial additions per worker of msm_merge_runs_kernel - are cut
  // into workgroup-sized pieces (msm_merge_long_kernel); G1 pieces run on lane pairs.  BELLMAN_HIP_LONG_K2=0: one lane
  // per point there too (A/B)
  constexpr bool LONG_ON_PAIRS = PAIRS_POSSIBLE;
  const bool long_pairs = LONG_ON_PAIRS && long_k2;
  // ... but never a run that is merely TYPICAL: where the average bucket already spans dozens of chunks (32 rows of 8 bits over
  // 2^11 G2 points: 64 chunks per bucket) "big" starts at twice the average (round 6, first cut: half of that table's runs
  // went down the long path, 0.81 -> 1.06 ms)
  const u32 big_chunks = std::max(std::max(32u, 4u * run_lanes), (u32)(2.0 * (double)p.n / (double)p.nb / (double)p.chunk));
  const u32 piece = long_pairs ? long_piece<HalfWorker>() : long_piece<XyzzWorker<FR>>();
  const u32 max_big = (u32)(nslots / (big_chunks + 1) + 1);
  // sum of ceil(L_r / piece) over the big runs: consecutive runs share one chunk, so sum L_r <= nslots + max_big
  const u32 max_pieces = (u32)((nslots + max_big) / piece + max_big + 1);
  const u32 H = 1u << p.hi_bits, Lw = 1u << p.lo_bits;
  const u64 nwords = (n + 63) / 64;
  size_t off = 0;
  auto carve = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~size_t(255); return at; };
  // zero-initialised region first: status words, the result slots (so that both leave in ONE copy), then the buckets
  // (all-zero XYZZ == identity)
  const size_t bits_bytes = (size_t)p.W * p.c * sizeof(Pt);   // per window: (c-1) bit sums U[w][p], then W plain totals T[w]
  const size_t o_err = carve(sizeof(ErrFlags));                // 256-byte slot
  const size_t o_bits = carve(bits_bytes);
  const size_t o_pts = carve((u64)p.NB * sizeof(Pt));
  const size_t zero_bytes = off;
  const size_t o_pairs_a = carve(npairs * 8), o_pairs_b = carve(npairs * 8);
  const size_t o_counts = carve((ncounts + 1) * 4), o_scan = carve(scan_tmp_elems(ncounts + 1) * 4), o_zstart = carve((u64)p.W * 4);
  const size_t o_head = carve(nslots * sizeof(Pt)), o_tail = carve(nslots * sizeof(Pt));
  const size_t o_long = carve((u64)max_long * sizeof(LongRun)), o_big = carve((u64)max_big * sizeof(BigRun));
  const size_t o_pieces = carve((u64)max_pieces * sizeof(Pt));
  const size_t o_rowcol = carve((u64)p.W * (H + Lw) * sizeof(Pt));
  // [r6] piece sums of the two-stage row / column sums (G1, 2^17 ... 2^21 buckets; below): pieces of `two_len` elements - the
  // launch of 2 NB / two_len lane pairs is then two wavefronts per SIMD (four with one lane per point and half the length)
  // [r6] the same for G2 sets reduced on lane triples (the 2^19 buckets of a 20-bit table): 16 workers per wavefront
  constexpr bool TWO_STAGE_FR = std::is_same<FR, FpOps>::value || std::is_same<FR, Fp2K3Ops>::value;
  u32 two_len = 16;
  if (TWO_STAGE_FR) {
    static const int len_env = [] { const char *e = getenv("BELLMAN_HIP_SUM_TWO_LEN"); return e && *e ? atoi(e) : 0; }();
    const u64 target = (u64)c.num_cus * 4 * 2 * (G2 ? 16 : 32);
    two_len = 4;
    while (two_len < 64 && 2ull * p.NB / two_len > target) two_len <<= 1;
    if (len_env >= 2) two_len = (u32)len_env;
  }
  const bool want_part = TWO_STAGE_FR && p.NB >= (1u << 17) && p.NB <= (1u << 21) && Lw >= 32 && H >= 32;
  const size_t o_part = want_part ? carve((u64)p.W * ((u64)H * (Lw / std::min(two_len, Lw)) + (u64)Lw * (H / std::min(two_len, H))) * sizeof(Pt)) : 0;
  const size_t o_prefix = density_dev ? carve((nwords + 1) * 4) : 0;
  char *ws = (char *)c.pool.acquire(off);
  if (!ws) return BH_ERR_HIP;
  job.dev_allocs.push_back(ws);
  MsmBuffers b;
  b.err = (ErrFlags *)(ws + o_err);
  b.pairs_a = (u64 *)(ws + o_pairs_a); b.pairs_b = (u64 *)(ws + o_pairs_b);
  b.counts = (u32 *)(ws + o_counts); b.scan_tmp = (u32 *)(ws + o_scan); b.zstart = (u32 *)(ws + o_zstart);
  b.word_prefix = density_dev ? (u32 *)(ws + o_prefix) : nullptr;
  Pt *pts = (Pt *)(ws + o_pts), *head = (Pt *)(ws + o_head), *tail = (Pt *)(ws + o_tail);
  LongRun *long_runs = (LongRun *)(ws + o_long);
  BigRun *big_runs = (BigRun *)(ws + o_big);
  Pt *piece_out = (Pt *)(ws + o_pieces);
  Pt *rowcol = (Pt *)(ws + o_rowcol), *bits = (Pt *)(ws + o_bits);
  Pt *part = want_part ? (Pt *)(ws + o_part) : nullptr;
  ErrFlags *err = b.err;
  job.err_dev = err;
  job.scalars_dev = scalars_dev; job.density_dev = density_dev; job.word_prefix = b.word_prefix;
  // what the error-resolution pass reads is what the accumulation reads: row 0 of a table at the 128-byte stride if
  // that is this job's source (a snapshot, like every table), else the base vector / row 0 of a dense table
  const bool from_padded_table = use_table && opts.padded_table;
  job.bases_dev = from_padded_table ? opts.padded_table : bases_dev;
  job.bases_stride = from_padded_table ? 128u : (u32)sizeof(Affine<M>);
  job.skip = skip; job.n_bases = n_bases; job.fmt = fmt;
  job.ref_n = opts.ref_n; job.always_resolve_ident = opts.always_resolve_ident;

  job.timed = (opts.flags & BH_MSM_STAGE_TIMES) != 0;
  if (job.timed) BH_HIP_CHECK(hipEventRecord(job.ev_begin, st));
  BH_HIP_CHECK(hipMemsetAsync(ws, 0, zero_bytes, st));
  const u64 *sorted = nullptr;
  // small multiexps over a window table: one launch for everything up to the filled buckets (msm_small_fill_kernel)
  static const bool small_on = [] { const char *e = getenv("BELLMAN_HIP_SMALL_FUSED"); return !(e && *e == '0'); }();
  // ... and only when no bucket is EXPECTED to outgrow its list: rows whose top one is a sliver of t = 255 - (Wd - 1) c bits (10-bit
  // rows: t = 5, 11 bits: 2, 12 bits: 3) send the top digit of every scalar to 2^t buckets, nd / 2^t entries each - such buckets fall
  // back to a scan of the whole digit table per worker (an explicit 11-bit table over 2^9 points took 8.9 ms in this kernel against
  // 0.7 through the sort: profiles/r6_call54_tiny_g2_bits.txt); the default 13-bit tables have t = 8
  const u32 top_bits = 255u - (p.Wd - 1) * p.c;
  const u64 sliver_load = top_bits >= 31 ? 0 : ((u64)p.nd >> top_bits);
  const bool small_fused = small_on && use_table && !opts.padded_table && p.W == 1 && p.nd <= SMALL_MAX_SCALARS && p.n <= SMALL_MAX_ENTRIES &&
                           (u64)p.n <= (u64)SMALL_MAX_PER_BUCKET * p.nb && p.c <= 15 && small_fill_lds_bytes(p.nd, p.n) <= 64 * 1024 &&
                           sliver_load + (u64)p.n / p.nb <= SMALL_LIST_CAP / 2 && !(opts.flags & BH_MSM_NO_SMALL_PATH);
  if (small_fused) {
    const u32 bpb = workers_per_block<F>(SMALL_THREADS, tree_per_wave<F>()) / SMALL_LANES_PER_BUCKET;   // buckets per workgroup
    const size_t lds = small_fill_lds_bytes(p.nd, p.n);
    hipLaunchKernelGGL(msm_small_fill_kernel<F>, dim3((p.nb + bpb - 1) / bpb), dim3(SMALL_THREADS), lds, st, scalars_dev, fmt, p.nd,
                       density_dev, b.word_prefix, (u64)skip, (u64)n_bases, p.c, p.Wd, p.base_stride,
                       (const Affine<M> *)bases_dev, pts, err);
    BH_HIP_CHECK(hipGetLastError());
  } else {
    int rc = msm_run_stages(p, b, scalars_dev, fmt, density_dev, skip, n_bases, st, &sorted);
    if (rc) return rc;
  }
  if (job.timed) BH_HIP_CHECK(hipEventRecord(job.ev_sorted, st));
  const bool hold = (opts.flags & BH_MSM_HOLD) != 0;
  if (hold) {
    // the digit / sort stage of a held job joins the barrier set of the accumulation chain: whichever accumulation is
    // started first waits for the sorts of ALL jobs issued so far (create_proof issues its multiexps held, then starts
    // them in chain order - no sort runs beside an accumulation and steals its SIMDs)
    if (!job.res.sort_event) BH_HIP_CHECK(hipEventCreateWithFlags(&job.res.sort_event, hipEventDisableTiming));
    BH_HIP_CHECK(hipEventRecord(job.res.sort_event, st));
    std::lock_guard<std::mutex> g(c.acc_mu);
    if (c.pending_barriers.size() < 32) c.pending_barriers.push_back(job.res.sort_event);
  }
  // everything after the sort: run now, or when the job is started (bh_msm_start / its wait)
  Context *cp = &c;
  MsmJobImpl *jp = &job;
  job.resume = [=]() mutable -> int {
  Context &c = *cp;
  MsmJobImpl &job = *jp;
  // 4. accumulate equal chunks, then fold the buckets that straddle chunk boundaries
  if (small_fused) {
    if (job.timed) BH_HIP_CHECK(hipEventRecord(job.ev_accum, st));
  } else {
    const u32 wpb = workers_per_block<F>(128, default_per_wave<F>());
    const dim3 grid((p.chunks_per_window + wpb - 1) / wpb, p.W);
    const Affine<M> *bases = (const Affine<M> *)bases_dev;
    // G1 records at a 128-byte stride for the gathers of the classic plan (api.hip bh_bases::padded)
    const bool padded = opts.padded_bases && !use_table && !G2 && F::LANES == 1 && !lds_acc;
    // ... and of the table plan, when the table is one that was laid out that way (bh_bases::table_padded)
    const bool padded_tab = opts.padded_table && use_table;
    // launches that fill the chip join the context's accumulation chain (common.hpp)
    static const bool chain_on = [] { const char *e = getenv("BELLMAN_HIP_ACC_CHAIN"); return !(e && *e == '0'); }();
    const bool chained = chain_on && (u64)grid.x * grid.y * 128 >= (u64)c.num_cus * 4 * 64;
    std::unique_lock<std::mutex> chain_lock(c.acc_mu, std::defer_lock);
    if (chained) {
      if (!job.res.acc_event) BH_HIP_CHECK(hipEventCreateWithFlags(&job.res.acc_event, hipEventDisableTiming));
      chain_lock.lock();
      if (c.last_acc_event) BH_HIP_CHECK(hipStreamWaitEvent(st, c.last_acc_event, 0));
      for (hipEvent_t ev : c.pending_barriers) BH_HIP_CHECK(hipStreamWaitEvent(st, ev, 0));   // bh_ctx_accumulations_after
      c.pending_barriers.clear();
    }
    // (every variant of the kernel takes the record stride, so a padded table is read correctly by all of them)
    const Affine<M> *acc_bases = padded_tab ? (const Affine<M> *)opts.padded_table : padded ? (const Affine<M> *)opts.padded_bases : bases;
    const u32 acc_stride = (padded || padded_tab) ? 128u : (u32)sizeof(Affine<M>);
    if constexpr (F::LANES == 1) {
      if (lds_acc)
        hipLaunchKernelGGL((msm_accumulate_kernel<F, true>), grid, dim3(128), 0, st, sorted, b.zstart, acc_bases, pts, head,
                           tail, p.n, p.c, p.chunk, p.chunks_per_window, err, acc_stride);
      else
        hipLaunchKernelGGL((msm_accumulate_kernel<F, false>), grid, dim3(128), 0, st, sorted, b.zstart, acc_bases, pts, head,
                           tail, p.n, p.c, p.chunk, p.chunks_per_window, err, acc_stride);
    } else {
      hipLaunchKernelGGL((msm_accumulate_kernel<F, false>), grid, dim3(128), 0, st, sorted, b.zstart, acc_bases, pts, head,
                         tail, p.n, p.c, p.chunk, p.chunks_per_window, err, acc_stride);
    }
    BH_HIP_CHECK(hipGetLastError());
    if (chained) {
      BH_HIP_CHECK(hipEventRecord(job.res.acc_event, st));
      c.last_acc_event = job.res.acc_event;
      chain_lock.unlock();
    }
    if (job.timed) BH_HIP_CHECK(hipEventRecord(job.ev_accum, st));   // brackets exactly the accumulate launch
    if (job.hp_stream) {   // the rest of the job (latency-bound chains) runs on the high-priority stream
      BH_HIP_CHECK(hipEventRecord(job.hp_event, st));
      BH_HIP_CHECK(hipStreamWaitEvent(job.hp_stream, job.hp_event, 0));
      st = job.hp_stream;
    }
    const u32 rwpb = workers_per_block<FR>(128, default_per_wave<FR>());
    const dim3 rgrid((p.chunks_per_window + rwpb - 1) / rwpb, p.W);
    hipLaunchKernelGGL(msm_merge_chunks_kernel<FR>, rgrid, dim3(128), 0, st, sorted, b.zstart, pts, head, tail, p.n,
                       p.c, p.chunk, p.chunks_per_window, walk, long_runs, max_long, big_runs, max_big, big_chunks, piece, err);
    BH_HIP_CHECK(hipGetLastError());
    // medium runs and the pieces of the big runs in ONE launch (a launch with nothing to do costs ~5 us): a wavefront per
    // SIMD for the medium runs, one workgroup per piece at a time for the big ones
    const u32 run_blocks = (u32)c.num_cus, long_blocks = std::min<u32>(max_pieces, (u32)c.num_cus * 8);
    const dim3 tgrid(run_blocks + long_blocks);
#define BH_TAIL(WKR, WKL)                                                                                                   \
    hipLaunchKernelGGL((msm_merge_tail_kernel<WKR, WKL>), tgrid, dim3(LONG_THREADS), 0, st, pts, head, tail, p.c,              \
                       p.chunks_per_window, long_runs, max_long, run_lanes, run_blocks, big_runs, max_big, piece_out,           \
                       max_pieces, err)
    // One launch (medium runs and big pieces side by side) - except the 128-bucket window tables of tiny G2 vectors, whose
    // 128 medium runs are the whole job: two launches there (G2 2^10: 0.74 against 0.85 ms; 2^12 ... 2^16 and every G1 size are
    // equal or faster fused - G1 2^16 0.79 against 0.88 ms; profiles/r6_call15_tail_split_ab.txt).  BELLMAN_HIP_TAIL_FUSED=0 / 1
    // forces either for an A/B
    static const int fused_env = [] { const char *e = getenv("BELLMAN_HIP_TAIL_FUSED"); return e && *e ? (*e == '0' ? 0 : 1) : -1; }();
    const bool fused = fused_env >= 0 ? fused_env == 1 : (PAIRS_POSSIBLE || p.NB > 128);
    if (!fused) {
      const dim3 rg((u32)c.num_cus * 4), lg(long_blocks);
#define BH_SPLIT(WKR, WKL)                                                                                                  \
      hipLaunchKernelGGL(msm_merge_runs_kernel<WKR>, rg, dim3(64), 0, st, pts, head, tail, p.c, p.chunks_per_window, long_runs,  \
                         max_long, run_lanes, err);                                                                         \
      hipLaunchKernelGGL(msm_merge_long_kernel<WKL>, lg, dim3(LONG_THREADS), 0, st, pts, head, tail, p.c, p.chunks_per_window,   \
                         big_runs, max_big, piece_out, max_pieces, err)
      if constexpr (PAIRS_POSSIBLE) {
        if (runs_on_pairs && long_pairs) { BH_SPLIT(HalfWorker, HalfWorker); }
        else if (long_pairs) { BH_SPLIT(XyzzWorker<FR>, HalfWorker); }
        else { BH_SPLIT(XyzzWorker<FR>, XyzzWorker<FR>); }
      } else {
        BH_SPLIT(XyzzWorker<FR>, XyzzWorker<FR>);
      }
#undef BH_SPLIT
    } else if constexpr (PAIRS_POSSIBLE) {
      if (runs_on_pairs && long_pairs) BH_TAIL(HalfWorker, HalfWorker);
      else if (long_pairs) BH_TAIL(XyzzWorker<FR>, HalfWorker);
      else BH_TAIL(XyzzWorker<FR>, XyzzWorker<FR>);
    } else {
      BH_TAIL(XyzzWorker<FR>, XyzzWorker<FR>);
    }
#undef BH_TAIL
    BH_HIP_CHECK(hipGetLastError());
  }
  // 5. reduce: rows (sum over lo, contiguous), columns (sum over hi, stride Lw), then bits.
  // G workers per output chosen so that each launch is about one wavefront per SIMD.
  Pt *rows = rowcol, *cols = rowcol + (u64)p.W * H;
  constexpr u32 PW = 64;   // workers per block of the sum kernel (one wavefront, or four wavefronts of 16 lane triples)
  // workers per output: minimise (serial adds per worker + tree depth) x (waves per SIMD, at least 1);
  // these kernels are latency-bound chains of point additions, not throughput-bound.
  auto pick_lanes = [&](u32 groups, u32 count) {
    // wavefront slots per SIMD the launch may fill before a step stretches (sum_slot_factor); a block of lane
    // triples is four wavefronts
    const double slots = (double)c.num_cus * 4 * sum_slot_factor(G2 ? (FR::LANES == 1 ? 1 : 2) : 0);
    const double waves_per_block = FR::LANES == 3 ? 4.0 : 1.0;
    u32 best = 1;
    double best_cost = 1e30;
    for (u32 g = 1, lg = 0; g <= PW; g <<= 1, lg++) {
      if (g > count && g > 1) break;
      // a lane-triple group wider than one wavefront pays two barriers and an LDS round trip (measured: ~2 additions)
      const double steps = (double)((count + g - 1) / g) + lg + ((FR::LANES == 3 && g > 16) ? 2.0 : 0.0);
      const double waves = (double)groups * g / (double)PW * waves_per_block;
      const double cost = steps * std::max(1.0, waves / slots);
      if (cost < best_cost) { best_cost = cost; best = g; }
    }
    return best;
  };
  auto blocks_for = [&](u32 groups, u32 lanes) { return (u32)(((u64)groups * lanes + PW - 1) / PW); };
  auto make_job = [&](const Pt *in, Pt *out, SumDesc d) {
    SumJob<FR> j;
    d.lanes = d.groups ? pick_lanes(d.groups, d.mode == SUM_BITS ? std::max(1u, d.count / 2) : d.count) : 1;
    j.in = in; j.out = out; j.d = d;
    j.nblocks = blocks_for(d.groups, d.lanes);
    return j;
  };
  // G1 launches that leave at least half of the SIMDs empty run on lane pairs (K2, above): half the latency per
  // addition for twice the lanes.  BELLMAN_HIP_SUM_K2=0 switches it off.
  static const bool k2_on = [] { const char *e = getenv("BELLMAN_HIP_SUM_K2"); return !(e && *e == '0'); }();
  // force: 0 = by the rules below, 1 = lane pairs, 2 = one lane per point (the first stage of a two-stage sum chooses)
  auto launch_sums = [&](SumJobs<FR> js, int force = 0) -> bool {
    const u32 total = js.j[0].nblocks + js.j[1].nblocks + js.j[2].nblocks;
    if (!total) return true;
    if constexpr (std::is_same<FR, FpOps>::value) {
      static const double k2_fill = [] { const char *e = getenv("BELLMAN_HIP_K2_SUM_FILL"); return e && *e ? atof(e) : 4.0; }();   // wavefronts per SIMD the lane-pair launch may reach
      // [r6] a handful of long sums (the bit sums and the total of ONE big bucket set): one workgroup of up to eight
      // wavefronts per output (msm_sum_k2_wide_kernel).  BELLMAN_HIP_SUM_WIDE=0 switches it off
      static const bool wide_on = [] { const char *e = getenv("BELLMAN_HIP_SUM_WIDE"); return !(e && *e == '0'); }();
      u32 groups_all = 0, max_sel = 0;
      for (int q = 0; q < 3; q++)
        if (js.j[q].nblocks) {
          groups_all += js.j[q].d.groups;
          max_sel = std::max(max_sel, js.j[q].d.mode == SUM_BITS ? js.j[q].d.count / 2 : js.j[q].d.count / js.j[q].d.splits);
        }
      if (k2_on && wide_on && force == 0 && max_sel >= 128 && groups_all <= 2u * (u32)c.num_cus) {
        // (never more than four wavefronts: a CU has four SIMDs, and the wavefronts of a workgroup that share one take turns -
        // eight were 198 us for 2 + 5 + 3 levels, profiles/r6_call32_timeline.txt)
        u32 nw = 2;
        while (nw < 4 && nw * 64 < max_sel) nw <<= 1;
        for (int q = 0; q < 3; q++)
          if (js.j[q].nblocks) js.j[q].nblocks = js.j[q].d.groups;
        hipLaunchKernelGGL(msm_sum_k2_wide_kernel<FR>, dim3(js.j[0].nblocks + js.j[1].nblocks + js.j[2].nblocks), dim3(64 * nw), 0, st, js);
        return hipGetLastError() == hipSuccess;
      }
      if (force != 2 && k2_on && (force == 1 || (double)total * 2 <= k2_fill * (double)c.num_cus * 4)) {   // blocks are single wavefronts
        for (int q = 0; q < 3; q++) {
          SumJob<FR> &j = js.j[q];
          if (!j.nblocks) continue;
          if (j.d.lanes > 32) j.d.lanes = 32;                  // a wavefront carries 32 lane pairs
          j.nblocks = (u32)(((u64)j.d.groups * j.d.lanes + 31) / 32);
        }
        hipLaunchKernelGGL(msm_sum_k2_kernel<FR>, dim3(js.j[0].nblocks + js.j[1].nblocks + js.j[2].nblocks), dim3(64), 0, st, js);
        return hipGetLastError() == hipSuccess;
      }
    }
    if constexpr (std::is_same<FR, Fp2K3Ops>::value) {
      // G2: the same idea on lane sextets (32 workers per 256-thread block) while the launch stays within one wavefront per SIMD
      if (k2_on && force != 2) {
        SumJobs<FR> k6 = js;
        u32 blocks6 = 0;
        for (int q = 0; q < 3; q++) {
          SumJob<FR> &j = k6.j[q];
          if (!j.nblocks) continue;
          if (j.d.lanes > 32) j.d.lanes = 32;
          j.nblocks = (u32)(((u64)j.d.groups * j.d.lanes + 31) / 32);
          blocks6 += j.nblocks;
        }
        static const double k6_fill = [] { const char *e = getenv("BELLMAN_HIP_K6_SUM_FILL"); return e && *e ? atof(e) : 8.0; }();   // blocks per CU the launch may reach
        if ((double)blocks6 <= k6_fill * (double)c.num_cus) {
          hipLaunchKernelGGL(msm_sum_k6_kernel<FR>, dim3(blocks6), dim3(256), 0, st, k6);
          return hipGetLastError() == hipSuccess;
        }
      }
    }
    hipLaunchKernelGGL(msm_sum_kernel<FR>, dim3(total), dim3(sum_block_threads<FR>()), 0, st, js);
    return hipGetLastError() == hipSuccess;
  };
  {
    const u32 cb = p.c - 1;   // bits of a bucket index
    SumDesc dr, dc;
    dr.mode = SUM_STRIDED; dr.groups = p.W * H; dr.count = Lw; dr.inner = H; dr.stride = 1; dr.istride = Lw; dr.group_shift = cb;
    dc = dr; dc.groups = p.W * Lw; dc.count = H; dc.inner = Lw; dc.stride = Lw; dc.istride = 1;
    SumJobs<FR> js;
    js.j[0] = make_job(pts, rows, dr); js.j[1] = make_job(pts, cols, dc); js.j[2] = js.j[1]; js.j[2].nblocks = 0;
    // [r6] G1, bucket sets of 2^17 ... 2^21 points (16 windows of 2^15, or the 2^19 buckets of a 20-bit window table): the launch
    // above gives every output 16-32 lane pairs - 8-16 serial additions, then a 4-5 level tree in which half, a quarter, ...
    // of the lanes work: 12-13 levels at four wavefronts per SIMD of which 61 % is useful work (0.50 ms for the 2^20 additions
    // of a 2^20-term multiexp, 0.27 at the multiplier's throughput).  Two stages instead: (1) every output is cut into pieces
    // of `len` consecutive elements and ONE worker adds up a piece - no tree, every lane busy - sized so that the launch is
    // two wavefronts per SIMD; (2) the handful of piece sums per output are folded by a small tree launch.
    // BELLMAN_HIP_SUM_TWO_STAGE=0: off; 1: stage one on lane pairs; 2: stage one with one lane per point
    static const int two_env = [] { const char *e = getenv("BELLMAN_HIP_SUM_TWO_STAGE"); return e && *e ? atoi(e) : -1; }();
    bool two_stage = false;
    if constexpr (TWO_STAGE_FR) {
      two_stage = two_env != 0 && part && p.NB >= (1u << 17) && p.NB <= (1u << 21) && Lw >= 32 && H >= 32;
    }
    if (two_stage) {
      // G2 (lane triples): stage one on the lane-triple kernel (16 workers per wavefront; sextets would halve that again)
      static const bool g2_k6_stage1 = [] { const char *e = getenv("BELLMAN_HIP_SUM_G2_STAGE1_K6"); return e && *e == '1'; }();
      const bool one_lane = two_env == 2 || (G2 && !g2_k6_stage1);
      const u32 len_r = std::min(two_len, Lw), len_c = std::min(two_len, H);
      const u32 Sr = Lw / len_r, Sc = H / len_c;
      Pt *part_r = part, *part_c = part + (u64)dr.groups * Sr;
      SumDesc r1 = dr, c1 = dc;
      r1.splits = Sr; r1.groups = dr.groups * Sr; r1.lanes = 1;
      c1.splits = Sc; c1.groups = dc.groups * Sc; c1.lanes = 1;
      SumJobs<FR> s1;
      // (blocks of the one-lane / lane-triple kernel: 64 workers, 84 for lane triples that each sum alone)
      const u32 wpb1 = (G2 && one_lane) ? 4 * default_per_wave<FR>() : PW;
      s1.j[0].in = pts; s1.j[0].out = Sr > 1 ? part_r : rows; s1.j[0].d = r1; s1.j[0].nblocks = (r1.groups + wpb1 - 1) / wpb1;
      s1.j[1].in = pts; s1.j[1].out = Sc > 1 ? part_c : cols; s1.j[1].d = c1; s1.j[1].nblocks = (c1.groups + wpb1 - 1) / wpb1;
      s1.j[2] = s1.j[1]; s1.j[2].nblocks = 0;
      if (!launch_sums(s1, one_lane ? 2 : 1)) return BH_ERR_HIP;
      SumDesc r2, c2;
      r2.mode = SUM_STRIDED; r2.groups = dr.groups; r2.count = Sr; r2.inner = dr.groups; r2.stride = 1; r2.istride = Sr; r2.group_shift = 0;
      c2 = r2; c2.groups = dc.groups; c2.count = Sc; c2.inner = dc.groups; c2.istride = Sc;
      SumJobs<FR> s2;
      s2.j[0] = make_job(part_r, rows, r2); if (Sr <= 1) s2.j[0].nblocks = 0;
      s2.j[1] = make_job(part_c, cols, c2); if (Sc <= 1) s2.j[1].nblocks = 0;
      s2.j[2] = s2.j[1]; s2.j[2].nblocks = 0;
      if (!launch_sums(s2, 1)) return BH_ERR_HIP;
      js.j[0].nblocks = js.j[1].nblocks = 0;   // (done)
    }
    if (G2 && FR::LANES == 1) {
      // single-lane G2 (one resident wavefront per SIMD, so sharing a SIMD doubles every step): the two jobs share
      // one launch, choose their lane counts jointly - the launch lasts as long as its longest chain, stretched by
      // how many wavefronts each SIMD has to interleave.  (With two wavefronts per SIMD - G1, K3-form G2 - they
      // interleave almost for free and the per-job choice above is better.)
      const double simds = (double)c.num_cus * 4;
      double best = 1e30;
      u32 best_r = js.j[0].d.lanes, best_c = js.j[1].d.lanes;
      for (u32 gr = 1, lr = 0; gr <= 64 && gr <= std::max(1u, dr.count); gr <<= 1, lr++)
        for (u32 gc = 1, lc = 0; gc <= 64 && gc <= std::max(1u, dc.count); gc <<= 1, lc++) {
          const double steps = std::max((double)((dr.count + gr - 1) / gr) + lr, (double)((dc.count + gc - 1) / gc) + lc);
          const double waves = ((double)dr.groups * gr + (double)dc.groups * gc) / 64.0;
          const double cost = steps * std::max(1.0, waves / simds);
          if (cost < best) { best = cost; best_r = gr; best_c = gc; }
        }
      js.j[0].d.lanes = best_r; js.j[0].nblocks = blocks_for(dr.groups, best_r);
      js.j[1].d.lanes = best_c; js.j[1].nblocks = blocks_for(dc.groups, best_c);
    }
    if (!launch_sums(js)) return BH_ERR_HIP;
    // sum_idx (idx+1) B[idx] = sum_p 2^p U[p] + T, idx = hi*2^l + lo:
    //   U[w][p], p < lo_bits from the column sums (weights lo), p >= lo_bits from the row sums (weights hi),
    //   T[w] = plain sum of all buckets = sum of the row sums.
    SumDesc bl, bh_, bt;
    bl.mode = SUM_BITS; bl.stride = 1; bl.istride = 0;
    bl.groups = p.W * p.lo_bits; bl.count = Lw; bl.inner = std::max(1u, p.lo_bits); bl.group_shift = p.lo_bits;
    bh_ = bl; bh_.groups = p.W * p.hi_bits; bh_.count = H; bh_.inner = std::max(1u, p.hi_bits); bh_.group_shift = p.hi_bits;
    // the plain total is the sum of the column sums as well as of the row sums: take the shorter vector
    // (this job is the longest chain of the launch)
    const bool t_from_cols = Lw < H;
    bt.mode = SUM_STRIDED; bt.groups = p.W; bt.count = t_from_cols ? Lw : H; bt.inner = 1; bt.stride = 1; bt.istride = 0;
    bt.group_shift = t_from_cols ? p.lo_bits : p.hi_bits;
    js.j[0] = make_job(cols, bits, bl);
    js.j[1] = make_job(rows, bits + (u64)p.W * p.lo_bits, bh_);
    js.j[2] = make_job(t_from_cols ? cols : rows, bits + (u64)p.W * cb, bt);
    if (!launch_sums(js)) return BH_ERR_HIP;
  }
  if (job.timed) BH_HIP_CHECK(hipEventRecord(job.ev_end, st));
  // status words + results to pinned host memory, one copy: [ErrFlags slot (256 B)][bit sums]
  job.host_result_bytes = (o_bits - o_err) + bits_bytes;
  if (job.host_result_bytes > job.res.pinned_bytes) return BH_ERR_INVALID_ARG;
  BH_HIP_CHECK(hipMemcpyAsync(job.host_result, ws + o_err, job.host_result_bytes, hipMemcpyDeviceToHost, st));
  return BH_OK;
  };   // job.resume
  if (hold) return BH_OK;
  const int rc2 = job.resume();
  job.resume = nullptr;
  return rc2;
}

template <> struct HostOf<FpOps> { typedef HostFpOps type; };
template <> struct HostOf<Fp2Ops> { typedef HostFp2Ops type; };

// host tail: result = sum_w 2^(c*w) (sum_p 2^p U[w][p] + T[w])
//   bits layout: [W][lo_bits] column-bit sums, [W][hi_bits] row-bit sums, [W] window totals
template <class F>
static void msm_host_tail(const MsmPlan &p, const XYZZ<F> *bits_dev_layout, void *out_dev_layout) {
  typedef typename HostOf<F>::type H;   // 64-bit-limb host arithmetic, identical record layout
  static_assert(sizeof(XYZZ<H>) == sizeof(XYZZ<F>) && sizeof(Affine<H>) == sizeof(Affine<F>), "layout");
  // the device kernels hand over lazily reduced coordinates ([0, 2p), ff.cuh); the host arithmetic wants
  // canonical ones
  std::vector<XYZZ<F>> canon(bits_dev_layout, bits_dev_layout + (size_t)p.W * p.c);
  for (XYZZ<F> &q : canon) { F::canon(q.x); F::canon(q.y); F::canon(q.zz); F::canon(q.zzz); }
  const XYZZ<H> *bits = reinterpret_cast<const XYZZ<H> *>(canon.data());
  XYZZ<H> acc;
  xyzz_set_identity(acc);
  const u32 cb = p.c - 1;
  const XYZZ<H> *lo = bits, *hi = bits + (size_t)p.W * p.lo_bits, *tot = bits + (size_t)p.W * cb;
  for (int w = (int)p.W - 1; w >= 0; w--) {
    for (int b = (int)p.c - 1; b >= 0; b--) {
      XYZZ<H> t;
      xyzz_dbl(t, acc);
      acc = t;
      if (b < (int)cb) {
        const XYZZ<H> &u = (b >= (int)p.lo_bits) ? hi[(size_t)w * p.hi_bits + (b - p.lo_bits)]
                                                 : lo[(size_t)w * p.lo_bits + b];
        xyzz_add(t, acc, u);
        acc = t;
      }
      if (b == 0) {   // the "+1" of the bucket weights
        xyzz_add(t, acc, tot[w]);
        acc = t;
      }
    }
  }
  Affine<H> res;   // caller buffers carry no alignment guarantee: go through an aligned local
  xyzz_to_affine(res, acc);
  memcpy(out_dev_layout, &res, sizeof res);
}

template <class F>
static int msm_finish(MsmJobImpl &job, void *out_affine, float *ms) {
  Context &c = *job.ctx;
  int rc = BH_OK;
  if (job.trivial) {
    memset(out_affine, 0, sizeof(Affine<F>));
    if (ms) ms[0] = ms[1] = ms[2] = ms[3] = 0.f;
    return job.early_rc;
  }
  if (job.resume) {   // held and never started: start it now
    rc = job.resume();
    job.resume = nullptr;
  }
  if (hipStreamSynchronize(job.stream) != hipSuccess) rc = BH_ERR_HIP;
  if (job.hp_stream && hipStreamSynchronize(job.hp_stream) != hipSuccess) rc = BH_ERR_HIP;
  if (rc == BH_OK) {
    const MsmPlan &p = job.plan;
    constexpr size_t ERR_SLOT = 256;   // the carve granularity of msm_enqueue
    ErrFlags ef;
    memcpy(&ef, job.host_result, sizeof ef);
    if (ms) {  // [0] whole device pipeline, [1] digits+sort, [2] bucket accumulation, [3] reductions
      ms[0] = ms[1] = ms[2] = ms[3] = 0.f;
      if (job.timed) {
        (void)hipEventElapsedTime(&ms[0], job.ev_begin, job.ev_end);
        (void)hipEventElapsedTime(&ms[1], job.ev_begin, job.ev_sorted);
        (void)hipEventElapsedTime(&ms[2], job.ev_sorted, job.ev_accum);
        (void)hipEventElapsedTime(&ms[3], job.ev_accum, job.ev_end);
      }
    }
    job.saw_eof = ef.eof != 0;
    job.saw_ident = ef.ident != 0;
    // [0] sorted entries  [1] zero digits among them  [2] mixed additions of the accumulate launch  [3] chunk lanes
    // [4] window bits c  [5] chunk length K  [6] windows W (1: window-table plan)  [7] digit columns per scalar
    job.done_stats[0] = (u64)p.W * p.n; job.done_stats[1] = ef.zeros; job.done_stats[2] = ef.madds;
    job.done_stats[3] = (u64)p.W * p.chunks_per_window; job.done_stats[4] = p.c; job.done_stats[5] = p.chunk;
    job.done_stats[6] = p.W; job.done_stats[7] = p.Wd;
    if (ef.ident && (ef.eof || job.always_resolve_ident)) {
      // both kinds of failure exist (or the caller folds shards): the reference reports the top window's first failure
      const u64 nref = job.ref_n ? job.ref_n : p.nd;
      const double c_ln = (nref < 32) ? 3.0 : std::ceil(std::log((double)nref));   // multiexp.rs:318-322
      const u32 c_ref = (u32)c_ln, w_ref = (255 + c_ref - 1) / c_ref, lo_ref = c_ref * (w_ref - 1);
      hipLaunchKernelGGL(msm_err_resolve_kernel<F>, dim3((p.nd + 255) / 256), dim3(256), 0, job.stream,
                         job.scalars_dev, job.fmt, p.nd, job.density_dev, job.word_prefix, job.skip, job.n_bases,
                         (const Affine<F> *)job.bases_dev, job.bases_stride, lo_ref, job.err_dev);
      if (hipMemcpyAsync(&ef, job.err_dev, sizeof ef, hipMemcpyDeviceToHost, job.stream) != hipSuccess ||
          hipStreamSynchronize(job.stream) != hipSuccess)
        rc = BH_ERR_HIP;
      else {
        job.saw_ident_top = ef.ident_top != 0;
        rc = !ef.eof ? BH_ERR_UNEXPECTED_IDENTITY : ef.ident_top ? BH_ERR_UNEXPECTED_IDENTITY : BH_ERR_UNEXPECTED_EOF;
      }
    } else if (ef.eof) {
      rc = BH_ERR_UNEXPECTED_EOF;
    } else if (ef.ident) {
      rc = BH_ERR_UNEXPECTED_IDENTITY;
    } else {
      msm_host_tail<F>(p, (const XYZZ<F> *)((const char *)job.host_result + ERR_SLOT), out_affine);
    }
  }
  for (void *ptr : job.dev_allocs) c.pool.release(ptr);
  job.dev_allocs.clear();
  return rc;
}


template <class F>
static int fixed_base_mul_t(const void *base_host, const void *scalars_dev, u64 n, int fmt, void *out_dev,
                            hipStream_t st, void *table_dev) {
  const u32 blocks = (u32)((n + 127) / 128);
  if (!blocks) return BH_OK;
  Affine<F> b;
  memcpy(&b, base_host, sizeof b);
  Affine<F> *table = (Affine<F> *)table_dev;   // FB_ROWS * FB_COLS records, owned by the caller until `st` has drained
  hipLaunchKernelGGL(fixed_base_rows_kernel<F>, dim3(1), dim3(64), 0, st, b, table);
  BH_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(fixed_base_table_kernel<F>, dim3((FB_ROWS * FB_COLS + 127) / 128), dim3(128), 0, st, table);
  BH_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(fixed_base_mul_kernel<F>, dim3(blocks), dim3(128), 0, st, (const Affine<F> *)table, scalars_dev, fmt, n,
                     (Affine<F> *)out_dev);
  BH_HIP_CHECK(hipGetLastError());
  return BH_OK;
}
template <class F>
static int test_point_add_t(void *r, const void *a, const void *b, u64 n, hipStream_t st) {
  const u32 blocks = (u32)((n + 127) / 128);
  if (!blocks) return BH_OK;
  hipLaunchKernelGGL(point_add_kernel<F>, dim3(blocks), dim3(128), 0, st, (Affine<F> *)r, (const Affine<F> *)a,
                     (const Affine<F> *)b, n);
  BH_HIP_CHECK(hipGetLastError());
  return BH_OK;
}
// Host-side group helpers.  Caller records are plain byte buffers without alignment guarantees,
// so every access goes through aligned locals.
template <class F>
static void generic_point_add(void *r, const void *a, const void *b, u64 n) {
  for (u64 i = 0; i < n; i++) {
    Affine<F> pa, pb, pr;
    memcpy(&pa, (const char *)a + i * sizeof pa, sizeof pa);
    memcpy(&pb, (const char *)b + i * sizeof pb, sizeof pb);
    XYZZ<F> x, y, z;
    xyzz_from_affine(x, pa);
    xyzz_from_affine(y, pb);
    xyzz_add(z, x, y);
    xyzz_to_affine(pr, z);
    memcpy((char *)r + i * sizeof pr, &pr, sizeof pr);
  }
}
template <class F>
static void generic_point_mul(void *r, const void *a, const void *k) {
  Affine<F> base, res;
  u32 kw[8];
  memcpy(&base, a, sizeof base);
  memcpy(kw, k, sizeof kw);
  XYZZ<F> acc;
  xyzz_set_identity(acc);
  for (int b = 255; b >= 0; b--) {
    XYZZ<F> t;
    xyzz_dbl(t, acc);
    acc = t;
    if (((kw[b >> 5] >> (b & 31)) & 1) && !aff_is_identity(base)) xyzz_madd(acc, base);
  }
  xyzz_to_affine(res, acc);
  memcpy(r, &res, sizeof res);
}
// [k] a with signed 4-bit windows: 8 precomputed multiples, 256 doublings, at most 64 additions (the plain
// double-and-add above costs ~128 more additions); the serial tails of create_proof (prover.rs:326-354) and the
// tiny-multiexp path run on the host, where this is ~2x faster
template <class F>
static void windowed_point_mul(void *r, const void *a, const u32 *k) {
  Affine<F> base, res;
  memcpy(&base, a, sizeof base);
  u32 kw[9];
  memcpy(kw, k, 32);
  kw[8] = 0;
  bool zero = true;
  for (int i = 0; i < 8; i++) zero &= kw[i] == 0;
  if (aff_is_identity(base) || zero) {
    memset(r, 0, sizeof res);
    return;
  }
  XYZZ<F> tab[8];   // (i + 1) * base
  xyzz_from_affine(tab[0], base);
  xyzz_dbl(tab[1], tab[0]);
  for (int i = 2; i < 8; i++) xyzz_add(tab[i], tab[i - 1], tab[0]);
  // signed digits d_j in [-8, 8], low to high with carry; 65 digits cover a 256-bit scalar + carry
  int digits[65];
  u32 carry = 0;
  for (int j = 0; j < 65; j++) {
    u32 v = (j < 64 ? (kw[j >> 3] >> ((j & 7) * 4)) & 15u : 0u) + carry;
    carry = 0;
    int d = (int)v;
    if (v > 8) { d = (int)v - 16; carry = 1; }
    digits[j] = d;
  }
  XYZZ<F> acc, t;
  xyzz_set_identity(acc);
  for (int j = 64; j >= 0; j--) {
    for (int b = 0; b < 4; b++) { xyzz_dbl(t, acc); acc = t; }
    const int d = digits[j];
    if (d) {
      XYZZ<F> q = tab[(d < 0 ? -d : d) - 1];
      if (d < 0) F::neg(q.y, q.y);
      xyzz_add(t, acc, q);
      acc = t;
    }
  }
  xyzz_to_affine(res, acc);
  memcpy(r, &res, sizeof res);
}
// sum_i [k_i] P_i on the host with ONE doubling chain (Straus, signed 4-bit windows per term) and one final inversion:
// the tail of create_proof folds g_c = ... + [s] a_answer + [r] b1_answer + h + l (prover.rs:339-354) - as separate
// scalar multiplications and affine additions that was two doubling chains and nine inversions.  Scalars equal to one
// cost a single addition.
template <class F>
static void point_lincomb(void *r, const void *pts, const u32 *scalars, u64 n) {
  struct Term {
    XYZZ<F> tab[8];
    int digits[65];
  };
  std::vector<Term> terms;
  std::vector<Affine<F>> ones;
  terms.reserve(n);
  for (u64 i = 0; i < n; i++) {
    Affine<F> base;
    memcpy(&base, (const char *)pts + i * sizeof base, sizeof base);
    u32 kw[9];
    if (scalars) memcpy(kw, scalars + 8 * i, 32); else { memset(kw, 0, 32); kw[0] = 1; }
    kw[8] = 0;
    bool zero = true, one = kw[0] == 1;
    for (int j = 0; j < 8; j++) { zero &= kw[j] == 0; if (j) one &= kw[j] == 0; }
    if (aff_is_identity(base) || zero) continue;
    if (one) { ones.push_back(base); continue; }
    terms.emplace_back();
    Term &t = terms.back();
    xyzz_from_affine(t.tab[0], base);
    xyzz_dbl(t.tab[1], t.tab[0]);
    for (int j = 2; j < 8; j++) xyzz_add(t.tab[j], t.tab[j - 1], t.tab[0]);
    u32 carry = 0;
    for (int j = 0; j < 65; j++) {
      u32 v = (j < 64 ? (kw[j >> 3] >> ((j & 7) * 4)) & 15u : 0u) + carry;
      carry = 0;
      int d = (int)v;
      if (v > 8) { d = (int)v - 16; carry = 1; }
      t.digits[j] = d;
    }
  }
  XYZZ<F> acc, tmp;
  xyzz_set_identity(acc);
  if (!terms.empty()) {
    for (int j = 64; j >= 0; j--) {
      if (!xyzz_is_identity(acc)) for (int b = 0; b < 4; b++) { xyzz_dbl(tmp, acc); acc = tmp; }
      for (Term &t : terms) {
        const int d = t.digits[j];
        if (!d) continue;
        XYZZ<F> q = t.tab[(d < 0 ? -d : d) - 1];
        if (d < 0) F::neg(q.y, q.y);
        xyzz_add(tmp, acc, q);
        acc = tmp;
      }
    }
  }
  for (const Affine<F> &b : ones) xyzz_madd(acc, b);
  Affine<F> res;
  xyzz_to_affine(res, acc);
  memcpy(r, &res, sizeof res);
}
// fast path: 64-bit-limb host arithmetic
template <class FD> static void host_point_lincomb_t(void *r, const void *pts, const u32 *scalars, u64 n) {
  point_lincomb<typename HostOf<FD>::type>(r, pts, scalars, n);
}
template <class FD> static void host_point_add_t(void *r, const void *a, const void *b, u64 n) {
  generic_point_add<typename HostOf<FD>::type>(r, a, b, n);
}
template <class FD> static void host_point_mul_t(void *r, const void *a, const u32 *k) {
  windowed_point_mul<typename HostOf<FD>::type>(r, a, k);
}
// the DEVICE headers compiled for the host (32-bit limbs): CPU-side unit tests of ff.cuh / ec.cuh
template <class F> static void devhdr_point_add_t(void *r, const void *a, const void *b, u64 n) {
  generic_point_add<F>(r, a, b, n);
}
template <class F> static void devhdr_point_mul_t(void *r, const void *a, const u32 *k) { generic_point_mul<F>(r, a, k); }

// OPS = the record format in memory (FpOps / Fp2Ops).  msm_enqueue_<group> itself (which kernel bundle accumulates,
// which one merges and reduces) is written out in msm_g1.hip / msm_g2.hip.
#define BH_INSTANTIATE_MSM_SUPPORT(SUFFIX, OPS)                                                               \
  int window_table_##SUFFIX(void *table_dev, u64 n, u32 c, u32 W, hipStream_t st) {                           \
    return window_table_t<OPS>(table_dev, n, c, W, st);                                                       \
  }                                                                                                           \
  int msm_finish_##SUFFIX(MsmJobImpl &job, void *out_affine, float *ms) {                                     \
    return msm_finish<OPS>(job, out_affine, ms);                                                              \
  }                                                                                                           \
  int fixed_base_mul_##SUFFIX(const void *base_host, const void *scalars_dev, u64 n, int fmt, void *out_dev,  \
                              hipStream_t st, void *table_dev) {                                              \
    return fixed_base_mul_t<OPS>(base_host, scalars_dev, n, fmt, out_dev, st, table_dev);                     \
  }                                                                                                           \
  int points_check_##SUFFIX(const void *pts_dev, u64 n, u32 *status_dev, hipStream_t st) {                    \
    return points_check_t<OPS>(pts_dev, n, status_dev, st);                                                   \
  }                                                                                                           \
  void host_point_add_##SUFFIX(void *r, const void *a, const void *b, u64 n) {                                \
    host_point_add_t<OPS>(r, a, b, n);                                                                        \
  }                                                                                                           \
  void host_point_mul_##SUFFIX(void *r, const void *a, const void *k) {                                       \
    host_point_mul_t<OPS>(r, a, (const u32 *)k);                                                              \
  }                                                                                                           \
  void host_point_lincomb_##SUFFIX(void *r, const void *pts, const void *scalars, u64 n) {                    \
    host_point_lincomb_t<OPS>(r, pts, (const u32 *)scalars, n);                                               \
  }

}  // namespace bh
