// Curve-dependent part of the MSM pipeline (bucket accumulation, reductions, host tail),
// written once over the field-ops bundle and instantiated for G1 (msm_g1.hip) and G2
// (msm_g2.hip) in separate translation units so they compile in parallel.
#pragma once
#include <string.h>

#include <algorithm>
#include <cmath>

#include "msm_scalar.cuh"
#include "msm_types.hpp"

namespace bh {

// Only launched when both EOF and an identity were seen: decides which error the reference
// would report (the highest window's, i.e. the first failing element of the top window;
// multiexp.rs:295-300).  c_ref = reference window size, top window = bits [lo_ref, 256).
template <class F>
__global__ void msm_err_resolve_kernel(const void *scalars, int fmt, u32 n, const u64 *density,
                                       const u32 *word_prefix, u64 skip, u64 n_bases,
                                       const Affine<F> *bases, u32 lo_ref, ErrFlags *err) {
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
  if (aff_is_identity(bases[k])) atomicOr(&err->ident_top, 1u);
}

// ============================================================================================
// 4. bucket accumulation
// ============================================================================================
template <class F>
__global__ __launch_bounds__(128) void msm_accumulate_kernel(const u64 *pairs, const Task *tasks,
                                                             const Affine<F> *bases, XYZZ<F> *pts,
                                                             ErrFlags *err) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= err->total_tasks) return;
  const Task task = tasks[t];
  XYZZ<F> acc;
  xyzz_set_identity(acc);
  bool saw_identity = false;
  for (u32 p = task.begin; p < task.end; p++) {
    const u32 idx = (u32)pairs[p];
    const Affine<F> q = bases[idx];
    if (aff_is_identity(q)) { saw_identity = true; continue; }
    xyzz_madd(acc, q);
  }
  if (saw_identity) atomicOr(&err->ident, 1u);
  pts[task.dest] = acc;
}

// ============================================================================================
// 5. reductions: one wavefront per output point
// ============================================================================================
// tree-reduce the 64 per-lane accumulators of a one-wave workgroup through LDS
template <class F>
__device__ __forceinline__ void wave_reduce_points(XYZZ<F> &acc, XYZZ<F> *slots) {
  const u32 lane = threadIdx.x;
  for (u32 off = 32; off >= 1; off >>= 1) {
    if (lane >= off && lane < 2 * off) slots[lane] = acc;
    __syncthreads();
    if (lane < off) {
      XYZZ<F> o = slots[lane + off];
      XYZZ<F> r;
      xyzz_add(r, acc, o);
      acc = r;
    }
    __syncthreads();
  }
}


// out[g] = sum of a set of in[] points chosen by the mode:
//   SUM_STRIDED: g = (outer, innerIdx): elements in[(outer << group_shift) + innerIdx*istride + t*stride]
//   SUM_BITS   : g = (outer, k): elements in[(outer << group_shift) + i], i < count, bit k of i set
template <class F>
__global__ __launch_bounds__(64) void msm_sum_kernel(const XYZZ<F> *in, XYZZ<F> *out, SumDesc d, u32 istride) {
  __shared__ XYZZ<F> slots[64];
  const u32 lane = threadIdx.x;
  const u32 g = blockIdx.x;
  XYZZ<F> acc;
  xyzz_set_identity(acc);
  const u32 outer = g / d.inner, in_idx = g % d.inner;
  const XYZZ<F> *base = in + ((u64)outer << d.group_shift);
  if (d.mode == SUM_STRIDED) {
    for (u32 t = lane; t < d.count; t += 64) {
      XYZZ<F> o = base[(u64)in_idx * istride + (u64)t * d.stride];
      XYZZ<F> r;
      xyzz_add(r, acc, o);
      acc = r;
    }
  } else {  // SUM_BITS: in_idx = bit position k
    for (u32 i = lane; i < d.count; i += 64) {
      if ((i >> in_idx) & 1) {
        XYZZ<F> o = base[i];
        XYZZ<F> r;
        xyzz_add(r, acc, o);
        acc = r;
      }
    }
  }
  wave_reduce_points<F>(acc, slots);
  if (lane == 0) out[g] = acc;
}

// merge the partial sums of split buckets back into their bucket slot
template <class F>
__global__ __launch_bounds__(64) void msm_merge_big_kernel(XYZZ<F> *pts, const BigBucket *big, const ErrFlags *err,
                                                          u32 NB, u32 max_big) {
  __shared__ XYZZ<F> slots[64];
  const u32 lane = threadIdx.x;
  u32 nbig = err->nbig;
  if (nbig > max_big) nbig = max_big;
  for (u32 e = blockIdx.x; e < nbig; e += gridDim.x) {
    const BigBucket bb = big[e];
    XYZZ<F> acc;
    xyzz_set_identity(acc);
    for (u32 t = lane; t < bb.ntasks; t += 64) {
      XYZZ<F> o = pts[(u64)NB + bb.first_task + t];
      XYZZ<F> r;
      xyzz_add(r, acc, o);
      acc = r;
    }
    wave_reduce_points<F>(acc, slots);
    if (lane == 0) pts[bb.bucket] = acc;
    __syncthreads();
  }
}

// ============================================================================================
// fixed-base scalar multiplication (fixture generation) and test hooks
// ============================================================================================
template <class F>
__global__ void fixed_base_mul_kernel(Affine<F> base, const void *scalars, int fmt, u64 n, Affine<F> *out) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fr_t s;
  load_scalar(scalars, i, fmt, s);
  XYZZ<F> acc;
  xyzz_set_identity(acc);
  for (int b = 254; b >= 0; b--) {
    XYZZ<F> t;
    xyzz_dbl(t, acc);
    acc = t;
    if ((s.l[b >> 5] >> (b & 31)) & 1) xyzz_madd(acc, base);
  }
  Affine<F> r;
  xyzz_to_affine(r, acc);
  out[i] = r;
}
template <class F>
__global__ void point_add_kernel(Affine<F> *r, const Affine<F> *a, const Affine<F> *b, u64 n) {
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
// ============================================================================================
// host orchestration
// ============================================================================================
template <class F>
static int msm_enqueue(MsmJobImpl &job, const void *bases_dev, u64 n_bases, u64 skip, const void *scalars_dev,
                       u64 n, int fmt, const u64 *density_dev, unsigned forced_c) {
  Context &c = *job.ctx;
  hipStream_t st = job.stream;
  const MsmPlan p = make_plan(n, forced_c);
  job.plan = p;
  if ((u64)p.W * n >= ((u64)1 << 32)) return BH_ERR_INVALID_ARG;  // pair positions are 32-bit
  auto alloc = [&](size_t bytes) -> void * {
    void *ptr = c.pool.acquire(bytes);
    if (ptr) job.dev_allocs.push_back(ptr);
    return ptr;
  };
  const u64 npairs = (u64)p.W * n;
  const u64 ncounts = (u64)p.W * 256 * p.num_tiles;
  MsmBuffers b;
  b.pairs_a = (u64 *)alloc(npairs * 8);
  b.pairs_b = (u64 *)alloc(npairs * 8);
  b.counts = (u32 *)alloc(ncounts * 4);
  b.scan_tmp = (u32 *)alloc(scan_tmp_elems(std::max<u64>(ncounts, p.NB + 1)) * 4);
  b.start = (u32 *)alloc((u64)p.W * (p.nb + 1) * 4);
  b.task_off = (u32 *)alloc(((u64)p.NB + 1) * 4);
  b.tasks = (Task *)alloc(p.max_tasks * sizeof(Task));
  b.big = (BigBucket *)alloc((u64)p.max_big * sizeof(BigBucket));
  XYZZ<F> *pts = (XYZZ<F> *)alloc(((u64)p.NB + p.max_tasks) * sizeof(XYZZ<F>));
  const u32 H = 1u << p.hi_bits, Lw = 1u << p.lo_bits;
  XYZZ<F> *rowcol = (XYZZ<F> *)alloc((u64)p.W * (H + Lw) * sizeof(XYZZ<F>));
  XYZZ<F> *bits = (XYZZ<F> *)alloc((u64)p.W * p.c * sizeof(XYZZ<F>));
  b.err = (ErrFlags *)alloc(sizeof(ErrFlags));
  b.word_prefix = nullptr;
  const u64 nwords = (n + 63) / 64;
  if (density_dev) b.word_prefix = (u32 *)alloc((nwords + 1) * 4);
  if (!b.pairs_a || !b.pairs_b || !b.counts || !b.scan_tmp || !b.start || !b.task_off || !b.tasks || !b.big ||
      !pts || !rowcol || !bits || !b.err || (density_dev && !b.word_prefix))
    return BH_ERR_HIP;
  ErrFlags *err = b.err;
  job.err_dev = err;
  job.scalars_dev = scalars_dev; job.density_dev = density_dev; job.word_prefix = b.word_prefix;
  job.bases_dev = bases_dev; job.skip = skip; job.n_bases = n_bases; job.fmt = fmt;

  BH_HIP_CHECK(hipEventRecord(job.ev_begin, st));
  BH_HIP_CHECK(hipMemsetAsync(err, 0, sizeof(ErrFlags), st));
  BH_HIP_CHECK(hipMemsetAsync(pts, 0, (u64)p.NB * sizeof(XYZZ<F>), st));   // all-zero XYZZ == identity
  const u64 *sorted = nullptr;
  {
    int rc = msm_run_stages(p, b, scalars_dev, fmt, density_dev, skip, n_bases, st, &sorted);
    if (rc) return rc;
  }
  Task *tasks = b.tasks;
  BigBucket *big = b.big;
  BH_HIP_CHECK(hipEventRecord(job.ev_sorted, st));
  // 4. accumulate
  hipLaunchKernelGGL(msm_accumulate_kernel<F>, dim3((u32)((p.max_tasks + 127) / 128)), dim3(128), 0, st, sorted,
                     tasks, (const Affine<F> *)bases_dev, pts, err);
  BH_HIP_CHECK(hipGetLastError());
  BH_HIP_CHECK(hipEventRecord(job.ev_accum, st));   // brackets exactly the accumulate launch
  hipLaunchKernelGGL(msm_merge_big_kernel<F>, dim3(256), dim3(64), 0, st, pts, big, err, p.NB, p.max_big);
  BH_HIP_CHECK(hipGetLastError());
  // 5. reduce: rows (sum over lo, contiguous), columns (sum over hi, stride Lw), then bits
  XYZZ<F> *rows = rowcol, *cols = rowcol + (u64)p.W * H;
  {
    SumDesc d;
    d.mode = SUM_STRIDED; d.groups = p.W * H; d.count = Lw; d.inner = H; d.stride = 1; d.group_shift = p.c;
    hipLaunchKernelGGL(msm_sum_kernel<F>, dim3(d.groups), dim3(64), 0, st, pts, rows, d, Lw);
    BH_HIP_CHECK(hipGetLastError());
    d.groups = p.W * Lw; d.count = H; d.inner = Lw; d.stride = Lw;
    hipLaunchKernelGGL(msm_sum_kernel<F>, dim3(d.groups), dim3(64), 0, st, pts, cols, d, 1u);
    BH_HIP_CHECK(hipGetLastError());
    // U[w][p]: p < lo_bits from the column sums (weights lo), p >= lo_bits from the row sums (weights hi)
    d.mode = SUM_BITS; d.stride = 1;
    d.groups = p.W * p.lo_bits; d.count = Lw; d.inner = p.lo_bits; d.group_shift = p.lo_bits;
    if (d.groups) {
      hipLaunchKernelGGL(msm_sum_kernel<F>, dim3(d.groups), dim3(64), 0, st, cols, bits, d, 0u);
      BH_HIP_CHECK(hipGetLastError());
    }
    d.groups = p.W * p.hi_bits; d.count = H; d.inner = p.hi_bits; d.group_shift = p.hi_bits;
    hipLaunchKernelGGL(msm_sum_kernel<F>, dim3(d.groups), dim3(64), 0, st, rows, bits + (u64)p.W * p.lo_bits, d, 0u);
    BH_HIP_CHECK(hipGetLastError());
  }
  BH_HIP_CHECK(hipEventRecord(job.ev_end, st));
  // results to pinned host memory
  const size_t bits_bytes = (size_t)p.W * p.c * sizeof(XYZZ<F>);
  job.host_result_bytes = bits_bytes + sizeof(ErrFlags);
  BH_HIP_CHECK(hipHostMalloc(&job.host_result, job.host_result_bytes, hipHostMallocDefault));
  BH_HIP_CHECK(hipMemcpyAsync(job.host_result, bits, bits_bytes, hipMemcpyDeviceToHost, st));
  BH_HIP_CHECK(hipMemcpyAsync((char *)job.host_result + bits_bytes, err, sizeof(ErrFlags), hipMemcpyDeviceToHost, st));
  return BH_OK;
}

// host tail: result = sum_w sum_p 2^(c*w + p) U[w][p]
//   bits layout: [W][lo_bits] column-bit sums, then [W][hi_bits] row-bit sums
template <class F>
static void msm_host_tail(const MsmPlan &p, const XYZZ<F> *bits, Affine<F> *out) {
  XYZZ<F> acc;
  xyzz_set_identity(acc);
  const XYZZ<F> *lo = bits, *hi = bits + (size_t)p.W * p.lo_bits;
  for (int w = (int)p.W - 1; w >= 0; w--) {
    for (int b = (int)p.c - 1; b >= 0; b--) {
      XYZZ<F> t;
      xyzz_dbl(t, acc);
      const XYZZ<F> &u = (b >= (int)p.lo_bits) ? hi[(size_t)w * p.hi_bits + (b - p.lo_bits)]
                                               : lo[(size_t)w * p.lo_bits + b];
      xyzz_add(acc, t, u);
    }
  }
  xyzz_to_affine(*out, acc);
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
  if (hipStreamSynchronize(job.stream) != hipSuccess) rc = BH_ERR_HIP;
  if (rc == BH_OK) {
    const MsmPlan &p = job.plan;
    const size_t bits_bytes = (size_t)p.W * p.c * sizeof(XYZZ<F>);
    ErrFlags ef;
    memcpy(&ef, (char *)job.host_result + bits_bytes, sizeof ef);
    if (ms) {  // [0] whole device pipeline, [1] digits+sort+tasks, [2] bucket accumulation, [3] reductions
      (void)hipEventElapsedTime(&ms[0], job.ev_begin, job.ev_end);
      (void)hipEventElapsedTime(&ms[1], job.ev_begin, job.ev_sorted);
      (void)hipEventElapsedTime(&ms[2], job.ev_sorted, job.ev_accum);
      (void)hipEventElapsedTime(&ms[3], job.ev_accum, job.ev_end);
    }
    if (ef.eof && ef.ident) {
      // both kinds of failure exist: the reference reports the top window's first failure
      const double cref = (p.n < 32) ? 3.0 : std::ceil(std::log((double)p.n));   // multiexp.rs:318-322
      const u32 c_ref = (u32)cref, w_ref = (255 + c_ref - 1) / c_ref, lo_ref = c_ref * (w_ref - 1);
      hipLaunchKernelGGL(msm_err_resolve_kernel<F>, dim3((p.n + 255) / 256), dim3(256), 0, job.stream,
                         job.scalars_dev, job.fmt, p.n, job.density_dev, job.word_prefix, job.skip, job.n_bases,
                         (const Affine<F> *)job.bases_dev, lo_ref, job.err_dev);
      if (hipMemcpyAsync(&ef, job.err_dev, sizeof ef, hipMemcpyDeviceToHost, job.stream) != hipSuccess ||
          hipStreamSynchronize(job.stream) != hipSuccess)
        rc = BH_ERR_HIP;
      else
        rc = ef.ident_top ? BH_ERR_UNEXPECTED_IDENTITY : BH_ERR_UNEXPECTED_EOF;
    } else if (ef.eof) {
      rc = BH_ERR_UNEXPECTED_EOF;
    } else if (ef.ident) {
      rc = BH_ERR_UNEXPECTED_IDENTITY;
    } else {
      msm_host_tail<F>(p, (const XYZZ<F> *)job.host_result, (Affine<F> *)out_affine);
    }
  }
  for (void *ptr : job.dev_allocs) c.pool.release(ptr);
  job.dev_allocs.clear();
  if (job.host_result) (void)hipHostFree(job.host_result);
  job.host_result = nullptr;
  return rc;
}


template <class F>
static int fixed_base_mul_t(const void *base_host, const void *scalars_dev, u64 n, int fmt, void *out_dev,
                            hipStream_t st) {
  const u32 blocks = (u32)((n + 127) / 128);
  if (!blocks) return BH_OK;
  Affine<F> b;
  memcpy(&b, base_host, sizeof b);
  hipLaunchKernelGGL(fixed_base_mul_kernel<F>, dim3(blocks), dim3(128), 0, st, b, scalars_dev, fmt, n,
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
template <class F>
static void host_point_add_t(void *r, const void *a, const void *b, u64 n) {
  for (u64 i = 0; i < n; i++) {
    XYZZ<F> x, y, z;
    xyzz_from_affine(x, ((const Affine<F> *)a)[i]);
    xyzz_from_affine(y, ((const Affine<F> *)b)[i]);
    xyzz_add(z, x, y);
    xyzz_to_affine(((Affine<F> *)r)[i], z);
  }
}
template <class F>
static void host_point_mul_t(void *r, const void *a, const u32 *k) {
  XYZZ<F> acc;
  xyzz_set_identity(acc);
  const Affine<F> &base = *(const Affine<F> *)a;
  for (int b = 255; b >= 0; b--) {
    XYZZ<F> t;
    xyzz_dbl(t, acc);
    acc = t;
    if (((k[b >> 5] >> (b & 31)) & 1) && !aff_is_identity(base)) xyzz_madd(acc, base);
  }
  xyzz_to_affine(*(Affine<F> *)r, acc);
}

#define BH_INSTANTIATE_MSM(SUFFIX, OPS)                                                                       \
  int msm_enqueue_##SUFFIX(MsmJobImpl &job, const void *bases_dev, u64 n_bases, u64 skip,                    \
                           const void *scalars_dev, u64 n, int fmt, const u64 *density_dev, unsigned fc) {   \
    return msm_enqueue<OPS>(job, bases_dev, n_bases, skip, scalars_dev, n, fmt, density_dev, fc);             \
  }                                                                                                           \
  int msm_finish_##SUFFIX(MsmJobImpl &job, void *out_affine, float *ms) {                                     \
    return msm_finish<OPS>(job, out_affine, ms);                                                              \
  }                                                                                                           \
  int fixed_base_mul_##SUFFIX(const void *base_host, const void *scalars_dev, u64 n, int fmt, void *out_dev,  \
                              hipStream_t st) {                                                               \
    return fixed_base_mul_t<OPS>(base_host, scalars_dev, n, fmt, out_dev, st);                                \
  }                                                                                                           \
  int test_point_add_##SUFFIX(void *r, const void *a, const void *b, u64 n, hipStream_t st) {                 \
    return test_point_add_t<OPS>(r, a, b, n, st);                                                             \
  }                                                                                                           \
  void host_point_add_##SUFFIX(void *r, const void *a, const void *b, u64 n) {                                \
    host_point_add_t<OPS>(r, a, b, n);                                                                        \
  }                                                                                                           \
  void host_point_mul_##SUFFIX(void *r, const void *a, const void *k) {                                       \
    host_point_mul_t<OPS>(r, a, (const u32 *)k);                                                              \
  }

}  // namespace bh
