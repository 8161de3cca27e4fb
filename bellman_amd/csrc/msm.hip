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
//                  ranking inside a tile uses wavefront ballots (match-any) + popcounts.
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
#include <string.h>

#include <thread>
#include <vector>

#include "msm_types.hpp"

namespace bh {

// ---- job management / dispatch (used by api.hip) ------------------------------------------------
constexpr size_t JOB_PINNED_BYTES = 256 * 1024;   // >= W*c*sizeof(XYZZ<Fp2>) + flags for every plan

MsmJobImpl *msm_job_new(Context *ctx, int group) {
  MsmJobImpl *j = new MsmJobImpl();
  j->ctx = ctx;
  j->group = group;
  {
    std::lock_guard<std::mutex> g(ctx->job_mu);
    if (!ctx->job_pool.empty()) {
      j->res = ctx->job_pool.back();
      ctx->job_pool.pop_back();
    }
  }
  if (!j->res.stream) {
    bool ok = hipStreamCreateWithFlags(&j->res.stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 4 && ok; i++) ok = hipEventCreate(&j->res.ev[i]) == hipSuccess;
    if (ok) ok = hipHostMalloc(&j->res.pinned, JOB_PINNED_BYTES, hipHostMallocDefault) == hipSuccess;
    j->res.pinned_bytes = JOB_PINNED_BYTES;
    if (!ok) { delete j; return nullptr; }
  }
  static const bool reduce_priority = [] { const char *e = getenv("BELLMAN_HIP_REDUCE_PRIORITY"); return e && *e == '1'; }();
  if (reduce_priority && !j->res.hp_stream) {
    int lo = 0, hi = 0;   // numerically lower = higher priority
    bool ok = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess &&
              hipStreamCreateWithPriority(&j->res.hp_stream, hipStreamNonBlocking, hi) == hipSuccess &&
              hipEventCreateWithFlags(&j->res.hp_event, hipEventDisableTiming) == hipSuccess;
    if (!ok) { delete j; return nullptr; }
  }
  j->hp_stream = reduce_priority ? j->res.hp_stream : nullptr;
  j->hp_event = j->res.hp_event;
  j->stream = j->res.stream;
  j->ev_begin = j->res.ev[0]; j->ev_sorted = j->res.ev[1]; j->ev_accum = j->res.ev[2]; j->ev_end = j->res.ev[3];
  j->host_result = j->res.pinned;
  return j;
}
void msm_job_delete(MsmJobImpl *j) {
  if (!j) return;
  for (void *ptr : j->dev_allocs) j->ctx->pool.release(ptr);
  {
    std::lock_guard<std::mutex> g(j->ctx->job_mu);
    j->ctx->job_pool.push_back(j->res);   // (res.dep_event travels with the recycled resources)
  }
  delete j;
}
hipStream_t msm_job_stream(MsmJobImpl &job) { return job.stream; }
// everything enqueued on `after` so far happens before anything the job enqueues from now on
int msm_job_after(MsmJobImpl &job, hipStream_t after) {
  if (!job.res.dep_event) BH_HIP_CHECK(hipEventCreateWithFlags(&job.res.dep_event, hipEventDisableTiming));
  BH_HIP_CHECK(hipEventRecord(job.res.dep_event, after));
  BH_HIP_CHECK(hipStreamWaitEvent(job.stream, job.res.dep_event, 0));
  return BH_OK;
}
// a job whose answer is already known (computed on the host at issue time)
void msm_job_set_result(MsmJobImpl &job, int rc, const void *affine_record) {
  job.trivial = true;
  job.early_rc = rc;
  job.has_result = affine_record != nullptr;
  if (affine_record) memcpy(job.result, affine_record, job.group == BH_G1 ? 96 : 192);
}
void msm_job_own(MsmJobImpl &job, void *dev_ptr) { job.dev_allocs.push_back(dev_ptr); }
int msm_job_enqueue(MsmJobImpl &job, const void *bases_dev, u64 n_bases, u64 skip, const void *scalars_dev, u64 n,
                    int fmt, const u64 *density_dev, const MsmOpts &opts, const WindowTable *table) {
  if (n == 0) { job.trivial = true; job.early_rc = BH_OK; return BH_OK; }
  return job.group == BH_G1
             ? msm_enqueue_g1(job, bases_dev, n_bases, skip, scalars_dev, n, fmt, density_dev, opts, table)
             : msm_enqueue_g2(job, bases_dev, n_bases, skip, scalars_dev, n, fmt, density_dev, opts, table);
}
int window_table(int group, void *table_dev, u64 n, u32 c, u32 W, hipStream_t st) {
  return group == BH_G1 ? window_table_g1(table_dev, n, c, W, st) : window_table_g2(table_dev, n, c, W, st);
}
// runs the group's finish (stream synchronise, error resolution, host tail) once; caller holds job.mu
static void msm_job_complete_locked(MsmJobImpl &job) {
  if (job.done) return;
  // whoever completes the job - its waiter, the wait of a sharded multiexp (one job per GPU), or another issuing thread
  // under back-pressure - may have another device current: the (rare) error-resolution launch and its copy go to the
  // job's stream and have to be issued with the job's device current
  (void)hipSetDevice(job.ctx->device);
  job.done_rc = job.group == BH_G1 ? msm_finish_g1(job, job.done_out, job.done_ms) : msm_finish_g2(job, job.done_out, job.done_ms);
  job.done = true;
}
// reserves a slot under the context's job cap: true = reserved (release it with msm_job_track / msm_slot_release)
bool msm_slot_try_reserve(Context &c) {
  std::lock_guard<std::mutex> g(c.job_mu);
  if (c.inflight.size() + c.issuing >= c.max_jobs) return false;
  c.issuing++;
  return true;
}
void msm_slot_release(Context &c) {
  std::lock_guard<std::mutex> g(c.job_mu);
  if (c.issuing) c.issuing--;
}
// the issued job takes the slot its caller reserved
void msm_job_track(MsmJobImpl &job) {
  std::lock_guard<std::mutex> g(job.ctx->job_mu);
  if (job.ctx->issuing) job.ctx->issuing--;
  if (!job.trivial) job.ctx->inflight.push_back(&job);
}
static void msm_job_untrack(MsmJobImpl &job) {
  std::lock_guard<std::mutex> g(job.ctx->job_mu);
  job.ctx->inflight.remove(&job);
}
// starts a job that was issued with BH_MSM_HOLD (no-op otherwise)
int msm_job_start(MsmJobImpl &job) {
  std::lock_guard<std::mutex> g(job.mu);
  if (job.done || !job.resume) return BH_OK;
  const int rc = job.resume();
  job.resume = nullptr;
  return rc;
}
size_t msm_jobs_in_flight(Context &c) {
  std::lock_guard<std::mutex> g(c.job_mu);
  return c.inflight.size();
}
// Back-pressure: completes the oldest job in flight on the calling thread.  true = a job was completed, or one is
// being completed by another thread right now (the caller should retry what it was doing); false = nothing in flight.
bool msm_complete_oldest(Context &c) {
  MsmJobImpl *pick = nullptr;
  bool busy = false;
  {
    std::lock_guard<std::mutex> g(c.job_mu);
    for (MsmJobImpl *j : c.inflight) {
      if (j->mu.try_lock()) { pick = j; break; }   // held across job_mu's release: the waiter of `j` cannot delete it
      busy = true;
    }
  }
  if (!pick) {
    if (busy) std::this_thread::yield();
    return busy;
  }
  msm_job_complete_locked(*pick);
  msm_job_untrack(*pick);   // lock order mu -> job_mu; the scan above only try_locks mu under job_mu
  pick->mu.unlock();
  return true;
}
int msm_job_finish(MsmJobImpl &job, void *out_affine, float *ms, u64 *stats8) {
  if (job.trivial) {
    const size_t rec = job.group == BH_G1 ? 96 : 192;
    if (job.has_result) memcpy(out_affine, job.result, rec); else memset(out_affine, 0, rec);
    if (ms) ms[0] = ms[1] = ms[2] = ms[3] = 0.f;
    if (stats8) memset(stats8, 0, 8 * sizeof(u64));
    return job.early_rc;
  }
  msm_job_untrack(job);   // first: a back-pressure scan that has not picked the job yet will never see it
  std::lock_guard<std::mutex> g(job.mu);   // ... and one that has finishes before we look
  msm_job_complete_locked(job);
  memcpy(out_affine, job.done_out, job.group == BH_G1 ? 96 : 192);
  if (ms) memcpy(ms, job.done_ms, sizeof job.done_ms);
  if (stats8) memcpy(stats8, job.done_stats, sizeof job.done_stats);
  return job.done_rc;
}
// bring-up / verification aid (bh_msm_debug_stages): stages 1-3 only - signed digits, radix sort, zero-digit count -
// with the sorted (digit, base) pairs and the per-window zero counts copied to the host
int msm_debug_stages(Context &c, const void *scalars_host, u64 n, int fmt, unsigned cbits, u64 *pairs_out, u32 *zstart_out) {
  const MsmPlan p = make_plan(n, cbits, 0, false);
  hipStream_t st = c.stream;
  const u64 npairs = (u64)p.W * n, ncounts = (u64)p.W * 256 * p.num_tiles;
  MsmBuffers b;
  std::vector<void *> owned;
  auto alloc = [&](size_t bytes) { void *q = c.pool.acquire(bytes); owned.push_back(q); return q; };
  void *sc = alloc(n * 32);
  b.pairs_a = (u64 *)alloc(npairs * 8);
  b.pairs_b = (u64 *)alloc(npairs * 8);
  b.counts = (u32 *)alloc((ncounts + 1) * 4);
  b.scan_tmp = (u32 *)alloc(scan_tmp_elems(ncounts + 1) * 4);
  b.zstart = (u32 *)alloc((u64)p.W * 4);
  b.err = (ErrFlags *)alloc(sizeof(ErrFlags));
  b.word_prefix = nullptr;
  int rc = BH_OK;
  for (void *q : owned) if (!q) rc = BH_ERR_HIP;
  const u64 *sorted = nullptr;
  if (rc == BH_OK && hipMemcpyAsync(sc, scalars_host, n * 32, hipMemcpyHostToDevice, st) != hipSuccess) rc = BH_ERR_HIP;
  if (rc == BH_OK && hipMemsetAsync(b.err, 0, sizeof(ErrFlags), st) != hipSuccess) rc = BH_ERR_HIP;
  if (rc == BH_OK) rc = msm_run_stages(p, b, sc, fmt, nullptr, 0, n, st, &sorted);
  if (rc == BH_OK && hipMemcpyAsync(pairs_out, sorted, npairs * 8, hipMemcpyDeviceToHost, st) != hipSuccess) rc = BH_ERR_HIP;
  if (rc == BH_OK && hipMemcpyAsync(zstart_out, b.zstart, (u64)p.W * 4, hipMemcpyDeviceToHost, st) != hipSuccess) rc = BH_ERR_HIP;
  if (hipStreamSynchronize(st) != hipSuccess && rc == BH_OK) rc = BH_ERR_HIP;
  for (void *q : owned) if (q) c.pool.release(q);
  return rc;
}
int fixed_base_mul(int group, const void *base_host, const void *scalars_dev, u64 n, int fmt, void *out_dev,
                   hipStream_t st, void *table_dev) {
  return group == BH_G1 ? fixed_base_mul_g1(base_host, scalars_dev, n, fmt, out_dev, st, table_dev)
                        : fixed_base_mul_g2(base_host, scalars_dev, n, fmt, out_dev, st, table_dev);
}
int points_check(int group, const void *pts_dev, u64 n, u32 *status_dev, hipStream_t st) {
  return group == BH_G1 ? points_check_g1(pts_dev, n, status_dev, st) : points_check_g2(pts_dev, n, status_dev, st);
}
void host_point_add(int group, void *r, const void *a, const void *b, u64 n) {
  if (group == BH_G1) host_point_add_g1(r, a, b, n); else host_point_add_g2(r, a, b, n);
}
void host_point_mul(int group, void *r, const void *a, const void *k) {
  if (group == BH_G1) host_point_mul_g1(r, a, k); else host_point_mul_g2(r, a, k);
}
void host_point_lincomb(int group, void *r, const void *pts, const void *scalars, u64 n) {
  if (group == BH_G1) host_point_lincomb_g1(r, pts, scalars, n); else host_point_lincomb_g2(r, pts, scalars, n);
}
}  // namespace bh
