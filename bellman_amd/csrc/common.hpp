// Shared host-side plumbing for libbellman_hip: error codes, context, workspace cache.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <condition_variable>
#include <functional>
#include <list>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <vector>

#include "../../include/bellman_hip.h"
struct bh_bases;
#include "ec.cuh"
#include "ff.cuh"

namespace bh {

#define BH_HIP_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      fprintf(stderr, "[bellman_hip] %s failed: %s (%s:%d)\n", #expr,                   \
              hipGetErrorString(_e), __FILE__, __LINE__);                               \
      return BH_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)

// Size-classed cache of device allocations: hipMalloc/hipFree synchronise the device, so the per-call
// workspaces of the (asynchronous, concurrent) MSM entry points are recycled.
//   * size classes at 1/8 octave (a request is rounded up by at most 12.5 %; power-of-two classes wasted up to
//     40 % of every workspace below 1 GiB);
//   * when hipMalloc fails, or an optional byte cap (bh_ctx_set_limits) would be exceeded, the idle blocks of
//     the OTHER classes are handed back to the driver and the allocation is retried;
//   * when that is not enough the context's pressure handler runs - it completes the oldest multiexp still in
//     flight (whose workspace then returns to the pool) - and the allocation is retried, until nothing is left in
//     flight.  This is the analogue of Worker::compute running a task inline once the queue is deep
//     (src/multicore.rs:47-73): a caller that issues more work than fits waits, it is not refused.
class DevicePool {
 public:
  ~DevicePool() { release_all(); }
  static size_t size_class(size_t bytes) {
    if (bytes < 256) return 256;
    int lg = 63 - __builtin_clzll((unsigned long long)bytes);
    const size_t step = size_t(1) << (lg - 3);
    return (bytes + step - 1) & ~(step - 1);
  }
  void set_cap(size_t bytes) { std::lock_guard<std::mutex> g(mu_); cap_ = bytes; }
  void set_pressure_handler(std::function<bool()> f) { pressure_ = std::move(f); }
  size_t bytes_held() { std::lock_guard<std::mutex> g(mu_); return held_; }
  size_t bytes_idle() { std::lock_guard<std::mutex> g(mu_); return idle_; }
  void *acquire(size_t bytes) {
    if (bytes == 0) bytes = 16;
    const size_t cap = size_class(bytes);
    for (;;) {
      bool over_cap = false;
      {
        std::lock_guard<std::mutex> g(mu_);
        auto it = free_.find(cap);
        if (it != free_.end() && !it->second.empty()) {
          void *p = it->second.back();
          it->second.pop_back();
          idle_ -= cap;
          live_[p] = cap;
          return p;
        }
        over_cap = cap_ && held_ + cap > cap_;
        if (!over_cap) held_ += cap;   // reserved before the (unlocked) hipMalloc so that concurrent callers see it
      }
      if (!over_cap) {
        void *p = nullptr;
        if (hipMalloc(&p, cap) == hipSuccess) {
          std::lock_guard<std::mutex> g(mu_);
          live_[p] = cap;
          return p;
        }
        (void)hipGetLastError();
        std::lock_guard<std::mutex> g(mu_);
        held_ -= cap;
      }
      if (trim_idle()) continue;                 // idle blocks of other classes went back to the driver
      if (pressure_ && pressure_()) continue;    // an in-flight job was completed: its workspace is back in the pool
      return nullptr;
    }
  }
  void release(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> g(mu_);
    auto it = live_.find(p);
    if (it == live_.end()) return;
    free_[it->second].push_back(p);
    idle_ += it->second;
    live_.erase(it);
  }
  // hands every idle block back to the driver; true if that freed anything
  bool trim_idle() {
    std::vector<void *> drop;
    {
      std::lock_guard<std::mutex> g(mu_);
      for (auto &kv : free_)
        for (void *p : kv.second) { drop.push_back(p); held_ -= kv.first; idle_ -= kv.first; }
      free_.clear();
    }
    for (void *p : drop) (void)hipFree(p);
    return !drop.empty();
  }
  void release_all() { (void)trim_idle(); }

 private:
  std::mutex mu_;
  std::map<size_t, std::vector<void *>> free_;
  std::map<void *, size_t> live_;
  size_t held_ = 0, idle_ = 0, cap_ = 0;   // bytes obtained from the driver / of those idle / optional cap (0 = none)
  std::function<bool()> pressure_;
};

// per-job HIP objects that are expensive to create: recycled across MSM jobs
struct JobResources {
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  // optional (BELLMAN_HIP_REDUCE_PRIORITY=1): the latency-bound merge / reduction kernels of the job on a
  // high-priority stream, ordered after the accumulation by an event
  hipStream_t hp_stream = nullptr;
  hipEvent_t hp_event = nullptr;
  hipEvent_t sort_event = nullptr; // lazily created: end of the digit / sort stage of a held job (BH_MSM_HOLD)
  hipEvent_t acc_event = nullptr;  // lazily created: recorded after the job's bucket accumulation launch (the accumulation chain)
  hipEvent_t dep_event = nullptr;  // lazily created: orders the job after another stream (bh_msm_async_dev_after)
  void *pinned = nullptr;          // host-pinned landing buffer for the job's result
  size_t pinned_bytes = 0;
};

// An Fr table entry pre-sliced for the multiplier (ff.cuh fe_mul_b): 9 limbs of 30 bits, padded to 48 bytes
struct alignas(16) BTw {
  u32 l[12];
};
// Per-size FFT tables (fft.hip), all in B form and two-level: entry e of a virtual n-entry table is
// hi[e >> lb] * lo[e & (2^lb - 1)].  [0] forward, [1] inverse.
struct FftTables {
  bool init = false;
  uint32_t lb = 0;
  BTw *tw_lo[2] = {nullptr, nullptr}, *tw_hi[2] = {nullptr, nullptr};   // omega_n^(+-e): inter-pass twiddles
  BTw *coset_lo = nullptr, *coset_hi = nullptr;                           // 7^i
  BTw *icoset_lo = nullptr, *icoset_hi = nullptr;                         // 7^-i / n
  fr_t minv;                                                              // n^-1 (Montgomery)
  fr_t zinv;                                                              // (7^n - 1)^-1: divide_by_z_on_coset
  BTw *minv_dev = nullptr;                                                // ... as a one-entry table
  // [r4] one-level tables in tile order (fft.hip; 2^12 .. 2^24 points): inter-pass twiddles per direction and non-last
  // pass - the inverse direction's first one carries the 1/n -, 7^i and 7^-i.  n entries each; [r5] an entry is the
  // 32-byte Montgomery element (the pointers are typed BTw* for the two-level tables they sit beside).
  bool one_level = false, one_level_failed = false;
  BTw *tw1[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  BTw *coset1 = nullptr, *icoset1 = nullptr;
  // [r5] what this size holds in device memory and when it was last used: the cache is bounded (Context::fft_table_budget),
  // least recently used sizes go first
  size_t bytes = 0;
  uint64_t last_use = 0;
};

struct Context {
  int device = 0;
  hipStream_t stream = nullptr;  // context stream for synchronous entry points
  DevicePool pool;
  std::mutex fft_mu;
  std::map<uint32_t, FftTables> fft_tables;  // keyed by log_n
  BTw *fft_master[2] = {nullptr, nullptr};   // omega_2048^(+-i), i < 1024: in-tile twiddles of every pass
  // [r5] The per-size tables are a cache with a budget (bh_ctx_set_limits; default an eighth of the device's memory): a
  // size whose complete one-level set would not fit runs on the small two-level tables, and building a table that does
  // not fit beside the others first drops the least recently used OTHER sizes.  Dropping needs the tables idle: a transform
  // holds fft_use_mu shared from get_tables to its last launch, the evictor takes it exclusively and synchronises the device.
  size_t fft_table_bytes = 0, fft_table_budget = 0;
  uint64_t fft_tick = 0;
  std::shared_mutex fft_use_mu;
  int num_cus = 256;
  std::mutex job_mu;
  std::vector<JobResources> job_pool;
  std::vector<hipStream_t> stream_pool;   // bh_stream_create / destroy recycle streams (creation costs ~1 ms)
  std::vector<hipStream_t> hp_stream_pool, hp_streams;   // the same for high-priority streams (bh_stream_create_priority)
  // multiexps issued and not yet completed, oldest first (guarded by job_mu).  Back-pressure (src/multicore.rs:47-73:
  // Worker::compute runs the task inline once 4 x threads are pending): when `max_jobs` are in flight, or the
  // workspace pool cannot serve an allocation, the issuing thread COMPLETES the oldest job itself (stream
  // synchronise + host tail; the result is kept in the job for its bh_msm_wait) instead of failing.
  std::list<struct MsmJobImpl *> inflight;
  uint32_t issuing = 0;           // slots reserved by calls that are between the cap check and msm_job_track (job_mu)
  uint32_t max_jobs = 64;
  size_t hbm_total = 0, table_bytes = 0, table_budget = 0;   // window tables built automatically stay below the budget
  std::vector<struct ::bh_bases *> tables;                     // handles that own an automatically built table (job_mu)
  // The accumulation chain: bucket-accumulation launches that fill the chip run one after the other in issue order
  // (each waits for the previous one's event) instead of sharing the SIMDs - two of them side by side take twice as
  // long each, so every job of a proof would finish late and all the latency-bound merge / reduction tails would pile up
  // at the end with the chip idle (profiles/archive/r3_call3_proof_timeline.txt).  Chained, job k's tail runs beside job k+1's
  // accumulation.  BELLMAN_HIP_ACC_CHAIN=0 switches it off.
  std::mutex acc_mu;
  hipEvent_t last_acc_event = nullptr;
  std::vector<hipEvent_t> barrier_events;   // events of bh_ctx_accumulations_after (a ring, recycled round robin)
  size_t barrier_next = 0;
  std::vector<hipEvent_t> pending_barriers; // ... not yet waited for by an accumulation
  int hw_queues_env = 0;          // GPU_MAX_HW_QUEUES seen when the context was created (0 = unset: the runtime's 4)
  bool configured_early = false;  // bh_runtime_configure ran before this library's first HIP call
};

}  // namespace bh

namespace bh {
void fft_tables_free(FftTables &t);   // fft.hip
void fft_master_free(Context &c);
}  // namespace bh

struct bh_ctx {
  bh::Context c;
};
