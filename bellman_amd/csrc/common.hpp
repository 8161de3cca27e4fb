// Shared host-side plumbing for libbellman_hip: error codes, context, workspace cache.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <map>
#include <mutex>
#include <vector>

#include "../../include/bellman_hip.h"
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

// Size-bucketed cache of device allocations: hipMalloc/hipFree synchronise the device, so
// the per-call workspaces of the (asynchronous, concurrent) MSM entry points are recycled.
class DevicePool {
 public:
  ~DevicePool() { release_all(); }
  void *acquire(size_t bytes) {
    if (bytes == 0) bytes = 16;
    size_t cap = 256;
    while (cap < bytes) cap <<= 1;
    if (cap > (size_t(1) << 30)) cap = (bytes + ((size_t(1) << 28) - 1)) & ~((size_t(1) << 28) - 1);
    {
      std::lock_guard<std::mutex> g(mu_);
      auto it = free_.find(cap);
      if (it != free_.end() && !it->second.empty()) {
        void *p = it->second.back();
        it->second.pop_back();
        live_[p] = cap;
        return p;
      }
    }
    void *p = nullptr;
    if (hipMalloc(&p, cap) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> g(mu_);
    live_[p] = cap;
    return p;
  }
  void release(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> g(mu_);
    auto it = live_.find(p);
    if (it == live_.end()) return;
    free_[it->second].push_back(p);
    live_.erase(it);
  }
  void release_all() {
    std::lock_guard<std::mutex> g(mu_);
    for (auto &kv : free_)
      for (void *p : kv.second) (void)hipFree(p);
    free_.clear();
  }

 private:
  std::mutex mu_;
  std::map<size_t, std::vector<void *>> free_;
  std::map<void *, size_t> live_;
};

// per-job HIP objects that are expensive to create: recycled across MSM jobs
struct JobResources {
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  // optional (BELLMAN_HIP_REDUCE_PRIORITY=1): the latency-bound merge / reduction kernels of the job on a
  // high-priority stream, ordered after the accumulation by an event
  hipStream_t hp_stream = nullptr;
  hipEvent_t hp_event = nullptr;
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
};

struct Context {
  int device = 0;
  hipStream_t stream = nullptr;  // context stream for synchronous entry points
  DevicePool pool;
  std::mutex fft_mu;
  std::map<uint32_t, FftTables> fft_tables;  // keyed by log_n
  BTw *fft_master[2] = {nullptr, nullptr};   // omega_2048^(+-i), i < 1024: in-tile twiddles of every pass
  int num_cus = 256;
  std::mutex job_mu;
  std::vector<JobResources> job_pool;
  std::vector<hipStream_t> stream_pool;   // bh_stream_create / destroy recycle streams (creation costs ~1 ms)
};

}  // namespace bh

namespace bh {
void fft_tables_free(FftTables &t);   // fft.hip
void fft_master_free(Context &c);
}  // namespace bh

struct bh_ctx {
  bh::Context c;
};
