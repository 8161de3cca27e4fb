// Shared declarations of the MSM pipeline (see msm.hip for the overview).
#pragma once
#include "common.hpp"

namespace bh {

struct ErrFlags {       // device-side status word block
  u32 eof;              // a dense entry found the base cursor at/after the end (multiexp.rs:55-61,74-80)
  u32 ident;            // an identity base was consumed (multiexp.rs:63-65)
  u32 ident_top;        // ... in the reference's top window, before the first EOF entry
  u32 nbig;             // number of split buckets
  u32 total_tasks;
  u32 pad[3];
};

struct Task {
  u32 begin, end;  // [begin, end) in the sorted pair array (global, window-major)
  u32 dest;        // slot in pts[]: bucket index, or NB + task index for a split bucket
};
struct BigBucket {
  u32 bucket, first_task, ntasks;
};

enum { SUM_STRIDED = 1, SUM_BITS = 2 };
struct SumDesc {
  u32 mode;
  u32 groups;      // number of outputs
  u32 count;       // elements per group
  u32 inner;       // groups per outer index
  u32 stride;      // element stride inside a group
  u32 group_shift; // log2 of the elements spanned by one outer index
};

struct MsmPlan {
  u32 n, c, W, nb, NB, lo_bits, hi_bits, num_tiles, chunk, sort_passes;
  u64 max_tasks;
  u32 max_big;
};

struct MsmJobImpl {
  Context *ctx = nullptr;
  int group = BH_G1;
  hipStream_t stream = nullptr;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  hipEvent_t ev_sorted = nullptr, ev_accum = nullptr;   // stage boundaries (profiling)
  MsmPlan plan;
  std::vector<void *> dev_allocs;     // returned to the pool on wait
  void *host_result = nullptr;        // pinned: W*c XYZZ + ErrFlags
  size_t host_result_bytes = 0;
  int early_rc = BH_OK;               // immediate result (n == 0 etc.)
  bool trivial = false;
  // inputs needed again by the (rare) error-resolution pass
  const void *scalars_dev = nullptr;
  const u64 *density_dev = nullptr;
  const u32 *word_prefix = nullptr;
  const void *bases_dev = nullptr;
  u64 skip = 0, n_bases = 0;
  int fmt = 0;
  ErrFlags *err_dev = nullptr;
};


// msm_stages.hip: curve-independent stages (digits, sort, bounds, tasks)
struct MsmBuffers {
  u64 *pairs_a, *pairs_b;
  u32 *counts, *scan_tmp, *start, *task_off, *word_prefix;
  Task *tasks;
  BigBucket *big;
  ErrFlags *err;
};
MsmPlan make_plan(u64 n, unsigned forced_c);
size_t scan_tmp_elems(u64 n);
// runs stages 1-3 on `st`; *sorted_out = the sorted pair array (pairs_a or pairs_b)
int msm_run_stages(const MsmPlan &p, const MsmBuffers &b, const void *scalars_dev, int fmt, const u64 *density_dev,
                   u64 skip, u64 n_bases, hipStream_t st, const u64 **sorted_out);

// msm_g1.hip / msm_g2.hip
int msm_enqueue_g1(MsmJobImpl &job, const void *bases_dev, u64 n_bases, u64 skip, const void *scalars_dev, u64 n,
                   int fmt, const u64 *density_dev, unsigned forced_c);
int msm_enqueue_g2(MsmJobImpl &job, const void *bases_dev, u64 n_bases, u64 skip, const void *scalars_dev, u64 n,
                   int fmt, const u64 *density_dev, unsigned forced_c);
int msm_finish_g1(MsmJobImpl &job, void *out_affine, float *ms);   // ms: float[4] or null
int msm_finish_g2(MsmJobImpl &job, void *out_affine, float *ms);
int fixed_base_mul_g1(const void *base_host, const void *scalars_dev, u64 n, int fmt, void *out_dev, hipStream_t st);
int fixed_base_mul_g2(const void *base_host, const void *scalars_dev, u64 n, int fmt, void *out_dev, hipStream_t st);
int test_point_add_g1(void *r, const void *a, const void *b, u64 n, hipStream_t st);
int test_point_add_g2(void *r, const void *a, const void *b, u64 n, hipStream_t st);
void host_point_add_g1(void *r, const void *a, const void *b, u64 n);
void host_point_add_g2(void *r, const void *a, const void *b, u64 n);
void host_point_mul_g1(void *r, const void *a, const void *k);
void host_point_mul_g2(void *r, const void *a, const void *k);

}  // namespace bh
