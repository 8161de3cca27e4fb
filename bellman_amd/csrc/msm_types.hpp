// Shared declarations of the MSM pipeline (see msm.hip for the overview).
#pragma once
#include "common.hpp"

namespace bh {

struct ErrFlags {       // device-side status word block
  u32 eof;              // a dense entry found the base cursor at/after the end (multiexp.rs:55-61,74-80)
  u32 ident;            // an identity base was consumed (multiexp.rs:63-65)
  u32 ident_top;        // ... in the reference's top window, before the first EOF entry
  u32 nlong;            // number of bucket runs queued for the wavefront-parallel merge
  u32 nbig;             // ... of those, runs long enough to be passed on to the workgroup-parallel merge
  u32 npieces;          // ... and the workgroup-sized pieces they were cut into (msm_merge_tail_kernel)
  // what the job executed (bh_msm_wait_stats): mixed additions into a non-empty accumulator by the accumulate launch
  // (entries that OPEN a bucket or a chunk partial are copies, ec.cuh xyzz_madd), zero digits the sort moved to the front
  unsigned long long madds, zeros;
};

enum { SUM_STRIDED = 1, SUM_BITS = 2 };
struct SumDesc {
  u32 mode;
  u32 groups;      // number of outputs
  u32 count;       // elements per group
  u32 inner;       // groups per outer index
  u32 stride;      // element stride inside a group
  u32 group_shift; // log2 of the elements spanned by one outer index
  u32 istride;     // STRIDED: element offset between consecutive inner indices
  u32 lanes;       // G: lanes cooperating on one output (power of two <= 64)
  u32 splits = 1;  // STRIDED: every group is cut into `splits` equal pieces of count / splits consecutive elements, each an
                   // output of its own (out[group * splits + piece]); `groups` then counts the pieces
};
struct LongRun {   // a bucket run that spans more chunks than its owner lane folds itself
  u32 w, lane, d;  // window, first chunk (holds the run's tail partial), digit
  u32 last;        // last chunk with a head partial of the run
};
// A run too long for a handful of lanes (boolean-heavy witnesses put a quarter of the scalars into bucket 1 of window 0 -
// the reason the reference has Exponent::One, multiexp.rs:172-182,245-252): its partials are cut into pieces of one
// workgroup's worth, pieces [piece0, piece0 + npieces) of the job; the workgroup that finishes the run's last piece folds
// the piece results (`done` counts finished pieces).
struct BigRun {
  u32 w, lane, d, last;
  u32 piece0, npieces, done, pad;
};

// per-job plan overrides (bh_msm_opts): zero = tuned default
struct MsmOpts {
  u32 c = 0;        // window bits
  u32 chunk = 0;    // K
  u32 flags = 0;    // BH_MSM_*
  // a shard of a multi-context multiexp (bh_msm_sharded_async): the reference's window size follows the TOTAL
  // number of exponents (multiexp.rs:318-322), and the shard reports whether an identity base was consumed in that
  // top window even when it saw no EOF itself - the fold needs it for the error precedence
  u64 ref_n = 0;
  bool always_resolve_ident = false;
  const void *padded_bases = nullptr;   // the same G1 records at a 128-byte stride (api.hip bh_bases::padded); null: none
  // a G1 window table whose records sit at a 128-byte stride (api.hip bh_bases::table_padded): only the bucket
  // accumulation of the table plan reads it; `bases_dev` of such a job is the dense base vector itself
  const void *padded_table = nullptr;
};

struct MsmPlan {
  u32 n, c, W, nb, NB, lo_bits, hi_bits, num_tiles, sort_passes;
  u32 chunk;             // K: sorted entries per accumulation lane
  u32 chunks_per_window; // ceil(n / K)
  // digit stage: nd scalars x Wd digits.  Classic plan: nd == n, Wd == W, base_stride == 0.
  // Window-table plan (bases registered with their multiples 2^(c*j) P): all Wd digit columns feed
  // ONE bucket set, so the stages after the digits see a single window of n = nd*Wd entries (W == 1)
  // and a pair's base field addresses table row j: j*base_stride + base index.
  u32 nd, Wd;
  u64 base_stride;
};
// multiples 2^(c*j) P_i, j < W, stored row-major [j][i] (row 0 = the bases themselves)
struct WindowTable {
  u32 c, W;
  u64 stride;   // number of bases per row
};

struct MsmJobImpl {
  Context *ctx = nullptr;
  int group = BH_G1;
  hipStream_t stream = nullptr;
  hipStream_t hp_stream = nullptr;     // high-priority stream of the merge / reduction phase (null: same stream)
  hipEvent_t hp_event = nullptr;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  hipEvent_t ev_sorted = nullptr, ev_accum = nullptr;   // stage boundaries (profiling)
  MsmPlan plan;
  std::vector<void *> dev_allocs;     // returned to the pool on wait
  void *host_result = nullptr;        // pinned (owned by `res`): W*c XYZZ + ErrFlags
  size_t host_result_bytes = 0;
  JobResources res;
  bool timed = false;                 // stage events recorded (BH_MSM_STAGE_TIMES)
  int early_rc = BH_OK;               // immediate result (n == 0 etc.)
  bool trivial = false;
  // a job answered on the host at issue time (a handful of terms, api.hip): the affine record to hand out
  bool has_result = false;
  alignas(16) unsigned char result[192];
  // inputs needed again by the (rare) error-resolution pass
  const void *scalars_dev = nullptr;
  const u64 *density_dev = nullptr;
  const u32 *word_prefix = nullptr;
  const void *bases_dev = nullptr;
  u32 bases_stride = 0;              // bytes between the records of `bases_dev`
  u64 skip = 0, n_bases = 0;
  int fmt = 0;
  ErrFlags *err_dev = nullptr;
  // completion state (msm.hip): whoever completes the job - its bh_msm_wait, or another issuing thread under
  // back-pressure - holds `mu` while it synchronises the stream and runs the host tail, and leaves the outcome here
  u64 ref_n = 0;                     // MsmOpts::ref_n
  bool always_resolve_ident = false;
  bool saw_eof = false, saw_ident = false, saw_ident_top = false;   // after completion
  std::function<int()> resume;       // the stages after the sort of a job issued with BH_MSM_HOLD (until it is started)
  std::mutex mu;
  bool done = false;
  int done_rc = BH_OK;
  float done_ms[4] = {0, 0, 0, 0};
  u64 done_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // bh_msm_wait_stats
  alignas(16) unsigned char done_out[192];
};


// msm_stages.hip: curve-independent stages (signed digits, radix sort, zero-digit counts)
struct MsmBuffers {
  u64 *pairs_a, *pairs_b;
  u32 *counts, *scan_tmp, *zstart, *word_prefix;   // zstart[w] = #entries with digit 0 in window w
  ErrFlags *err;
};
MsmPlan make_plan(u64 n, unsigned forced_c, unsigned forced_chunk, bool g2);
MsmPlan make_table_plan(u64 n, const WindowTable &t, unsigned forced_chunk, bool g2, int num_cus);
unsigned table_window_bits(u64 n_bases, bool g2);   // the c a window table is built for by default
size_t scan_tmp_elems(u64 n);
// runs stages 1-2 (+ per-window first non-zero position) on `st`; *sorted_out = sorted pairs
int msm_run_stages(const MsmPlan &p, const MsmBuffers &b, const void *scalars_dev, int fmt, const u64 *density_dev,
                   u64 skip, u64 n_bases, hipStream_t st, const u64 **sorted_out);

// msm_g1.hip / msm_g2.hip
int msm_enqueue_g1(MsmJobImpl &job, const void *bases_dev, u64 n_bases, u64 skip, const void *scalars_dev, u64 n,
                   int fmt, const u64 *density_dev, const MsmOpts &opts, const WindowTable *table);
int msm_enqueue_g2(MsmJobImpl &job, const void *bases_dev, u64 n_bases, u64 skip, const void *scalars_dev, u64 n,
                   int fmt, const u64 *density_dev, const MsmOpts &opts, const WindowTable *table);
int msm_finish_g1(MsmJobImpl &job, void *out_affine, float *ms);   // ms: float[4] or null
int msm_finish_g2(MsmJobImpl &job, void *out_affine, float *ms);
int fixed_base_mul_g1(const void *base_host, const void *scalars_dev, u64 n, int fmt, void *out_dev, hipStream_t st, void *table_dev);
int fixed_base_mul_g2(const void *base_host, const void *scalars_dev, u64 n, int fmt, void *out_dev, hipStream_t st, void *table_dev);
constexpr size_t FIXED_BASE_TABLE_RECORDS = 32 * 255;   // msm_ec.cuh FB_ROWS x FB_COLS
// per-point status word of the uncompressed-point loader (api.hip decode kernel + point_check_kernel)
enum PointStatus : u32 {
  PT_COMPRESSED = 1,        // compression flag set on an uncompressed point
  PT_SORT = 2,              // sort flag set on an uncompressed point
  PT_RANGE = 4,             // a coordinate is not < p
  PT_INF_NONZERO = 8,       // infinity flag with non-zero coordinate bits
  PT_IS_INF = 16,           // (valid) identity
  PT_OFF_CURVE = 32,
  PT_NOT_IN_SUBGROUP = 64,
  PT_INVALID_MASK = PT_COMPRESSED | PT_SORT | PT_RANGE | PT_INF_NONZERO | PT_OFF_CURVE | PT_NOT_IN_SUBGROUP,
};
// on-curve + prime-order-subgroup test of decoded points (skips entries already invalid / identity)
// fills rows 1 .. W-1 of a window table whose row 0 holds the n bases
int window_table_g1(void *table_dev, u64 n, u32 c, u32 W, hipStream_t st);
int window_table_g2(void *table_dev, u64 n, u32 c, u32 W, hipStream_t st);
int window_table(int group, void *table_dev, u64 n, u32 c, u32 W, hipStream_t st);
int points_check_g1(const void *pts_dev, u64 n, u32 *status_dev, hipStream_t st);
int points_check_g2(const void *pts_dev, u64 n, u32 *status_dev, hipStream_t st);
int points_check(int group, const void *pts_dev, u64 n, u32 *status_dev, hipStream_t st);
void host_point_add_g1(void *r, const void *a, const void *b, u64 n);
void host_point_add_g2(void *r, const void *a, const void *b, u64 n);
void host_point_mul_g1(void *r, const void *a, const void *k);
void host_point_mul_g2(void *r, const void *a, const void *k);
void host_point_lincomb_g1(void *r, const void *pts, const void *scalars, u64 n);
void host_point_lincomb_g2(void *r, const void *pts, const void *scalars, u64 n);

}  // namespace bh
