// extern "C" surface of libbellman_hip (declared and documented in include/bellman_hip.h).
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <memory>
#include <thread>

#include "../../include/bellman_hip.h"
#include "common.hpp"
#include "msm_types.hpp"
#include "shard_cuts.hpp"

namespace bh {
// fft.hip
int ntt_run(Context &c, fr_t *data, fr_t *scratch, uint32_t log_n, int mode, hipStream_t st);
int fr_mul_assign(Context &c, fr_t *a, const fr_t *b, u64 n, hipStream_t st);
int fr_sub_assign(Context &c, fr_t *a, const fr_t *b, u64 n, hipStream_t st);
int fr_divide_by_z(Context &c, fr_t *a, uint32_t log_n, hipStream_t st);
int fr_distribute_powers(Context &c, fr_t *a, u64 n, const fr_t &g, hipStream_t st);
int h_poly_dev(Context &c, fr_t *a, fr_t *b, fr_t *cc, fr_t *scratch, uint32_t log_n, hipStream_t st);
int fr_gen_powers(Context &c, fr_t *out, u64 n, const fr_t &g, const fr_t &scale, hipStream_t st);
// msm.hip
struct MsmJobImpl;
MsmJobImpl *msm_job_new(Context *ctx, int group);
void msm_job_delete(MsmJobImpl *j);
int msm_job_enqueue(MsmJobImpl &job, const void *bases_dev, u64 n_bases, u64 skip, const void *scalars_dev, u64 n,
                    int fmt, const u64 *density_dev, const MsmOpts &opts, const WindowTable *table);
int msm_job_finish(MsmJobImpl &job, void *out_affine, float *ms, u64 *stats8 = nullptr);
int msm_debug_stages(Context &c, const void *scalars_host, u64 n, int fmt, unsigned cbits, u64 *pairs_out, u32 *zstart_out);
void msm_job_track(MsmJobImpl &job);
int msm_job_start(MsmJobImpl &job);
bool msm_slot_try_reserve(Context &c);
void msm_slot_release(Context &c);
bool msm_complete_oldest(Context &c);
size_t msm_jobs_in_flight(Context &c);
hipStream_t msm_job_stream(MsmJobImpl &job);
int msm_job_after(MsmJobImpl &job, hipStream_t after);
void msm_job_set_result(MsmJobImpl &job, int rc, const void *affine_record);
void msm_job_own(MsmJobImpl &job, void *dev_ptr);
int fixed_base_mul(int group, const void *base_host, const void *scalars_dev, u64 n, int fmt, void *out_dev,
                   hipStream_t st, void *table_dev);
void host_point_add(int group, void *r, const void *a, const void *b, u64 n);
void host_point_mul(int group, void *r, const void *a, const void *k);
void host_point_lincomb(int group, void *r, const void *pts, const void *scalars, u64 n);

// packs strided host records (optionally with an `infinity` flag byte) into dense device records
__global__ void pack_bases_kernel(const unsigned char *raw, size_t stride, long inf_offset, u32 rec_words,
                                  u32 *out, u64 n) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned char *src = raw + i * stride;
  const bool inf = inf_offset >= 0 && src[inf_offset] != 0;
  for (u32 w = 0; w < rec_words; w++) {
    u32 v = 0;
    if (!inf) v = (u32)src[4 * w] | ((u32)src[4 * w + 1] << 8) | ((u32)src[4 * w + 2] << 16) | ((u32)src[4 * w + 3] << 24);
    out[i * rec_words + w] = v;
  }
}

// Zcash uncompressed encoding -> device records: one thread per 48-byte big-endian coordinate.
// coordinate order in: G1 x|y ; G2 x.c1|x.c0|y.c1|y.c0   out: G1 x|y ; G2 x.c0|x.c1|y.c0|y.c1
// `status` (optional, one word per point, zeroed by the caller) receives the PointStatus bits of the
// encoding rules of from_uncompressed_unchecked; `bad_flag` (optional) is the legacy any-compressed flag.
__global__ void decode_uncompressed_kernel(const unsigned char *raw, u32 coords_per_point, fp_t *out, u64 n,
                                           u32 *bad_flag, u32 *status) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * coords_per_point) return;
  const u64 pt = t / coords_per_point;
  const u32 ci = (u32)(t % coords_per_point);
  const unsigned char *p0 = raw + pt * coords_per_point * 48;
  const unsigned char flags = p0[0];
  if ((flags & 0x80) && bad_flag) atomicOr(bad_flag, 1u);   // compressed form is not this entry point's format
  const bool inf = (flags & 0x40) != 0;
  const unsigned char *src = p0 + (size_t)ci * 48;
  fp_t v;
  u32 nonzero = 0;
#pragma unroll
  for (int w = 0; w < 12; w++) {   // limb w (little-endian) = bytes [44-4w, 48-4w) big-endian
    const unsigned char *b = src + 44 - 4 * w;
    u32 x = ((u32)b[0] << 24) | ((u32)b[1] << 16) | ((u32)b[2] << 8) | (u32)b[3];
    if (ci == 0 && w == 11) x &= 0x1fffffffu;   // strip the three flag bits
    nonzero |= x;
    v.l[w] = inf ? 0u : x;
  }
  if (status) {
    u32 st = 0;
    if (ci == 0) {
      if (flags & 0x80) st |= PT_COMPRESSED;
      if (flags & 0x20) st |= PT_SORT;
      if (inf) st |= PT_IS_INF;
    }
    if (inf && nonzero) st |= PT_INF_NONZERO;
    // canonical coordinate: value < p  (Fp::from_bytes)
    u32 borrow = 0;
#pragma unroll
    for (int w = 0; w < 12; w++) (void)subb(v.l[w], FpParams::mod(w), borrow, borrow);
    if (!borrow) st |= PT_RANGE;
    if (st) atomicOr(status + pt, st);
  }
  if (!inf) fe_to_mont(v, v);
  // G2 stores c1 before c0 on the wire
  const u32 co = (coords_per_point == 4) ? (ci ^ 1u) : ci;
  out[pt * coords_per_point + co] = v;
}

// the inverse: Montgomery records -> the uncompressed encoding (Parameters::write, groth16/src/lib.rs:258-287 through
// bls12_381's to_uncompressed): one thread per coordinate; an all-zero record is the identity (flag 0x40, zeros)
__global__ void encode_uncompressed_kernel(const fp_t *pts, u32 coords_per_point, unsigned char *raw, u64 n) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * coords_per_point) return;
  const u64 pt = t / coords_per_point;
  const u32 ci = (u32)(t % coords_per_point);          // position on the wire
  const fp_t *rec = pts + pt * coords_per_point;
  u32 any = 0;
  for (u32 k = 0; k < coords_per_point; k++)
#pragma unroll
    for (int w = 0; w < 12; w++) any |= rec[k].l[w];
  const u32 co = (coords_per_point == 4) ? (ci ^ 1u) : ci;   // G2 stores c1 before c0 on the wire
  fp_t v = rec[co], c;
  fe_from_mont(c, v);
  u32 *dst = reinterpret_cast<u32 *>(raw + pt * coords_per_point * 48 + (size_t)ci * 48);
#pragma unroll
  for (int w = 0; w < 12; w++) {
    u32 x = any ? __builtin_bswap32(c.l[11 - w]) : 0u;
    if (!any && ci == 0 && w == 0) x = 0x40u;   // first byte of the record: the infinity flag
    dst[w] = x;
  }
}

// smallest index whose status makes Parameters::read fail (lib.rs:300-315)
__global__ void first_bad_point_kernel(const u32 *status, u64 n, u32 forbid_identity, unsigned long long *min_idx) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    const u32 st = status[i];
    if ((st & PT_INVALID_MASK) || (forbid_identity && (st & PT_IS_INF))) atomicMin(min_idx, (unsigned long long)i);
  }
}
}  // namespace bh

using namespace bh;

constexpr size_t HOST_PREFIX_POINTS = 64;   // leading records mirrored on the host for the tiny-multiexp path
constexpr size_t TINY_MSM_MAX = 8;          // multiexps of at most this many terms are answered on the host

struct bh_bases {
  int group;
  void *dev;
  size_t n;
  bool owned;
  // the first min(n, HOST_PREFIX_POINTS) records on the host: create_proof's `inputs` multiexps have one or
  // two terms (prover.rs:275-280,296-300,312-316) - a kernel pipeline for them is all launch latency
  std::vector<unsigned char> host_prefix = {};
  // optional window table (bh_bases_precompute): [W][n] affine records, row 0 = a copy of the bases
  void *table = nullptr;
  WindowTable tab = {0, 0, 0};
  bool auto_table = false;   // built at registration under the context's table budget; bh_ctx_trim may drop it
  bh_ctx *ctx = nullptr;
  // [r4] G1 vectors that run the classic plan (>= 2^17 points, owned by the handle): a second copy of the records at a
  // 128-byte stride, which is what the bucket accumulation gathers from - every gather is exactly one cache line (a
  // 96-byte record at a 96-byte stride straddles two lines half of the time).  profiles/archive/r4_call11_padded_bases.txt:
  // FETCH_SIZE of the accumulate launch 2.20 -> 1.58 GB at 2^20; accumulate -1 % at 2^20 (the table sits in the
  // Infinity Cache either way), -3 % at 2^22 (it does not).  Dense `dev` stays what every other path and the API see.
  // BELLMAN_HIP_BASE_PAD=0 switches it off; vectors whose copy would exceed 1/16 of the device memory are not padded.
  void *padded = nullptr;
  // [r4] ... and the window table of a G1 vector too large for the Infinity Cache (>= 2^19 points: 16 rows of 2^19
  // records are 0.8 GB) is KEPT at that stride only: `table` then holds W * n records of 128 bytes, read by the bucket
  // accumulation alone (MsmOpts::padded_table); everything else of such a job reads the dense `dev`.
  bool table_padded = false;
};
static inline size_t table_rec_bytes(const bh_bases *b) { return b->group == BH_G1 ? (b->table_padded ? 128 : 96) : 192; }
// a scalar vector resident in HBM (bh_scalars_*): create_proof hands the same assignment to up to four multiexps
struct bh_scalars {
  bh_ctx *ctx;
  void *dev;
  size_t n;
  int fmt;
  bool pooled;   // dev came from the context's pool (else: adopted from the caller, not freed)
};
struct bh_msm_job {
  MsmJobImpl *impl;
};

// hipMalloc'ed block freed on scope exit unless released: the BH_HIP_CHECK early returns of the register /
// read paths must not leak device memory
struct DevGuard {
  void *p = nullptr;
  DevGuard() = default;
  DevGuard(const DevGuard &) = delete;
  DevGuard &operator=(const DevGuard &) = delete;
  ~DevGuard() { if (p) (void)hipFree(p); }
  void *release() { void *q = p; p = nullptr; return q; }
};

// registers the finished device vector: mirrors its leading records on the host (ordered after whatever filled
// the vector on the context stream)
static int finish_bases(bh_ctx *ctx, bh_bases *b) {
  const size_t rec = b->group == BH_G1 ? 96 : 192;
  const size_t k = b->n < HOST_PREFIX_POINTS ? b->n : HOST_PREFIX_POINTS;
  b->host_prefix.resize(k * rec);
  if (k) {
    BH_HIP_CHECK(hipMemcpyAsync(b->host_prefix.data(), b->dev, k * rec, hipMemcpyDeviceToHost, ctx->c.stream));
    BH_HIP_CHECK(hipStreamSynchronize(ctx->c.stream));
  }
  return BH_OK;
}
// Base vectors of up to 2^18 (G1) / 2^22 (G2) points get their window table at registration (BELLMAN_HIP_TABLE_MAX_LOG2
// overrides both limits, BELLMAN_HIP_TABLE_MAX_LOG2_G1 the G1 limit alone; 0 = never): a multiexp over a few thousand
// terms is a chain of latency-bound steps, and with the table the chain loses the 255-step doubling ladder over the
// windows and all but one of its bucket reductions (the CRS is registered once per circuit).
// G1 history: 2^16 until round 4; 2^18 after profiles/archive/r4_call13_fft_batched_loads_and_plan_sweeps.txt (2^17 1.04 vs
// 1.31 ms, 2^18 1.47 vs 1.72).  2^19 ... 2^22 were measured and NOT adopted (profiles/archive/r4_call16_g1_tables_2p19_2p22.txt,
// r4_call17_g1_tables_in_proofs.txt): a multiexp called alone gets faster - the classic plan ends in a HOST tail of 256
// doublings + 256 additions (16 windows x 16 bit sums, 0.27-0.44 ms by box), the table plan in 19 + 20: wall 2.21 vs
// 2.67 ms at 2^19, 3.92-3.95 vs 4.09-4.28 at 2^20, 6.95 vs 7.42 at 2^21, 12.3 vs 13.5 at 2^22 - but its bucket
// accumulation is slower per addition (2.89-2.94 vs 2.54-2.57 ms at 2^20: the same 16 n gathers come from a 2 GB table
// in HBM instead of a 128 MB vector in the Infinity Cache), and wherever the host tail is hidden behind the next job
// that is all that counts: two multiexps in flight 265 vs 282 M terms/s, a 2^20 proof 77.8 vs 74.9 ms (GPU part 23.6 vs
// 20.7), twelve proof threads 37-40 vs 43-45 proofs/s.  A caller whose multiexps run one at a time can opt in with
// BELLMAN_HIP_TABLE_MAX_LOG2_G1=22 (or bh_bases_precompute).
static unsigned auto_table_max_log2(int group) {
  static const int v = [] {
    const char *e = getenv("BELLMAN_HIP_TABLE_MAX_LOG2");
    if (!e || !*e) return -1;
    const long x = strtol(e, nullptr, 10);
    return (int)(x < 0 ? 0 : x > 24 ? 24 : x);
  }();
  static const int v1 = [] {
    const char *e = getenv("BELLMAN_HIP_TABLE_MAX_LOG2_G1");
    if (!e || !*e) return -1;
    const long x = strtol(e, nullptr, 10);
    return (int)(x < 0 ? 0 : x > 24 ? 24 : x);
  }();
  if (group == BH_G1 && v1 >= 0) return (unsigned)v1;
  // [r6] G1: up to 2^24 points (it was 2^18): a 13-row table at a 128-byte stride is 1.7 GB per 2^20 points, built in 0.13 s;
  // 2^23 24.6 -> 21.4 ms, 2^24 44.6 -> 40.5 (profiles/r6_call44_g1_tables_2p23_2p24.txt).  Always within the context's table
  // budget (a quarter of the memory by default): two of the 28 GB tables of 2^24-point queries fit, the next query stays classic
  return v >= 0 ? (unsigned)v : (group == BH_G1 ? 24u : 22u);
}
// G1 tables of 2^19 points and more are stored at a 128-byte record stride (bh_bases::table_padded);
// BELLMAN_HIP_TABLE_PAD=0 keeps them dense
static bool table_will_pad(const bh_bases *b) {
  static const bool on = [] { const char *e = getenv("BELLMAN_HIP_TABLE_PAD"); return !(e && *e == '0'); }();
  return on && b->group == BH_G1 && b->n >= ((size_t)1 << 19);
}
static size_t table_bytes_for(const bh_bases *b, unsigned c) {
  const u32 W = (256 + c - 1) / c;
  return (size_t)W * b->n * (b->group == BH_G1 ? (table_will_pad(b) ? 128 : 96) : 192);
}
static int new_bases(bh_ctx *ctx, int group, void *dev, size_t n, bool owned, bh_bases **out) {
  bh_bases *b = new bh_bases{group, dev, n, owned};
  b->ctx = ctx;
  // A wrapped (non-owned) buffer is a LIVE view: the caller may fill or rewrite it after wrapping, on any stream.
  // Nothing is snapshotted from it - no host mirror of the leading records (tiny multiexps take the kernel
  // pipeline), no automatic window table - so every multiexp reads the buffer as it is when the job runs.
  // (bh_bases_precompute on such a handle is an explicit snapshot, documented in the header.)
  if (!owned) { *out = b; return BH_OK; }
  int rc = finish_bases(ctx, b);
  if (rc != BH_OK) {
    if (dev) (void)hipFree(dev);
    delete b;
    return rc;
  }
  // automatic window table: only while all automatic tables of the context stay within its budget (default a
  // quarter of the device's memory, BELLMAN_HIP_TABLE_BUDGET_MB / bh_ctx_set_limits): a table is 13-32 x its base
  // vector (2^22 G2 points: 12.9 GB) and must not starve the per-proof workspaces
  static const bool pad_on = [] { const char *e = getenv("BELLMAN_HIP_BASE_PAD"); return !(e && *e == '0'); }();
  const unsigned lg_table = auto_table_max_log2(group);
  const bool table_size = lg_table && n > TINY_MSM_MAX && n <= (size_t(1) << lg_table);   // (takes a window table instead)
  if (table_size) {
    const size_t need = table_bytes_for(b, table_window_bits(n, group == BH_G2));
    bool fits;
    {
      std::lock_guard<std::mutex> g(ctx->c.job_mu);
      fits = ctx->c.table_bytes + need <= ctx->c.table_budget;
      if (fits) ctx->c.table_bytes += need;   // reserved
    }
    if (fits) {
      if (bh_bases_precompute(ctx, b, 0) == BH_OK) {   // best effort
        b->auto_table = true;
        std::lock_guard<std::mutex> g(ctx->c.job_mu);
        ctx->c.tables.push_back(b);
        const size_t actual = (size_t)b->tab.W * b->n * table_rec_bytes(b);   // (dense after all, if the padded copy did not fit)
        if (actual < need) ctx->c.table_bytes -= need - actual;
      } else {
        std::lock_guard<std::mutex> g(ctx->c.job_mu);
        ctx->c.table_bytes -= need;
      }
    }
  }
  // (a vector that got its window table gathers from the table; one that did not - too large, or over the table budget - gets the
  // 128-byte-stride copy of its points for the classic plan's gathers)
  if (pad_on && group == BH_G1 && !b->table && n >= ((size_t)1 << 17) && n < ((size_t)1 << 31) &&
      (ctx->c.hbm_total == 0 || n * 128 <= ctx->c.hbm_total / 16)) {
    if (hipMalloc(&b->padded, n * 128) == hipSuccess) {
      if (hipMemcpy2DAsync(b->padded, 128, dev, 96, 96, n, hipMemcpyDeviceToDevice, ctx->c.stream) != hipSuccess ||
          hipStreamSynchronize(ctx->c.stream) != hipSuccess) {
        (void)hipFree(b->padded);
        b->padded = nullptr;
      }
    } else {
      (void)hipGetLastError();
      b->padded = nullptr;
    }
  }
  *out = b;
  return BH_OK;
}

static inline hipStream_t pick_stream(bh_ctx *ctx, void *stream) { return stream ? (hipStream_t)stream : ctx->c.stream; }

extern "C" {

const char *bh_version(void) { return "bellman_hip 0.4 (gfx950, built " __DATE__ " " __TIME__ ")"; }

// A proof keeps 6-7 job streams in flight (one per multiexp + the h block); the HIP runtime multiplexes streams onto
// GPU_MAX_HW_QUEUES hardware queues, 4 by default, and jobs that share a queue run one after the other (MiMC-322
// proof 2.99 ms with 4 queues, 2.23 ms with 16; 12 concurrent 2^20 proofs 33.1 -> 36.2 /s:
// profiles/archive/r2_call12_hw_queues.txt).  The runtime reads the variable when it initialises, so the host program calls
// bh_runtime_configure() before its first HIP call (the Python loader and bench.py set the variable themselves); the
// library no longer touches the environment behind the caller's back when it is loaded.
static std::atomic<bool> g_hip_touched{false};      // this library has made a HIP call
static std::atomic<bool> g_configured_early{false};
int bh_runtime_configure(void) {
  const bool early = !g_hip_touched.load();
  const char *cur = getenv("GPU_MAX_HW_QUEUES");
  if (!cur || !*cur) setenv("GPU_MAX_HW_QUEUES", "16", 0);
  if (early) g_configured_early.store(true);
  return early ? 1 : 0;
}

int bh_ctx_create(int device, bh_ctx **out) {
  int count = 0;
  const char *q = getenv("GPU_MAX_HW_QUEUES");
  const int queues_env = (q && *q) ? atoi(q) : 0;
  const bool early = g_configured_early.load() || (!g_hip_touched.load() && queues_env > 0);
  g_hip_touched.store(true);
  if (hipGetDeviceCount(&count) != hipSuccess || count <= device || device < 0) return BH_ERR_NO_DEVICE;
  BH_HIP_CHECK(hipSetDevice(device));
  bh_ctx *ctx = new bh_ctx();
  ctx->c.device = device;
  ctx->c.hw_queues_env = queues_env;
  ctx->c.configured_early = early;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
    ctx->c.num_cus = prop.multiProcessorCount;
    ctx->c.hbm_total = prop.totalGlobalMem;
  }
  // jobs in flight per context: a 2^20-term multiexp holds ~0.6 GB of workspace - by default as many as fit a
  // quarter of the device's memory at that size, at least 8 (create_proof issues 8, prover.rs:244-318)
  {
    const char *e = getenv("BELLMAN_HIP_MAX_JOBS");
    long v = e && *e ? strtol(e, nullptr, 10) : 0;
    if (v <= 0) v = (long)(ctx->c.hbm_total / 4 / (size_t(640) << 20));
    ctx->c.max_jobs = (uint32_t)(v < 8 ? 8 : v > 4096 ? 4096 : v);
  }
  {
    const char *e = getenv("BELLMAN_HIP_TABLE_BUDGET_MB");
    ctx->c.table_budget = (e && *e) ? (size_t)strtoull(e, nullptr, 10) << 20 : ctx->c.hbm_total / 4;
    const char *ef = getenv("BELLMAN_HIP_FFT_TABLE_BUDGET_MB");
    ctx->c.fft_table_budget = (ef && *ef) ? (size_t)strtoull(ef, nullptr, 10) << 20 : ctx->c.hbm_total / 8;
  }
  {
    const char *e = getenv("BELLMAN_HIP_POOL_CAP_MB");
    if (e && *e) ctx->c.pool.set_cap((size_t)strtoull(e, nullptr, 10) << 20);
  }
  Context *cp = &ctx->c;
  ctx->c.pool.set_pressure_handler([cp] { return msm_complete_oldest(*cp); });
  BH_HIP_CHECK(hipStreamCreateWithFlags(&ctx->c.stream, hipStreamNonBlocking));
  *out = ctx;
  return BH_OK;
}
int bh_ctx_set_limits(bh_ctx *ctx, uint32_t max_jobs_in_flight, size_t pool_cap_bytes, size_t table_budget_bytes,
                      size_t fft_table_budget_bytes) {
  if (!ctx) return BH_ERR_INVALID_ARG;
  if (max_jobs_in_flight) ctx->c.max_jobs = max_jobs_in_flight;
  if (pool_cap_bytes != (size_t)-1) ctx->c.pool.set_cap(pool_cap_bytes);
  if (table_budget_bytes != (size_t)-1) { std::lock_guard<std::mutex> g(ctx->c.job_mu); ctx->c.table_budget = table_budget_bytes; }
  if (fft_table_budget_bytes != (size_t)-1) { std::lock_guard<std::mutex> g(ctx->c.fft_mu); ctx->c.fft_table_budget = fft_table_budget_bytes; }
  return BH_OK;
}
int bh_ctx_info(bh_ctx *ctx, bh_ctx_info_t *info) {
  if (!ctx || !info) return BH_ERR_INVALID_ARG;
  memset(info, 0, sizeof *info);
  info->device = ctx->c.device;
  info->num_cus = (uint32_t)ctx->c.num_cus;
  info->hbm_bytes = ctx->c.hbm_total;
  info->hw_queues_requested = (uint32_t)ctx->c.hw_queues_env;
  info->hw_queues_set_before_hip_init = ctx->c.configured_early ? 1u : 0u;
  info->max_jobs_in_flight = ctx->c.max_jobs;
  info->jobs_in_flight = (uint32_t)msm_jobs_in_flight(ctx->c);
  info->pool_bytes_held = ctx->c.pool.bytes_held();
  info->pool_bytes_idle = ctx->c.pool.bytes_idle();
  {
    std::lock_guard<std::mutex> g(ctx->c.job_mu);
    info->table_bytes = ctx->c.table_bytes;
    info->table_budget = ctx->c.table_budget;
  }
  {
    std::lock_guard<std::mutex> g(ctx->c.fft_mu);
    info->fft_table_bytes = ctx->c.fft_table_bytes;
    info->fft_table_budget = ctx->c.fft_table_budget;
  }
  return BH_OK;
}
int bh_ctx_trim(bh_ctx *ctx) {
  // Workspaces of finished jobs are kept in a size-bucketed cache (a 2^26-term multiexp leaves ~30 GB
  // there): hand the idle blocks and the cached FFT tables back to the driver.  Blocks in use stay.
  if (!ctx) return BH_ERR_INVALID_ARG;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  BH_HIP_CHECK(hipDeviceSynchronize());
  ctx->c.pool.release_all();
  {
    // window tables built automatically at registration are a cache too (the handles stay usable without them;
    // bh_bases_precompute rebuilds one on request) - unless a multiexp is still in flight: a job issued with BH_MSM_HOLD
    // and not yet started has captured its table pointer but enqueued nothing the device synchronise above could have
    // waited for, so the tables stay until a trim finds the context idle
    std::lock_guard<std::mutex> g(ctx->c.job_mu);
    if (ctx->c.inflight.empty() && ctx->c.issuing == 0)
    for (bh_bases *b : ctx->c.tables) {
      if (b->table) (void)hipFree(b->table);
      b->table = nullptr;
      b->table_padded = false;
      b->auto_table = false;
    }
    if (ctx->c.inflight.empty() && ctx->c.issuing == 0) {
      ctx->c.tables.clear();
      ctx->c.table_bytes = 0;
    }
  }
  {
    std::unique_lock<std::shared_mutex> ex(ctx->c.fft_use_mu);   // no transform between its table lookup and its launches
    BH_HIP_CHECK(hipDeviceSynchronize());
    std::lock_guard<std::mutex> g(ctx->c.fft_mu);
    for (auto &kv : ctx->c.fft_tables) fft_tables_free(kv.second);
    ctx->c.fft_tables.clear();
    ctx->c.fft_table_bytes = 0;
    fft_master_free(ctx->c);
  }
  return BH_OK;
}
void bh_ctx_destroy(bh_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->c.device);
  (void)hipDeviceSynchronize();
  for (auto &kv : ctx->c.fft_tables) fft_tables_free(kv.second);
  fft_master_free(ctx->c);
  ctx->c.pool.release_all();
  {
    // base handles that outlive the context (a host that releases them later - Python objects collected after the
    // Worker was closed) keep their memory, tables included, and must not reach back into the context's bookkeeping
    std::lock_guard<std::mutex> g(ctx->c.job_mu);
    for (bh_bases *b : ctx->c.tables) { b->auto_table = false; b->ctx = nullptr; }
    ctx->c.tables.clear();
    ctx->c.table_bytes = 0;
  }
  for (auto &r : ctx->c.job_pool) {
    for (int i = 0; i < 4; i++) if (r.ev[i]) (void)hipEventDestroy(r.ev[i]);
    if (r.dep_event) (void)hipEventDestroy(r.dep_event);
    if (r.acc_event) (void)hipEventDestroy(r.acc_event);
    if (r.sort_event) (void)hipEventDestroy(r.sort_event);
    if (r.pinned) (void)hipHostFree(r.pinned);
    if (r.hp_event) (void)hipEventDestroy(r.hp_event);
    if (r.hp_stream) (void)hipStreamDestroy(r.hp_stream);
    if (r.stream) (void)hipStreamDestroy(r.stream);
  }
  ctx->c.job_pool.clear();
  for (hipStream_t st : ctx->c.stream_pool) (void)hipStreamDestroy(st);
  ctx->c.stream_pool.clear();
  for (hipEvent_t ev : ctx->c.barrier_events) (void)hipEventDestroy(ev);
  ctx->c.barrier_events.clear();
  for (hipStream_t st : ctx->c.hp_streams) (void)hipStreamDestroy(st);
  ctx->c.hp_streams.clear();
  ctx->c.hp_stream_pool.clear();
  if (ctx->c.stream) (void)hipStreamDestroy(ctx->c.stream);
  delete ctx;
}
uint32_t bh_ctx_log_num_cus(const bh_ctx *ctx) {
  uint32_t p = 0;
  while ((1u << (p + 1)) <= (uint32_t)ctx->c.num_cus) p++;
  return p;
}
// served from the context's size-bucketed pool: hipMalloc / hipFree synchronise the device and a
// prover allocates the same handful of vectors for every proof
int bh_dev_alloc(bh_ctx *ctx, size_t bytes, void **dev_ptr) {
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  *dev_ptr = ctx->c.pool.acquire(bytes);
  return *dev_ptr ? BH_OK : BH_ERR_HIP;
}
int bh_dev_free(bh_ctx *ctx, void *dev_ptr) {
  // callers free after synchronising on the work that used the buffer (bh_msm_wait / bh_ctx_synchronize)
  ctx->c.pool.release(dev_ptr);
  return BH_OK;
}
int bh_dev_upload(bh_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes) {
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));   // entry points may be called from any host thread
  BH_HIP_CHECK(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->c.stream));
  BH_HIP_CHECK(hipStreamSynchronize(ctx->c.stream));
  return BH_OK;
}
int bh_dev_download(bh_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes) {
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  BH_HIP_CHECK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->c.stream));
  BH_HIP_CHECK(hipStreamSynchronize(ctx->c.stream));
  return BH_OK;
}
int bh_dev_zero(bh_ctx *ctx, void *dev_ptr, size_t bytes) {
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  if (bytes) BH_HIP_CHECK(hipMemsetAsync(dev_ptr, 0, bytes, ctx->c.stream));
  return BH_OK;
}
int bh_stream_create(bh_ctx *ctx, void **stream) {
  // recycled through the context: hipStreamCreate / Destroy cost about a millisecond together, which was a third of
  // a MiMC-sized proof (one stream per proof, groth16_prover.cpp)
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  {
    std::lock_guard<std::mutex> g(ctx->c.job_mu);
    if (!ctx->c.stream_pool.empty()) {
      *stream = (void *)ctx->c.stream_pool.back();
      ctx->c.stream_pool.pop_back();
      return BH_OK;
    }
  }
  hipStream_t st = nullptr;
  BH_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *stream = (void *)st;
  return BH_OK;
}
int bh_stream_create_priority(bh_ctx *ctx, int high, void **stream) {
  if (!high) return bh_stream_create(ctx, stream);
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  {
    std::lock_guard<std::mutex> g(ctx->c.job_mu);
    if (!ctx->c.hp_stream_pool.empty()) {
      *stream = (void *)ctx->c.hp_stream_pool.back();
      ctx->c.hp_stream_pool.pop_back();
      return BH_OK;
    }
  }
  int lo = 0, hi = 0;   // numerically lower = higher priority
  BH_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t st = nullptr;
  BH_HIP_CHECK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi));
  std::lock_guard<std::mutex> g(ctx->c.job_mu);
  ctx->c.hp_streams.push_back(st);
  *stream = (void *)st;
  return BH_OK;
}
int bh_stream_destroy(bh_ctx *ctx, void *stream) {
  // the caller has synchronised the stream; it goes back to the pool (destroyed with the context)
  if (!stream) return BH_OK;
  std::lock_guard<std::mutex> g(ctx->c.job_mu);
  for (hipStream_t h : ctx->c.hp_streams)
    if (h == (hipStream_t)stream) { ctx->c.hp_stream_pool.push_back(h); return BH_OK; }
  ctx->c.stream_pool.push_back((hipStream_t)stream);
  return BH_OK;
}
int bh_stream_synchronize(bh_ctx *ctx, void *stream) {
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  BH_HIP_CHECK(hipStreamSynchronize(pick_stream(ctx, stream)));
  return BH_OK;
}
int bh_dev_upload_on(bh_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes, void *stream) {
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  if (bytes) BH_HIP_CHECK(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, pick_stream(ctx, stream)));
  return BH_OK;
}
int bh_dev_zero_on(bh_ctx *ctx, void *dev_ptr, size_t bytes, void *stream) {
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  if (bytes) BH_HIP_CHECK(hipMemsetAsync(dev_ptr, 0, bytes, pick_stream(ctx, stream)));
  return BH_OK;
}
int bh_ctx_accumulations_after(bh_ctx *ctx, void *stream) {
  // the next chip-filling bucket accumulation issued on this context - and through the accumulation chain
  // (common.hpp) every later one - starts after everything enqueued on `stream` so far
  if (!ctx || !stream) return BH_ERR_INVALID_ARG;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  std::lock_guard<std::mutex> g(ctx->c.acc_mu);
  constexpr size_t RING = 64;   // an event is only referenced until the next accumulation is issued
  if (ctx->c.barrier_events.size() < RING) {
    hipEvent_t ev = nullptr;
    BH_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    ctx->c.barrier_events.push_back(ev);
    ctx->c.barrier_next = ctx->c.barrier_events.size() - 1;
  } else {
    ctx->c.barrier_next = (ctx->c.barrier_next + 1) % RING;
  }
  hipEvent_t ev = ctx->c.barrier_events[ctx->c.barrier_next];
  BH_HIP_CHECK(hipEventRecord(ev, (hipStream_t)stream));
  if (ctx->c.pending_barriers.size() < RING / 2) ctx->c.pending_barriers.push_back(ev);
  return BH_OK;
}
int bh_ctx_synchronize(bh_ctx *ctx) {
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  BH_HIP_CHECK(hipDeviceSynchronize());
  return BH_OK;
}

// ---- EvaluationDomain ------------------------------------------------------------------------
int bh_fft_fr_dev(bh_ctx *ctx, void *data_dev, uint32_t log_n, int mode, void *stream) {
  if (log_n >= 32) return BH_ERR_DEGREE_TOO_LARGE;
  if (mode < 0 || mode > 3) return BH_ERR_INVALID_ARG;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  hipStream_t st = pick_stream(ctx, stream);
  void *scratch = nullptr;
  if (log_n > 11) {   // more than one pass (fft.hip NTT_MAX_R)
    scratch = ctx->c.pool.acquire(sizeof(fr_t) << log_n);
    if (!scratch) return BH_ERR_HIP;
  }
  int rc = ntt_run(ctx->c, (fr_t *)data_dev, (fr_t *)scratch, log_n, mode, st);
  if (scratch) {
    // the scratch buffer may be recycled by another stream: fence before returning it
    if (hipStreamSynchronize(st) != hipSuccess && rc == BH_OK) rc = BH_ERR_HIP;
    ctx->c.pool.release(scratch);
  }
  return rc;
}
int bh_fft_fr(bh_ctx *ctx, void *data_host, uint32_t log_n, int mode) {
  if (log_n >= 32) return BH_ERR_DEGREE_TOO_LARGE;
  const size_t bytes = sizeof(fr_t) << log_n;
  void *d = ctx->c.pool.acquire(bytes);
  if (!d) return BH_ERR_HIP;
  int rc = bh_dev_upload(ctx, d, data_host, bytes);
  if (rc == BH_OK) rc = bh_fft_fr_dev(ctx, d, log_n, mode, nullptr);
  if (rc == BH_OK) rc = bh_dev_download(ctx, data_host, d, bytes);
  ctx->c.pool.release(d);
  return rc;
}
int bh_fr_mul_assign_dev(bh_ctx *ctx, void *a, const void *b, size_t n, void *stream) {
  return fr_mul_assign(ctx->c, (fr_t *)a, (const fr_t *)b, n, pick_stream(ctx, stream));
}
int bh_fr_sub_assign_dev(bh_ctx *ctx, void *a, const void *b, size_t n, void *stream) {
  return fr_sub_assign(ctx->c, (fr_t *)a, (const fr_t *)b, n, pick_stream(ctx, stream));
}
int bh_fr_divide_by_z_on_coset_dev(bh_ctx *ctx, void *a, uint32_t log_n, void *stream) {
  if (log_n >= 32) return BH_ERR_DEGREE_TOO_LARGE;
  return fr_divide_by_z(ctx->c, (fr_t *)a, log_n, pick_stream(ctx, stream));
}
int bh_fr_distribute_powers_dev(bh_ctx *ctx, void *a, size_t n, const void *g_host, void *stream) {
  fr_t g;
  memcpy(&g, g_host, sizeof g);
  return fr_distribute_powers(ctx->c, (fr_t *)a, n, g, pick_stream(ctx, stream));
}
int bh_fr_powers_dev(bh_ctx *ctx, void *out, size_t n, const void *g_host, const void *scale_host, void *stream) {
  fr_t g, sc;
  memcpy(&g, g_host, sizeof g);
  memcpy(&sc, scale_host, sizeof sc);
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  return fr_gen_powers(ctx->c, (fr_t *)out, n, g, sc, pick_stream(ctx, stream));
}
int bh_h_poly_fr_dev_on(bh_ctx *ctx, void *a, void *b, void *c, void *scratch, uint32_t log_n, void *stream) {
  // enqueue only: the caller owns `scratch` (2^log_n Fr; may be null up to 2^11) until the stream has drained
  if (log_n >= 32) return BH_ERR_DEGREE_TOO_LARGE;
  if (!ctx || !a || !b || !c || (log_n > 11 && !scratch)) return BH_ERR_INVALID_ARG;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  return h_poly_dev(ctx->c, (fr_t *)a, (fr_t *)b, (fr_t *)c, (fr_t *)scratch, log_n, pick_stream(ctx, stream));
}
int bh_h_poly_fr_dev(bh_ctx *ctx, void *a, void *b, void *c, uint32_t log_n, void *stream) {
  if (log_n >= 32) return BH_ERR_DEGREE_TOO_LARGE;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  hipStream_t st = pick_stream(ctx, stream);
  void *scratch = ctx->c.pool.acquire(sizeof(fr_t) << log_n);
  if (!scratch) return BH_ERR_HIP;
  int rc = h_poly_dev(ctx->c, (fr_t *)a, (fr_t *)b, (fr_t *)c, (fr_t *)scratch, log_n, st);
  if (hipStreamSynchronize(st) != hipSuccess && rc == BH_OK) rc = BH_ERR_HIP;
  ctx->c.pool.release(scratch);
  return rc;
}
int bh_h_poly_fr(bh_ctx *ctx, const void *a_host, const void *b_host, const void *c_host, size_t n_evals,
                 void *h_out_host, size_t *h_len) {
  // EvaluationDomain::from_coeffs (domain.rs:47-79): m = next power of two >= len, zero padded
  uint32_t log_n = 0;
  size_t m = 1;
  while (m < n_evals) {
    m *= 2;
    log_n++;
    if (log_n >= 32) return BH_ERR_DEGREE_TOO_LARGE;
  }
  const size_t bytes = sizeof(fr_t) * m, in_bytes = sizeof(fr_t) * n_evals;
  void *d[3] = {ctx->c.pool.acquire(bytes), ctx->c.pool.acquire(bytes), ctx->c.pool.acquire(bytes)};
  const void *h[3] = {a_host, b_host, c_host};
  int rc = (d[0] && d[1] && d[2]) ? BH_OK : BH_ERR_HIP;
  for (int i = 0; i < 3 && rc == BH_OK; i++) {
    if (hipMemsetAsync(d[i], 0, bytes, ctx->c.stream) != hipSuccess) rc = BH_ERR_HIP;
    if (rc == BH_OK && in_bytes &&
        hipMemcpyAsync(d[i], h[i], in_bytes, hipMemcpyHostToDevice, ctx->c.stream) != hipSuccess)
      rc = BH_ERR_HIP;
  }
  if (rc == BH_OK) rc = bh_h_poly_fr_dev(ctx, d[0], d[1], d[2], log_n, nullptr);
  if (rc == BH_OK) rc = bh_dev_download(ctx, h_out_host, d[0], sizeof(fr_t) * (m - 1));   // prover.rs:238-239
  if (h_len) *h_len = m - 1;
  for (int i = 0; i < 3; i++) ctx->c.pool.release(d[i]);
  return rc;
}

int bh_h_poly_fr_scalars(bh_ctx *ctx, const void *a_host, const void *b_host, const void *c_host, size_t n_evals,
                         bh_scalars **h_out) {
  if (!ctx || !h_out || (n_evals && (!a_host || !b_host || !c_host))) return BH_ERR_INVALID_ARG;
  uint32_t log_n = 0;
  size_t m = 1;
  while (m < n_evals) {   // EvaluationDomain::from_coeffs (domain.rs:47-79)
    m *= 2;
    log_n++;
    if (log_n >= 32) return BH_ERR_DEGREE_TOO_LARGE;
  }
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  const size_t bytes = sizeof(fr_t) * m, in_bytes = sizeof(fr_t) * n_evals;
  // the three vectors are staged on three streams so that the copies of b and c overlap the first transforms... the
  // h block itself is a dependent chain, so: one caller-private stream (concurrent proofs do not serialise on the
  // context stream)
  void *stv = nullptr;
  int rc = bh_stream_create(ctx, &stv);
  if (rc != BH_OK) return rc;
  hipStream_t st = (hipStream_t)stv;
  void *d[3] = {ctx->c.pool.acquire(bytes), ctx->c.pool.acquire(bytes), ctx->c.pool.acquire(bytes)};
  void *scratch = log_n > 11 ? ctx->c.pool.acquire(bytes) : nullptr;
  const void *h[3] = {a_host, b_host, c_host};
  rc = (d[0] && d[1] && d[2] && (log_n <= 11 || scratch)) ? BH_OK : BH_ERR_HIP;
  for (int i = 0; i < 3 && rc == BH_OK; i++) {
    if (m > n_evals && hipMemsetAsync((char *)d[i] + in_bytes, 0, bytes - in_bytes, st) != hipSuccess) rc = BH_ERR_HIP;
    if (rc == BH_OK && in_bytes && hipMemcpyAsync(d[i], h[i], in_bytes, hipMemcpyHostToDevice, st) != hipSuccess)
      rc = BH_ERR_HIP;
  }
  if (rc == BH_OK) rc = h_poly_dev(ctx->c, (fr_t *)d[0], (fr_t *)d[1], (fr_t *)d[2], (fr_t *)scratch, log_n, st);
  if (hipStreamSynchronize(st) != hipSuccess && rc == BH_OK) rc = BH_ERR_HIP;
  (void)bh_stream_destroy(ctx, stv);
  ctx->c.pool.release(d[1]);
  ctx->c.pool.release(d[2]);
  ctx->c.pool.release(scratch);
  if (rc != BH_OK) { ctx->c.pool.release(d[0]); return rc; }
  *h_out = new bh_scalars{ctx, d[0], m - 1, BH_SCALARS_MONT, true};   // a.len() - 1 coefficients, prover.rs:238-239
  return BH_OK;
}

// ---- bases -------------------------------------------------------------------------------------
int bh_bases_register(bh_ctx *ctx, int group, const void *host_points, size_t n, size_t stride, long inf_offset,
                      bh_bases **out) {
  if (group != BH_G1 && group != BH_G2) return BH_ERR_INVALID_ARG;
  const size_t rec = group == BH_G1 ? 96 : 192;
  if (stride < rec) return BH_ERR_INVALID_ARG;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  DevGuard dev;
  BH_HIP_CHECK(hipMalloc(&dev.p, n ? n * rec : 16));
  if (n) {
    if (stride == rec && inf_offset < 0) {
      BH_HIP_CHECK(hipMemcpyAsync(dev.p, host_points, n * rec, hipMemcpyHostToDevice, ctx->c.stream));
    } else {
      DevGuard raw;
      BH_HIP_CHECK(hipMalloc(&raw.p, n * stride));
      BH_HIP_CHECK(hipMemcpyAsync(raw.p, host_points, n * stride, hipMemcpyHostToDevice, ctx->c.stream));
      hipLaunchKernelGGL(pack_bases_kernel, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->c.stream,
                         (const unsigned char *)raw.p, stride, inf_offset, (u32)(rec / 4), (u32 *)dev.p, (u64)n);
      BH_HIP_CHECK(hipGetLastError());
      BH_HIP_CHECK(hipStreamSynchronize(ctx->c.stream));
    }
    BH_HIP_CHECK(hipStreamSynchronize(ctx->c.stream));
  }
  return new_bases(ctx, group, dev.release(), n, true, out);
}
int bh_bases_register_uncompressed(bh_ctx *ctx, int group, const void *host_bytes, size_t n, bh_bases **out) {
  if (group != BH_G1 && group != BH_G2) return BH_ERR_INVALID_ARG;
  const size_t rec = group == BH_G1 ? 96 : 192;
  const u32 cpp = group == BH_G1 ? 2 : 4;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  DevGuard dev, raw;
  BH_HIP_CHECK(hipMalloc(&dev.p, n ? n * rec : 16));
  if (n) {
    BH_HIP_CHECK(hipMalloc(&raw.p, n * rec + 4));
    u32 *flag = (u32 *)((char *)raw.p + n * rec);
    BH_HIP_CHECK(hipMemcpyAsync(raw.p, host_bytes, n * rec, hipMemcpyHostToDevice, ctx->c.stream));
    BH_HIP_CHECK(hipMemsetAsync(flag, 0, 4, ctx->c.stream));
    const u64 threads = (u64)n * cpp;
    hipLaunchKernelGGL(decode_uncompressed_kernel, dim3((u32)((threads + 255) / 256)), dim3(256), 0, ctx->c.stream,
                       (const unsigned char *)raw.p, cpp, (fp_t *)dev.p, (u64)n, flag, (u32 *)nullptr);
    BH_HIP_CHECK(hipGetLastError());
    u32 bad = 0;
    BH_HIP_CHECK(hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, ctx->c.stream));
    BH_HIP_CHECK(hipStreamSynchronize(ctx->c.stream));
    if (bad) return BH_ERR_INVALID_ARG;
  }
  return new_bases(ctx, group, dev.release(), n, true, out);
}
int bh_bases_read_uncompressed(bh_ctx *ctx, int group, const void *host_bytes, size_t n, unsigned flags,
                               bh_bases **out, size_t *bad_index) {
  if (!ctx || !out || (group != BH_G1 && group != BH_G2) || (n && !host_bytes)) return BH_ERR_INVALID_ARG;
  const size_t rec = group == BH_G1 ? 96 : 192;
  const u32 cpp = group == BH_G1 ? 2 : 4;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  DevGuard dev;
  BH_HIP_CHECK(hipMalloc(&dev.p, n ? n * rec : 16));
  if (n) {
    hipStream_t st = ctx->c.stream;
    DevGuard raw;
    const size_t status_off = (n * rec + 15) & ~size_t(15);
    BH_HIP_CHECK(hipMalloc(&raw.p, status_off + n * 4 + 16));
    u32 *status = (u32 *)((char *)raw.p + status_off);
    unsigned long long *min_idx = (unsigned long long *)((char *)status + ((n * 4 + 7) & ~size_t(7)));
    unsigned long long first = ~0ULL;
    u32 first_status = 0;
    BH_HIP_CHECK(hipMemcpyAsync(raw.p, host_bytes, n * rec, hipMemcpyHostToDevice, st));
    BH_HIP_CHECK(hipMemsetAsync(status, 0, n * 4, st));
    BH_HIP_CHECK(hipMemsetAsync(min_idx, 0xff, 8, st));
    const u64 threads = (u64)n * cpp;
    hipLaunchKernelGGL(decode_uncompressed_kernel, dim3((u32)((threads + 255) / 256)), dim3(256), 0, st,
                       (const unsigned char *)raw.p, cpp, (fp_t *)dev.p, (u64)n, (u32 *)nullptr, status);
    BH_HIP_CHECK(hipGetLastError());
    if (flags & BH_POINTS_CHECKED) {
      int r = points_check(group, dev.p, n, status, st);
      if (r != BH_OK) { (void)hipStreamSynchronize(st); return r; }
    }
    const u64 blocks = (n + 255) / 256;
    hipLaunchKernelGGL(first_bad_point_kernel, dim3((u32)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, status,
                       (u64)n, (flags & BH_POINTS_FORBID_IDENTITY) ? 1u : 0u, min_idx);
    BH_HIP_CHECK(hipGetLastError());
    BH_HIP_CHECK(hipMemcpyAsync(&first, min_idx, 8, hipMemcpyDeviceToHost, st));
    BH_HIP_CHECK(hipStreamSynchronize(st));
    if (first != ~0ULL) {
      BH_HIP_CHECK(hipMemcpyAsync(&first_status, status + first, 4, hipMemcpyDeviceToHost, st));
      BH_HIP_CHECK(hipStreamSynchronize(st));
      if (bad_index) *bad_index = (size_t)first;
      return (first_status & PT_INVALID_MASK) ? BH_ERR_INVALID_POINT : BH_ERR_POINT_AT_INFINITY;
    }
  }
  return new_bases(ctx, group, dev.release(), n, true, out);
}
int bh_bases_download(bh_ctx *ctx, const bh_bases *b, size_t first, size_t count, void *out_host) {
  if (!ctx || !b || first + count > b->n) return BH_ERR_INVALID_ARG;
  const size_t rec = b->group == BH_G1 ? 96 : 192;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  if (count) {
    BH_HIP_CHECK(hipMemcpyAsync(out_host, (const char *)b->dev + first * rec, count * rec, hipMemcpyDeviceToHost, ctx->c.stream));
    BH_HIP_CHECK(hipStreamSynchronize(ctx->c.stream));
  }
  return BH_OK;
}
int bh_bases_write_uncompressed(bh_ctx *ctx, const bh_bases *b, size_t first, size_t count, void *out_host_bytes) {
  if (!ctx || !b || first + count > b->n || (count && !out_host_bytes)) return BH_ERR_INVALID_ARG;
  const size_t rec = b->group == BH_G1 ? 96 : 192;
  const u32 cpp = b->group == BH_G1 ? 2 : 4;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  if (!count) return BH_OK;
  void *raw = ctx->c.pool.acquire(count * rec);
  if (!raw) return BH_ERR_HIP;
  const u64 threads = (u64)count * cpp;
  hipLaunchKernelGGL(encode_uncompressed_kernel, dim3((u32)((threads + 255) / 256)), dim3(256), 0, ctx->c.stream,
                     (const fp_t *)((const char *)b->dev + first * rec), cpp, (unsigned char *)raw, (u64)count);
  int rc = hipGetLastError() == hipSuccess ? BH_OK : BH_ERR_HIP;
  if (rc == BH_OK && hipMemcpyAsync(out_host_bytes, raw, count * rec, hipMemcpyDeviceToHost, ctx->c.stream) != hipSuccess) rc = BH_ERR_HIP;
  if (hipStreamSynchronize(ctx->c.stream) != hipSuccess && rc == BH_OK) rc = BH_ERR_HIP;
  ctx->c.pool.release(raw);
  return rc;
}
int bh_bases_copy_dev(bh_ctx *ctx, int group, const void *dev_points, size_t n, bh_bases **out) {
  if (!ctx || !out || (group != BH_G1 && group != BH_G2)) return BH_ERR_INVALID_ARG;
  const size_t rec = group == BH_G1 ? 96 : 192;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  DevGuard dev;
  BH_HIP_CHECK(hipMalloc(&dev.p, n ? n * rec : 16));
  if (n) {
    BH_HIP_CHECK(hipMemcpyAsync(dev.p, dev_points, n * rec, hipMemcpyDeviceToDevice, ctx->c.stream));
    BH_HIP_CHECK(hipStreamSynchronize(ctx->c.stream));
  }
  return new_bases(ctx, group, dev.release(), n, true, out);
}
int bh_bases_wrap_dev(bh_ctx *ctx, int group, const void *dev_points, size_t n, bh_bases **out) {
  if (!ctx || !out || (group != BH_G1 && group != BH_G2)) return BH_ERR_INVALID_ARG;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  return new_bases(ctx, group, const_cast<void *>(dev_points), n, false, out);
}
int bh_bases_precompute(bh_ctx *ctx, bh_bases *b, unsigned window_bits) {
  if (!ctx || !b) return BH_ERR_INVALID_ARG;
  if (b->auto_table) {   // replaced by an explicit table: no longer the context's to drop or to count
    std::lock_guard<std::mutex> g(ctx->c.job_mu);
    auto &v = ctx->c.tables;
    for (size_t i = 0; i < v.size(); i++)
      if (v[i] == b) { v.erase(v.begin() + i); break; }
    const size_t bytes = (size_t)b->tab.W * b->n * table_rec_bytes(b);
    ctx->c.table_bytes = ctx->c.table_bytes > bytes ? ctx->c.table_bytes - bytes : 0;
    b->auto_table = false;
  }
  if (b->table) { (void)hipFree(b->table); b->table = nullptr; b->table_padded = false; }
  if (b->n == 0) return BH_OK;
  const u32 c = window_bits ? window_bits : table_window_bits(b->n, b->group == BH_G2);
  if (c < 2 || c > 24) return BH_ERR_INVALID_ARG;
  const u32 W = (256 + c - 1) / c;
  const size_t rec = b->group == BH_G1 ? 96 : 192;
  if ((u64)W * b->n >= ((u64)1 << 31)) return BH_ERR_INVALID_ARG;   // table rows must fit the 31-bit base field
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  void *t = nullptr;
  if (hipMalloc(&t, (size_t)W * b->n * rec) != hipSuccess) {
    (void)hipGetLastError();
    return BH_ERR_HIP;   // not enough HBM for the table: the caller keeps using the plain bases
  }
  hipStream_t st = ctx->c.stream;
  int rc = BH_OK;
  if (hipMemcpyAsync(t, b->dev, b->n * rec, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = BH_ERR_HIP;
  if (rc == BH_OK) rc = window_table(b->group, t, b->n, c, W, st);
  if (rc == BH_OK && hipStreamSynchronize(st) != hipSuccess) rc = BH_ERR_HIP;
  if (rc != BH_OK) { (void)hipFree(t); return rc; }
  bool padded = false;
  if (table_will_pad(b)) {
    // re-laid at a 128-byte stride (one cache line per gathered record); the dense build buffer is dropped.  If the
    // second allocation fails the dense table is kept - it is the same table
    void *tp = nullptr;
    if (hipMalloc(&tp, (size_t)W * b->n * 128) == hipSuccess) {
      if (hipMemcpy2DAsync(tp, 128, t, rec, rec, (size_t)W * b->n, hipMemcpyDeviceToDevice, st) == hipSuccess &&
          hipStreamSynchronize(st) == hipSuccess) {
        (void)hipFree(t);
        t = tp;
        padded = true;
      } else {
        (void)hipGetLastError();
        (void)hipFree(tp);
      }
    } else {
      (void)hipGetLastError();
    }
  }
  b->table = t;
  b->table_padded = padded;
  b->tab = WindowTable{c, W, (u64)b->n};
  return BH_OK;
}
int bh_bases_table_info(const bh_bases *b, unsigned *window_bits, unsigned *rows, size_t *bytes) {
  if (!b) return BH_ERR_INVALID_ARG;
  if (window_bits) *window_bits = b->table ? b->tab.c : 0;
  if (rows) *rows = b->table ? b->tab.W : 0;
  if (bytes) *bytes = b->table ? (size_t)b->tab.W * b->n * table_rec_bytes(b) : 0;
  return BH_OK;
}
void bh_bases_release(bh_ctx *ctx, bh_bases *b) {
  (void)ctx;
  if (!b) return;
  if (b->auto_table && b->ctx) {
    std::lock_guard<std::mutex> g(b->ctx->c.job_mu);
    auto &v = b->ctx->c.tables;
    for (size_t i = 0; i < v.size(); i++)
      if (v[i] == b) { v.erase(v.begin() + i); break; }
    const size_t bytes = (size_t)b->tab.W * b->n * table_rec_bytes(b);
    b->ctx->c.table_bytes = b->ctx->c.table_bytes > bytes ? b->ctx->c.table_bytes - bytes : 0;
  }
  if (b->table) (void)hipFree(b->table);
  if (b->padded) (void)hipFree(b->padded);
  if (b->owned && b->dev) (void)hipFree(b->dev);
  delete b;
}
size_t bh_bases_len(const bh_bases *b) { return b->n; }

// ---- multiexp -----------------------------------------------------------------------------------
// multiexp of a handful of terms on the host, at issue time (src/multiexp.rs:210-332 semantics incl. the error
// precedence of Appendix A item 6).  Returns false when the case is not eligible (bases beyond the mirrored prefix).
static bool tiny_msm_on_host(const bh_bases *bases, size_t skip, const void *scalars_host, size_t n, int fmt,
                             const uint64_t *density_host, int *rc_out, unsigned char *result) {
  const size_t rec = bases->group == BH_G1 ? 96 : 192;
  const size_t have = bases->host_prefix.size() / rec;
  if (bases->n > have && skip + n > have) return false;   // a base index could fall outside the mirror
  memset(result, 0, rec);
  // pass 1: base index of every dense entry, error conditions
  fr_t sc[TINY_MSM_MAX];
  size_t base_of[TINY_MSM_MAX];
  bool dense[TINY_MSM_MAX];
  size_t cursor = skip;
  bool eof = false, ident = false, ident_top = false;
  // the reference's window size for fewer than 32 terms is 3 (multiexp.rs:318-319): top window = bits [252, 255)
  const u32 lo_ref = 3 * ((255 + 2) / 3 - 1);
  for (size_t i = 0; i < n; i++) {
    dense[i] = density_host ? ((density_host[i >> 6] >> (i & 63)) & 1) : true;
    if (!dense[i]) continue;
    base_of[i] = cursor++;
    memcpy(&sc[i], (const char *)scalars_host + i * 32, 32);
    if (fmt == BH_SCALARS_MONT) {
      fe_from_mont(sc[i], sc[i]);
    } else {
      for (int k = 0; k < 2; k++) {   // values in [q, 2^256) are taken mod q, as on the device
        fr_t t;
        u32 br = 0;
        for (int w = 0; w < 8; w++) t.l[w] = subb(sc[i].l[w], FrParams::mod(w), br, br);
        if (!br) sc[i] = t;
      }
    }
    if (base_of[i] >= bases->n) { eof = true; dense[i] = false; continue; }   // every dense entry checks EOF first
    if (fe_is_zero(sc[i])) continue;                                          // a zero scalar skips its base unseen
    const unsigned char *b = bases->host_prefix.data() + base_of[i] * rec;
    bool is_id = true;
    for (size_t k = 0; k < rec; k++) is_id &= b[k] == 0;
    if (is_id) {
      ident = true;
      bool top = false;
      for (u32 bit = lo_ref; bit < 256; bit++) top |= (sc[i].l[bit >> 5] >> (bit & 31)) & 1;
      if (top && !eof) ident_top = true;   // in the top window, before the first EOF entry
    }
  }
  if (eof && ident) { *rc_out = ident_top ? BH_ERR_UNEXPECTED_IDENTITY : BH_ERR_UNEXPECTED_EOF; return true; }
  if (eof) { *rc_out = BH_ERR_UNEXPECTED_EOF; return true; }
  if (ident) { *rc_out = BH_ERR_UNEXPECTED_IDENTITY; return true; }
  // pass 2: the sum
  alignas(16) unsigned char acc[192], term[192];
  memset(acc, 0, sizeof acc);
  fr_t one;
  fe_zero(one);
  one.l[0] = 1;
  for (size_t i = 0; i < n; i++) {
    if (!dense[i] || fe_is_zero(sc[i])) continue;
    const unsigned char *b = bases->host_prefix.data() + base_of[i] * rec;
    if (fe_eq(sc[i], one)) memcpy(term, b, rec); else host_point_mul(bases->group, term, b, &sc[i]);
    host_point_add(bases->group, acc, acc, term, 1);
  }
  memcpy(result, acc, rec);
  *rc_out = BH_OK;
  return true;
}

static int msm_common(bh_ctx *ctx, const bh_bases *bases, size_t skip, const void *scalars, bool scalars_on_host,
                      size_t n, int fmt, const uint64_t *density, bool density_on_host, size_t density_len,
                      const bh_msm_opts *o, bh_msm_job **out, u64 shard_ref_n = 0, void *after_stream = nullptr) {
  if (!ctx || !bases || !out) return BH_ERR_INVALID_ARG;
  MsmOpts opts;
  if (shard_ref_n) { opts.ref_n = shard_ref_n; opts.always_resolve_ident = true; }
  if (o) {
    if (o->window_bits && (o->window_bits < 2 || o->window_bits > 24)) return BH_ERR_INVALID_ARG;
    opts.c = o->window_bits; opts.chunk = o->chunk; opts.flags = o->flags;
  }
  if (fmt != BH_SCALARS_CANONICAL && fmt != BH_SCALARS_MONT) return BH_ERR_INVALID_ARG;
  if (density && density_len != n) return BH_ERR_INVALID_ARG;   // multiexp.rs:324-329 (assert)
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  // back-pressure (src/multicore.rs:47-73): at the cap the issuing thread completes the oldest job itself
  while (!msm_slot_try_reserve(ctx->c))
    if (!msm_complete_oldest(ctx->c)) std::this_thread::yield();   // the slots are held by calls still issuing
  // ONE snapshot of the handle's table state, taken after the slot is reserved (bh_ctx_trim leaves tables alone while a
  // call is issuing) and under job_mu: both the pointer the job gathers from and the stride it is read at come from it
  // (a trim + bh_bases_precompute between two reads of the handle could pair a freed table with the other stride)
  void *tab_ptr;
  bool ptab;
  WindowTable tab_info;
  {
    std::lock_guard<std::mutex> g(ctx->c.job_mu);
    tab_ptr = bases->table;
    ptab = tab_ptr && bases->table_padded;
    tab_info = bases->tab;
    opts.padded_bases = bases->padded;   // (read by the classic plan only)
  }
  opts.padded_table = ptab ? tab_ptr : nullptr;
  MsmJobImpl *impl = msm_job_new(&ctx->c, bases->group);
  if (!impl) { msm_slot_release(ctx->c); return BH_ERR_HIP; }
  if (n && n <= TINY_MSM_MAX && scalars_on_host && (!density || density_on_host) && !(opts.flags & BH_MSM_NO_SMALL_PATH) &&
      !shard_ref_n) {
    int trc = BH_OK;
    alignas(16) unsigned char res[192];
    if (tiny_msm_on_host(bases, skip, scalars, n, fmt, density, &trc, res)) {
      msm_job_set_result(*impl, trc, trc == BH_OK ? res : nullptr);
      msm_job_track(*impl);   // trivial: only gives the slot back
      *out = new bh_msm_job{impl};
      return BH_OK;
    }
  }
  hipStream_t st = msm_job_stream(*impl);
  const void *sc_dev = scalars;
  const u64 *dn_dev = density;
  int rc = BH_OK;
  if (after_stream) rc = msm_job_after(*impl, (hipStream_t)after_stream);
  if (rc == BH_OK && n && scalars_on_host) {
    void *p = ctx->c.pool.acquire(n * 32);
    if (!p) rc = BH_ERR_HIP;
    else {
      msm_job_own(*impl, p);
      if (hipMemcpyAsync(p, scalars, n * 32, hipMemcpyHostToDevice, st) != hipSuccess) rc = BH_ERR_HIP;
      sc_dev = p;
    }
  }
  if (rc == BH_OK && n && density && density_on_host) {
    const size_t nw = (n + 63) / 64;
    void *p = ctx->c.pool.acquire(nw * 8);
    if (!p) rc = BH_ERR_HIP;
    else {
      msm_job_own(*impl, p);
      if (hipMemcpyAsync(p, density, nw * 8, hipMemcpyHostToDevice, st) != hipSuccess) rc = BH_ERR_HIP;
      dn_dev = (const u64 *)p;
    }
  }
  if (rc == BH_OK)
    rc = msm_job_enqueue(*impl, (tab_ptr && !ptab) ? tab_ptr : bases->dev, bases->n, skip, sc_dev, n, fmt, dn_dev,
                         opts, tab_ptr ? &tab_info : nullptr);
  if (rc != BH_OK) {
    float ms[4];
    unsigned char dummy[192];
    (void)msm_job_finish(*impl, dummy, ms);
    msm_job_delete(impl);
    msm_slot_release(ctx->c);
    return rc;
  }
  msm_job_track(*impl);
  *out = new bh_msm_job{impl};
  return BH_OK;
}
// ---- scalars resident in HBM --------------------------------------------------------------------
int bh_scalars_register(bh_ctx *ctx, const void *scalars_host, size_t n, int scalar_fmt, bh_scalars **out) {
  if (!ctx || !out || (n && !scalars_host)) return BH_ERR_INVALID_ARG;
  if (scalar_fmt != BH_SCALARS_CANONICAL && scalar_fmt != BH_SCALARS_MONT) return BH_ERR_INVALID_ARG;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  void *d = ctx->c.pool.acquire(n ? n * 32 : 32);
  if (!d) return BH_ERR_HIP;
  if (n) {
    // pageable host memory: the copy is staged by the runtime and the call returns once the source may be reused
    if (hipMemcpyAsync(d, scalars_host, n * 32, hipMemcpyHostToDevice, ctx->c.stream) != hipSuccess ||
        hipStreamSynchronize(ctx->c.stream) != hipSuccess) {
      ctx->c.pool.release(d);
      return BH_ERR_HIP;
    }
  }
  *out = new bh_scalars{ctx, d, n, scalar_fmt, true};
  return BH_OK;
}
int bh_scalars_adopt_dev(bh_ctx *ctx, void *scalars_dev, size_t n, int scalar_fmt, int take_ownership, bh_scalars **out) {
  if (!ctx || !out || (n && !scalars_dev)) return BH_ERR_INVALID_ARG;
  if (scalar_fmt != BH_SCALARS_CANONICAL && scalar_fmt != BH_SCALARS_MONT) return BH_ERR_INVALID_ARG;
  *out = new bh_scalars{ctx, scalars_dev, n, scalar_fmt, take_ownership != 0};
  return BH_OK;
}
void bh_scalars_release(bh_scalars *s) {
  if (!s) return;
  if (s->pooled) s->ctx->c.pool.release(s->dev);   // callers release after the multiexps that read it were waited on
  delete s;
}
size_t bh_scalars_len(const bh_scalars *s) { return s ? s->n : 0; }
const void *bh_scalars_dev_ptr(const bh_scalars *s) { return s ? s->dev : nullptr; }
int bh_msm_async_scalars(bh_ctx *ctx, const bh_bases *bases, size_t skip, const bh_scalars *scalars, size_t first,
                         size_t n, const uint64_t *density_words, size_t density_len, const bh_msm_opts *opts,
                         bh_msm_job **job) {
  if (!scalars || scalars->ctx != ctx || first > scalars->n || n > scalars->n - first) return BH_ERR_INVALID_ARG;
  // scalars on the device, density map on the host (a DensityTracker's words, 2^20 bits = 128 KiB)
  return msm_common(ctx, bases, skip, (const char *)scalars->dev + first * 32, false, n, scalars->fmt, density_words, true,
                    density_len, opts, job);
}
int bh_msm_async(bh_ctx *ctx, const bh_bases *bases, size_t skip, const void *scalars_host, size_t n, int fmt,
                 const uint64_t *density_words, size_t density_len, bh_msm_job **job) {
  return msm_common(ctx, bases, skip, scalars_host, true, n, fmt, density_words, true, density_len, nullptr, job);
}
int bh_msm_async_opts(bh_ctx *ctx, const bh_bases *bases, size_t skip, const void *scalars_host, size_t n, int fmt,
                      const uint64_t *density_words, size_t density_len, const bh_msm_opts *opts, bh_msm_job **job) {
  return msm_common(ctx, bases, skip, scalars_host, true, n, fmt, density_words, true, density_len, opts, job);
}
int bh_msm_async_dev(bh_ctx *ctx, const bh_bases *bases, size_t skip, const void *scalars_dev, size_t n, int fmt,
                     const uint64_t *density_words_dev, size_t density_len, bh_msm_job **job) {
  return msm_common(ctx, bases, skip, scalars_dev, false, n, fmt, density_words_dev, false, density_len, nullptr, job);
}
int bh_msm_async_dev_opts(bh_ctx *ctx, const bh_bases *bases, size_t skip, const void *scalars_dev, size_t n, int fmt,
                          const uint64_t *density_words_dev, size_t density_len, const bh_msm_opts *opts,
                          bh_msm_job **job) {
  return msm_common(ctx, bases, skip, scalars_dev, false, n, fmt, density_words_dev, false, density_len, opts, job);
}
// ---- one multiexp over several contexts of this process (one per GPU) -----------------------------
struct bh_msm_sharded_job {
  int group;
  std::vector<bh_msm_job *> jobs;                 // one per shard, in shard order
  std::vector<std::vector<uint64_t>> density;     // per-shard density words (re-based to bit 0), alive until the wait
};
int bh_msm_sharded_async(bh_ctx *const *ctxs, const bh_bases *const *shards, size_t n_shards, size_t skip,
                         const void *scalars_host, size_t n_scalars, int scalar_fmt, const uint64_t *density_words,
                         size_t density_len, bh_msm_sharded_job **out) {
  if (!ctxs || !shards || !n_shards || !out || (n_scalars && !scalars_host)) return BH_ERR_INVALID_ARG;
  if (density_words && density_len != n_scalars) return BH_ERR_INVALID_ARG;   // multiexp.rs:324-329
  for (size_t k = 0; k < n_shards; k++)
    if (!ctxs[k] || !shards[k] || shards[k]->group != shards[0]->group) return BH_ERR_INVALID_ARG;
  const int group = shards[0]->group;
  std::vector<size_t> lens(n_shards);
  for (size_t k = 0; k < n_shards; k++) lens[k] = shards[k]->n;
  std::vector<size_t> cut, off;
  shard_cuts(lens.data(), n_shards, skip, density_words, n_scalars, cut, off);
  std::unique_ptr<bh_msm_sharded_job> sj(new bh_msm_sharded_job());
  sj->group = group;
  sj->density.resize(n_shards);
  int rc = BH_OK;
  for (size_t k = 0; k < n_shards && rc == BH_OK; k++) {
    const size_t lo = cut[k], hi = cut[k + 1] > cut[k] ? cut[k + 1] : cut[k], n = hi - lo;
    // bases before the shard's first used record: only the shard that contains `skip` starts inside itself
    const size_t local_skip = skip > off[k] ? (skip - off[k] < shards[k]->n ? skip - off[k] : shards[k]->n) : 0;
    const uint64_t *dw = nullptr;
    if (density_words && n) {
      std::vector<uint64_t> &d = sj->density[k];
      d.assign((n + 63) / 64, 0);
      const size_t sh = lo & 63, w0 = lo >> 6, nw_all = (n_scalars + 63) / 64;
      for (size_t w = 0; w < d.size(); w++) {
        uint64_t x = density_words[w0 + w] >> sh;
        if (sh && w0 + w + 1 < nw_all) x |= density_words[w0 + w + 1] << (64 - sh);
        d[w] = x;
      }
      if (n & 63) d.back() &= (((uint64_t)1 << (n & 63)) - 1);
      dw = d.data();
    }
    bh_msm_job *j = nullptr;
    const bh_msm_opts o = {0, 0, BH_MSM_NO_SMALL_PATH};
    rc = msm_common(ctxs[k], shards[k], local_skip, (const char *)scalars_host + lo * 32, true, n, scalar_fmt, dw, true,
                    dw ? n : 0, &o, &j, n_scalars ? n_scalars : 1);
    if (rc == BH_OK) sj->jobs.push_back(j);
  }
  if (rc != BH_OK) {
    unsigned char sink[192];
    for (bh_msm_job *j : sj->jobs) (void)bh_msm_wait(j, sink);
    return rc;
  }
  *out = sj.release();
  return BH_OK;
}
int bh_msm_sharded_wait(bh_msm_sharded_job *job, void *out_affine) {
  if (!job || !out_affine) return BH_ERR_INVALID_ARG;
  const size_t rec = job->group == BH_G1 ? 96 : 192;
  alignas(16) unsigned char acc[192], part[192];
  memset(acc, 0, sizeof acc);
  // Error precedence of the whole multiexp from the shards': only the last shard can
  // reach the end of the bases, and every entry of an earlier shard precedes the first EOF entry, so an identity
  // consumed in the reference's top window by ANY shard wins over the EOF; without an EOF any identity is the error.
  bool hip_fail = false, eof = false, ident = false, ident_top = false;
  int other = BH_OK;
  for (size_t k = 0; k < job->jobs.size(); k++) {
    MsmJobImpl *impl = job->jobs[k]->impl;
    const int rc = msm_job_finish(*impl, part, nullptr);
    if (rc < 0) { hip_fail = true; other = rc; }
    eof |= impl->saw_eof;
    ident |= impl->saw_ident;
    // a shard that saw both resolves "top window, before its first EOF entry" itself
    ident_top |= impl->saw_ident_top;
    if (rc == BH_OK) host_point_add(job->group, acc, acc, part, 1);
    msm_job_delete(impl);
    delete job->jobs[k];
  }
  delete job;
  if (hip_fail) return other;
  if (eof && ident) return ident_top ? BH_ERR_UNEXPECTED_IDENTITY : BH_ERR_UNEXPECTED_EOF;
  if (eof) return BH_ERR_UNEXPECTED_EOF;
  if (ident) return BH_ERR_UNEXPECTED_IDENTITY;
  memcpy(out_affine, acc, rec);
  return BH_OK;
}

int bh_msm_async_dev_after(bh_ctx *ctx, const bh_bases *bases, size_t skip, const void *scalars_dev, size_t n, int fmt,
                           const uint64_t *density_words_dev, size_t density_len, const bh_msm_opts *opts, void *after_stream,
                           bh_msm_job **job) {
  if (!after_stream) return BH_ERR_INVALID_ARG;
  return msm_common(ctx, bases, skip, scalars_dev, false, n, fmt, density_words_dev, false, density_len, opts, job, 0,
                    after_stream);
}
int bh_msm_start(bh_msm_job *job) {
  if (!job) return BH_ERR_INVALID_ARG;
  BH_HIP_CHECK(hipSetDevice(job->impl->ctx->device));
  return msm_job_start(*job->impl);
}
int bh_msm_wait_profile(bh_msm_job *job, void *out_affine, float *stage_ms4) {
  if (!job) return BH_ERR_INVALID_ARG;
  float ms[4] = {0, 0, 0, 0};
  int rc = msm_job_finish(*job->impl, out_affine, ms);
  if (stage_ms4) memcpy(stage_ms4, ms, sizeof ms);
  msm_job_delete(job->impl);
  delete job;
  return rc;
}
int bh_msm_wait_stats(bh_msm_job *job, void *out_affine, float *stage_ms4, uint64_t *stats8) {
  if (!job) return BH_ERR_INVALID_ARG;
  float ms[4] = {0, 0, 0, 0};
  u64 st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int rc = msm_job_finish(*job->impl, out_affine, ms, st);
  if (stage_ms4) memcpy(stage_ms4, ms, sizeof ms);
  if (stats8) memcpy(stats8, st, sizeof st);
  msm_job_delete(job->impl);
  delete job;
  return rc;
}
int bh_msm_plan_info(size_t n, int group, unsigned forced_c, unsigned *out9) {
  if (!out9 || (group != BH_G1 && group != BH_G2)) return BH_ERR_INVALID_ARG;
  const MsmPlan p = make_plan(n, forced_c, 0, group == BH_G2);
  out9[0] = p.c; out9[1] = p.W; out9[2] = p.nb; out9[3] = p.chunk; out9[4] = p.chunks_per_window; out9[5] = p.sort_passes;
  out9[6] = p.lo_bits; out9[7] = p.hi_bits; out9[8] = (unsigned)((u64)p.W * p.n);
  return BH_OK;
}
int bh_msm_debug_stages(bh_ctx *ctx, const void *scalars_host, size_t n, int scalar_fmt, unsigned c,
                        uint64_t *pairs_out_host, uint32_t *zstart_out_host) {
  if (!ctx || !scalars_host || !n || !pairs_out_host || !zstart_out_host) return BH_ERR_INVALID_ARG;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  return msm_debug_stages(ctx->c, scalars_host, n, scalar_fmt, c, (u64 *)pairs_out_host, zstart_out_host);
}
int bh_msm_wait_timed(bh_msm_job *job, void *out_affine, float *device_ms) {
  float ms[4];
  int rc = bh_msm_wait_profile(job, out_affine, ms);
  if (device_ms) *device_ms = ms[0];
  return rc;
}
void bh_point_add(int group, void *r, const void *a, const void *b, size_t n) { host_point_add(group, r, a, b, n); }
int bh_msm_wait(bh_msm_job *job, void *out_affine) { return bh_msm_wait_timed(job, out_affine, nullptr); }

int bh_fixed_base_mul_dev(bh_ctx *ctx, int group, const void *base_affine_host, const void *scalars_dev, size_t n,
                          int fmt, void *out_dev, void *stream) {
  if (!ctx || (group != BH_G1 && group != BH_G2)) return BH_ERR_INVALID_ARG;
  if (!n) return BH_OK;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  // the window table of the base (32 x 255 affine records, built on the stream in front of the multiplications): a
  // plain allocation freed with hipFreeAsync-like semantics is not available for pool blocks, so it is a pool block
  // returned once the stream has drained - the call stays asynchronous for a caller-supplied stream only up to here
  hipStream_t st = pick_stream(ctx, stream);
  const size_t rec = group == BH_G1 ? 96 : 192;
  void *table = ctx->c.pool.acquire(FIXED_BASE_TABLE_RECORDS * rec);
  if (!table) return BH_ERR_HIP;
  int rc = fixed_base_mul(group, base_affine_host, scalars_dev, n, fmt, out_dev, st, table);
  if (hipStreamSynchronize(st) != hipSuccess && rc == BH_OK) rc = BH_ERR_HIP;
  ctx->c.pool.release(table);
  return rc;
}

void bh_point_mul(int group, void *r, const void *a, const void *k_canonical) { host_point_mul(group, r, a, k_canonical); }
void bh_point_lincomb(int group, void *r, const void *points, const void *scalars_canonical, size_t n) {
  host_point_lincomb(group, r, points, scalars_canonical, n);
}

}  // extern "C"
