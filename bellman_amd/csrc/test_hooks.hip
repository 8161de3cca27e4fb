// libbellman_hip_test.so: the test hooks of include/bellman_hip_test.h (field / group arithmetic on its own, the MSM
// sort stages, plan and shard-cut helpers) with the kernels only they use.  Links against libbellman_hip.so and is NOT
// part of the product: the shipped library exports exactly include/bellman_hip.h (tests/test_abi_cpu.py).
#include <string.h>

#include "../../include/bellman_hip_test.h"
#include "msm_ec.cuh"
#include "shard_cuts.hpp"

namespace bh {
template <class P>
__global__ void fe_mul_kernel(Fe<P> *r, const Fe<P> *a, const Fe<P> *b, u64 n) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<P> x = a[i], y = b[i], z;
  fe_mul(z, x, y);
  r[i] = z;
}
template <class P>
static int test_fe_mul(void *r, const void *a, const void *b, u64 n, hipStream_t st) {
  if (!n) return BH_OK;
  hipLaunchKernelGGL(fe_mul_kernel<P>, dim3((u32)((n + 255) / 256)), dim3(256), 0, st, (Fe<P> *)r, (const Fe<P> *)a,
                     (const Fe<P> *)b, n);
  BH_HIP_CHECK(hipGetLastError());
  return BH_OK;
}
}  // namespace bh

// ---- test hook: the group law in the multi-lane forms on its own (tests/test_gpu_parity.py::test_g2_k3_group_law,
// test_g2_lane_pair_group_law) --------------------------------------------------------------------------------
namespace bh {
template <class F>
__global__ __launch_bounds__(128) void lanes_group_law_kernel(XYZZ<Fp2Ops> *r_add, XYZZ<Fp2Ops> *r_madd, XYZZ<Fp2Ops> *r_dbl,
                                                              const Affine<Fp2Ops> *a, const Affine<Fp2Ops> *b, u32 n) {
  u32 in_block, i;
  if (!worker_index<F>(default_per_wave<F>(), in_block, i) || i >= n) return;
  Affine<F> pa, pb;
  load_affine<F>(pa, a + i);
  load_affine<F>(pb, b + i);
  XYZZ<F> x, y, z;
  xyzz_from_affine(x, pa);
  xyzz_from_affine(y, pb);
  xyzz_add(z, x, y);
  store_xyzz<F>(&r_add[i], z);
  z = x;
  if (!aff_is_identity(pb)) xyzz_madd(z, pb);
  store_xyzz<F>(&r_madd[i], z);
  xyzz_dbl(z, x);
  store_xyzz<F>(&r_dbl[i], z);
}
// out_*: n affine records each on the HOST (XYZZ results converted with the host arithmetic)
template <class F>
static int test_g2_lanes(Context &c, void *out_add, void *out_madd, void *out_dbl, const void *a_dev, const void *b_dev, u64 n) {
  typedef XYZZ<Fp2Ops> Pt;
  if (!n) return BH_OK;
  Pt *d = (Pt *)c.pool.acquire(3 * n * sizeof(Pt));
  if (!d) return BH_ERR_HIP;
  const u32 wpb = workers_per_block<F>(128, default_per_wave<F>());
  hipLaunchKernelGGL(lanes_group_law_kernel<F>, dim3((u32)((n + wpb - 1) / wpb)), dim3(128), 0, c.stream, d, d + n, d + 2 * n,
                     (const Affine<Fp2Ops> *)a_dev, (const Affine<Fp2Ops> *)b_dev, (u32)n);
  int rc = hipGetLastError() == hipSuccess ? BH_OK : BH_ERR_HIP;
  std::vector<Pt> h(3 * n);
  if (rc == BH_OK && hipMemcpyAsync(h.data(), d, 3 * n * sizeof(Pt), hipMemcpyDeviceToHost, c.stream) != hipSuccess) rc = BH_ERR_HIP;
  if (hipStreamSynchronize(c.stream) != hipSuccess && rc == BH_OK) rc = BH_ERR_HIP;
  c.pool.release(d);
  if (rc != BH_OK) return rc;
  typedef HostFp2Ops H;
  void *outs[3] = {out_add, out_madd, out_dbl};
  for (int k = 0; k < 3; k++)
    for (u64 i = 0; i < n; i++) {
      Pt q = h[k * n + i];
      Fp2Ops::canon(q.x); Fp2Ops::canon(q.y); Fp2Ops::canon(q.zz); Fp2Ops::canon(q.zzz);   // lazily reduced on the device
      XYZZ<H> hq;
      memcpy(&hq, &q, sizeof hq);
      Affine<H> aff;
      xyzz_to_affine(aff, hq);
      memcpy((char *)outs[k] + i * sizeof aff, &aff, sizeof aff);
    }
  return BH_OK;
}
// the lane-sextet (K6) general addition of the merge kernels on its own: r[i] = a[i] + b[i], one sextet per pair of points
__global__ __launch_bounds__(64) void k6_add_kernel(XYZZ<Fp2Ops> *r, const Affine<Fp2Ops> *a, const Affine<Fp2Ops> *b, u32 n) {
  const u32 t = (k3_lane() * 43u) >> 8;
  const u32 i = blockIdx.x * K6_PER_WAVE + t;
  if (t >= K6_PER_WAVE || i >= n) return;
  const bool y = k6_side();
  auto from_affine = [&](HalfPt &h, const Affine<Fp2Ops> *p) {
    Fp2K3Ops::load(h.u, y ? &p->y : &p->x);
    fp_t ox, oy;
    Fp2K3Ops::load(ox, &p->x);
    Fp2K3Ops::load(oy, &p->y);
    if (Fp2K3Ops::is_zero_canonical(ox, oy)) { fe_zero(h.u); fe_zero(h.v); }   // the all-zero record is the identity
    else Fp2K3Ops::one(h.v);
  };
  HalfPt pa, pb;
  from_affine(pa, a + i);
  from_affine(pb, b + i);
  k6_add(pa, pa, pb);
  k6_store(&r[i], pa);
}
static int test_g2_k6(Context &c, void *out_add, const void *a_dev, const void *b_dev, u64 n) {
  typedef XYZZ<Fp2Ops> Pt;
  if (!n) return BH_OK;
  Pt *d = (Pt *)c.pool.acquire(n * sizeof(Pt));
  if (!d) return BH_ERR_HIP;
  hipLaunchKernelGGL(k6_add_kernel, dim3((u32)((n + K6_PER_WAVE - 1) / K6_PER_WAVE)), dim3(64), 0, c.stream, d,
                     (const Affine<Fp2Ops> *)a_dev, (const Affine<Fp2Ops> *)b_dev, (u32)n);
  int rc = hipGetLastError() == hipSuccess ? BH_OK : BH_ERR_HIP;
  std::vector<Pt> h(n);
  if (rc == BH_OK && hipMemcpyAsync(h.data(), d, n * sizeof(Pt), hipMemcpyDeviceToHost, c.stream) != hipSuccess) rc = BH_ERR_HIP;
  if (hipStreamSynchronize(c.stream) != hipSuccess && rc == BH_OK) rc = BH_ERR_HIP;
  c.pool.release(d);
  if (rc != BH_OK) return rc;
  typedef HostFp2Ops H;
  for (u64 i = 0; i < n; i++) {
    Pt q = h[i];
    Fp2Ops::canon(q.x); Fp2Ops::canon(q.y); Fp2Ops::canon(q.zz); Fp2Ops::canon(q.zzz);   // lazily reduced on the device
    XYZZ<H> hq;
    memcpy(&hq, &q, sizeof hq);
    Affine<H> aff;
    xyzz_to_affine(aff, hq);
    memcpy((char *)out_add + i * sizeof aff, &aff, sizeof aff);
  }
  return BH_OK;
}
static int test_g2_k3(Context &c, void *out_add, void *out_madd, void *out_dbl, const void *a_dev, const void *b_dev, u64 n) {
  return test_g2_lanes<Fp2K3Ops>(c, out_add, out_madd, out_dbl, a_dev, b_dev, n);
}
static int test_g2_pairs(Context &c, void *out_add, void *out_madd, void *out_dbl, const void *a_dev, const void *b_dev, u64 n) {
  return test_g2_lanes<Fp2PairOps>(c, out_add, out_madd, out_dbl, a_dev, b_dev, n);
}
}  // namespace bh

using namespace bh;
extern "C" {
int bh_test_shard_cuts(const size_t *lens, size_t n_shards, size_t skip, const uint64_t *density_words, size_t n_scalars,
                       size_t *cuts_out) {
  if (!lens || !n_shards || !cuts_out) return BH_ERR_INVALID_ARG;
  std::vector<size_t> cut, off;
  shard_cuts(lens, n_shards, skip, density_words, n_scalars, cut, off);
  memcpy(cuts_out, cut.data(), (n_shards + 1) * sizeof(size_t));
  return BH_OK;
}
size_t bh_test_pool_size_class(size_t bytes) { return DevicePool::size_class(bytes); }
int bh_test_fr_mul_dev(bh_ctx *ctx, void *r, const void *a, const void *b, size_t n) {
  int rc = test_fe_mul<FrParams>(r, a, b, n, ctx->c.stream);
  if (rc == BH_OK) BH_HIP_CHECK(hipStreamSynchronize(ctx->c.stream));
  return rc;
}
int bh_test_fp_mul_dev(bh_ctx *ctx, void *r, const void *a, const void *b, size_t n) {
  int rc = test_fe_mul<FpParams>(r, a, b, n, ctx->c.stream);
  if (rc == BH_OK) BH_HIP_CHECK(hipStreamSynchronize(ctx->c.stream));
  return rc;
}
int bh_test_point_add_dev(bh_ctx *ctx, int group, void *r, const void *a, const void *b, size_t n) {
  int rc = group == BH_G1 ? test_point_add_t<FpOps>(r, a, b, n, ctx->c.stream) : test_point_add_t<Fp2Ops>(r, a, b, n, ctx->c.stream);
  if (rc == BH_OK) BH_HIP_CHECK(hipStreamSynchronize(ctx->c.stream));
  return rc;
}
int bh_test_g2_k3_dev(bh_ctx *ctx, void *out_add_host, void *out_madd_host, void *out_dbl_host, const void *a_dev,
                      const void *b_dev, size_t n) {
  if (!ctx) return BH_ERR_INVALID_ARG;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  return test_g2_k3(ctx->c, out_add_host, out_madd_host, out_dbl_host, a_dev, b_dev, n);
}
int bh_test_g2_k6_dev(bh_ctx *ctx, void *out_add_host, const void *a_dev, const void *b_dev, size_t n) {
  if (!ctx) return BH_ERR_INVALID_ARG;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  return test_g2_k6(ctx->c, out_add_host, a_dev, b_dev, n);
}
int bh_test_g2_pairs_dev(bh_ctx *ctx, void *out_add_host, void *out_madd_host, void *out_dbl_host, const void *a_dev,
                         const void *b_dev, size_t n) {
  if (!ctx) return BH_ERR_INVALID_ARG;
  BH_HIP_CHECK(hipSetDevice(ctx->c.device));
  return test_g2_pairs(ctx->c, out_add_host, out_madd_host, out_dbl_host, a_dev, b_dev, n);
}
void bh_test_fr_mul_host(void *r, const void *a, const void *b, size_t n) {
  for (size_t i = 0; i < n; i++) fe_mul(((fr_t *)r)[i], ((const fr_t *)a)[i], ((const fr_t *)b)[i]);
}
void bh_test_fr_mul_bform_host(void *r, const void *a, const void *b, size_t n) {
  // the FFT's multiplier: second operand pre-sliced into 30-bit limbs (ff.cuh fe_to_bform / fe_mul_b)
  for (size_t i = 0; i < n; i++) {
    u32 B[9];
    fe_to_bform<FrParams>(B, ((const fr_t *)b)[i]);
    fe_mul_b<FrParams>(((fr_t *)r)[i], ((const fr_t *)a)[i], B);
  }
}
void bh_test_fp_mul_host(void *r, const void *a, const void *b, size_t n) {
  for (size_t i = 0; i < n; i++) fe_mul(((fp_t *)r)[i], ((const fp_t *)a)[i], ((const fp_t *)b)[i]);
}
int bh_test_fp_lazy_host(int op, void *r, const void *a, const void *b) {
  // the lazily reduced Fp helpers of the curve code (ff.cuh), compiled for the host; operands in [0, 2p)
  fp_t x, y, z;
  memcpy(&x, a, sizeof x);
  if (b) memcpy(&y, b, sizeof y); else fe_zero(y);
  int flag = 0;
  switch (op) {
    case 0: fpl_add(z, x, y); break;
    case 1: fpl_sub(z, x, y); break;
    case 2: fpl_neg(z, x); break;
    case 3: fpl_canon(z, x); break;
    case 4: z = x; flag = fpl_is_zero(x) ? 1 : 0; break;
    case 5: z = fp_mul_call(x, y); break;   // lazily reduced Montgomery product
    case 6: z = fp_sqr_call(x); break;
    case 7: z = x; flag = fpl_eq(x, y) ? 1 : 0; break;
    default: return BH_ERR_INVALID_ARG;
  }
  memcpy(r, &z, sizeof z);
  return flag;
}
void bh_test_fr_inv_host(void *r, const void *a, size_t n) {
  for (size_t i = 0; i < n; i++) fe_inv(((fr_t *)r)[i], ((const fr_t *)a)[i]);
}
void bh_test_point_add_host(int group, void *r, const void *a, const void *b, size_t n) {
  if (group == BH_G1) devhdr_point_add_t<FpOps>(r, a, b, n); else devhdr_point_add_t<Fp2Ops>(r, a, b, n);
}
void bh_test_point_mul_host(int group, void *r, const void *a, const void *k) {
  if (group == BH_G1) devhdr_point_mul_t<FpOps>(r, a, (const u32 *)k); else devhdr_point_mul_t<Fp2Ops>(r, a, (const u32 *)k);
}
}  // extern "C"
