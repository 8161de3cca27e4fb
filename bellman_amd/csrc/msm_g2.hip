// G2 (Fp2) instantiation of the MSM pipeline.  Three kernel bundles over the same records in memory:
//   lane triples (fp2k3.cuh)  - a G2 addition with the latency and the register footprint of a G1 addition: the
//                               latency-bound jobs (accumulation below 2^18 terms; merge + reduction of up to 2^17
//                               buckets, i.e. every window-table plan);
//   lane pairs (fp2pair.cuh)  - schoolbook Fp2 products with one reduction per lane, two wavefronts per SIMD: the
//                               throughput-bound accumulation of large jobs [round 4];
//   one lane per point        - merge + reduction above 2^17 buckets (the classic 16-window plans), and the
//                               accumulation when forced (BH_MSM_G2_SINGLE_LANE; the default of large jobs until round 4).
#include "msm_ec.cuh"
namespace bh {
int msm_enqueue_g2(MsmJobImpl &job, const void *bases_dev, u64 n_bases, u64 skip, const void *scalars_dev, u64 n, int fmt,
                   const u64 *density_dev, const MsmOpts &opts, const WindowTable *table) {
  const bool with_table = table && !(opts.flags & BH_MSM_NO_TABLE) && (opts.c == 0 || opts.c == table->c);
  const MsmPlan pl = with_table ? make_table_plan(n, *table, opts.chunk, true, 256) : make_plan(n, opts.c, opts.chunk, true);
  // 0 = lane triples, 1 = lane pairs, 2 = one lane per point
  const bool f_single = opts.flags & BH_MSM_G2_SINGLE_LANE, f_triples = !f_single && (opts.flags & BH_MSM_G2_LANE_TRIPLES);
  const bool f_pairs = opts.flags & BH_MSM_G2_LANE_PAIRS;   // accumulation only; combines with the two above
  const int acc = f_pairs ? 1 : f_single ? 2 : f_triples ? 0 : (n >= ((u64)1 << 18) ? 1 : 0);
  const bool red_single = f_single ? true : f_triples ? false : (u64)pl.NB > ((u64)1 << 17);
#define BH_G2_CASE(F, FR) return msm_enqueue<F, FR>(job, bases_dev, n_bases, skip, scalars_dev, n, fmt, density_dev, opts, table)
  if (red_single) {
    if (acc == 0) BH_G2_CASE(Fp2K3Ops, Fp2Ops);
    if (acc == 1) BH_G2_CASE(Fp2PairOps, Fp2Ops);
    BH_G2_CASE(Fp2Ops, Fp2Ops);
  }
  if (acc == 0) BH_G2_CASE(Fp2K3Ops, Fp2K3Ops);
  if (acc == 1) BH_G2_CASE(Fp2PairOps, Fp2K3Ops);
  BH_G2_CASE(Fp2Ops, Fp2K3Ops);
#undef BH_G2_CASE
}
BH_INSTANTIATE_MSM_SUPPORT(g2, Fp2Ops)
}

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
int test_g2_k3(Context &c, void *out_add, void *out_madd, void *out_dbl, const void *a_dev, const void *b_dev, u64 n) {
  return test_g2_lanes<Fp2K3Ops>(c, out_add, out_madd, out_dbl, a_dev, b_dev, n);
}
int test_g2_pairs(Context &c, void *out_add, void *out_madd, void *out_dbl, const void *a_dev, const void *b_dev, u64 n) {
  return test_g2_lanes<Fp2PairOps>(c, out_add, out_madd, out_dbl, a_dev, b_dev, n);
}
}  // namespace bh
