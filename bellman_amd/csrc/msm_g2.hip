// G2 (Fp2) instantiation of the MSM pipeline.  Two kernel bundles: the lane-triple K3 form (fp2k3.cuh) - a G2
// addition with the latency and the register footprint of a G1 addition, up to 2x faster on the latency-bound jobs
// of up to 2^17 terms - and one lane per point, which executes ~20 % fewer instructions per addition and wins once
// the accumulation is throughput-bound (profiles/r2_call4_*: 2^17 3.5 vs 3.8 ms, 2^18 5.4 = 5.4, 2^19 9.3 vs 8.0).
// The merge + reduction kernels choose separately, by bucket count: the 2^15 buckets of a window-table plan are
// reduced in lane triples whatever the job size, the 2^19 of a classic 16-window plan one lane per point.
#include "msm_ec.cuh"
namespace bh {
BH_INSTANTIATE_MSM(g2, Fp2Ops, Fp2K3Ops, Fp2Ops, BH_MSM_G2_LANE_TRIPLES, BH_MSM_G2_SINGLE_LANE, ((u64)1 << 18) - 1, (u64)1 << 17)
}

// ---- test hook: the K3 group law on its own (tests/test_gpu_parity.py::test_g2_k3_group_law) -----------------
namespace bh {
__global__ __launch_bounds__(128) void k3_group_law_kernel(XYZZ<Fp2Ops> *r_add, XYZZ<Fp2Ops> *r_madd, XYZZ<Fp2Ops> *r_dbl,
                                                           const Affine<Fp2Ops> *a, const Affine<Fp2Ops> *b, u32 n) {
  typedef Fp2K3Ops F;
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
int test_g2_k3(Context &c, void *out_add, void *out_madd, void *out_dbl, const void *a_dev, const void *b_dev, u64 n) {
  typedef XYZZ<Fp2Ops> Pt;
  if (!n) return BH_OK;
  Pt *d = (Pt *)c.pool.acquire(3 * n * sizeof(Pt));
  if (!d) return BH_ERR_HIP;
  const u32 wpb = workers_per_block<Fp2K3Ops>(128, default_per_wave<Fp2K3Ops>());
  hipLaunchKernelGGL(k3_group_law_kernel, dim3((u32)((n + wpb - 1) / wpb)), dim3(128), 0, c.stream, d, d + n, d + 2 * n,
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
}  // namespace bh
