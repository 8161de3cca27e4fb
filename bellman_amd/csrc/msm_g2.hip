// G2 (Fp2) instantiation of the MSM pipeline.  Three kernel bundles over the same records in memory:
//   lane triples (fp2k3.cuh)  - a G2 addition with the latency and the register footprint of a G1 addition: the
//                               latency-bound jobs (accumulation below 2^15 terms; merge + reduction of up to 2^17
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
  // (profiles/archive/r4_call8.txt: pairs from 2^15 terms - 2^15 1.57 vs 1.73 ms, 2^16 1.67 vs 1.96, 2^17 2.20 vs 2.88; equal at 2^14)
  const int acc = f_pairs ? 1 : f_single ? 2 : f_triples ? 0 : (n >= ((u64)1 << 15) ? 1 : 0);
  // [r6] ... but a big TABLE set (the 2^19 buckets of 20-bit rows) stays on lane triples: with the two-stage sums of msm_ec.cuh
  // its reduction is ~1.5 ms there, 2.8 with one lane per point (profiles/r6_call38_*, r6_call39_*)
  static const bool big_table_triples = [] { const char *e = getenv("BELLMAN_HIP_G2_BIG_TABLE_TRIPLES"); return !(e && *e == '0'); }();
  const bool red_single = f_single ? true : f_triples ? false : ((u64)pl.NB > ((u64)1 << 17) && !(with_table && big_table_triples));
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
