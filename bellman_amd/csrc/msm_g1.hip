// G1 (Fp) instantiation of the MSM pipeline: one lane per point (a single kernel bundle: no flag, no size switch)
#include "msm_ec.cuh"
namespace bh {
int msm_enqueue_g1(MsmJobImpl &job, const void *bases_dev, u64 n_bases, u64 skip, const void *scalars_dev, u64 n, int fmt,
                   const u64 *density_dev, const MsmOpts &opts, const WindowTable *table) {
  return msm_enqueue<FpOps, FpOps>(job, bases_dev, n_bases, skip, scalars_dev, n, fmt, density_dev, opts, table);
}
BH_INSTANTIATE_MSM_SUPPORT(g1, FpOps)
}
