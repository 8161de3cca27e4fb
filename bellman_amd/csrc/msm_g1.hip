// G1 (Fp) instantiation of the MSM pipeline: one lane per point (a single kernel bundle: no flag, no size switch)
#include "msm_ec.cuh"
namespace bh {
BH_INSTANTIATE_MSM(g1, FpOps, FpOps, FpOps, 0u, 0u, ~(u64)0, ~(u64)0)
}
