// G1 (Fp) instantiation of the MSM pipeline
#include "msm_ec.cuh"
namespace bh {
BH_INSTANTIATE_MSM(g1, FpOps)
}
