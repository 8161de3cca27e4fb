// G1 (Fp) instantiation of the MSM pipeline: one lane per point (no alternative kernel bundle: flag 0 never matches)
#include "msm_ec.cuh"
namespace bh {
BH_INSTANTIATE_MSM(g1, FpOps, FpOps, FpOps, 0u)
}
