// G2 (Fp2) instantiation of the MSM pipeline
#include "msm_ec.cuh"
namespace bh {
BH_INSTANTIATE_MSM(g2, Fp2Ops)
}
