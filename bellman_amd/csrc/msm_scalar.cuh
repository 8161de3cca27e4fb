// scalar loading / digit extraction shared by the MSM stages
#pragma once
#include "msm_types.hpp"

namespace bh {
__device__ __forceinline__ void load_scalar(const void *scalars, u64 i, int fmt, fr_t &s) {
  const uint4 *q = reinterpret_cast<const uint4 *>(scalars) + 2 * i;
  uint4 a = q[0], b = q[1];
  s.l[0] = a.x; s.l[1] = a.y; s.l[2] = a.z; s.l[3] = a.w;
  s.l[4] = b.x; s.l[5] = b.y; s.l[6] = b.z; s.l[7] = b.w;
  if (fmt == BH_SCALARS_MONT) {
    fe_from_mont(s, s);
  } else {
    // The reference's Exponent always comes from a reduced field element (multiexp.rs:172-183).  A canonical
    // buffer handed over the C ABI carries no such guarantee: values in [q, 2^256) are taken mod q (2^256 < 3q)
    // so that the signed-digit recoding's "top window < 2^(c-1)" invariant holds for every input.
#pragma unroll
    for (int k = 0; k < 2; k++) {
      fr_t t;
      u32 br = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) t.l[i] = subb(s.l[i], FrParams::mod(i), br, br);
      if (!br) s = t;
    }
  }
}
__device__ __forceinline__ u32 extract_bits(const fr_t &s, u32 lo, u32 width) {
  // bits [lo, lo+width) of the 256-bit little-endian value (width <= 31)
  if (lo >= 256) return 0;
  u32 w = lo >> 5, sh = lo & 31;
  u64 two = s.l[w];
  if (w + 1 < 8) two |= (u64)s.l[w + 1] << 32;
  return (u32)(two >> sh) & ((1u << width) - 1);
}

}  // namespace bh
