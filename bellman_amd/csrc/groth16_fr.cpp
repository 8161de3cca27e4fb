// bellman::Fr: the scalar field on the host (4 x 64-bit Montgomery limbs), used by circuit synthesis and
// by the handful of scalars create_proof / generate_parameters handle on the CPU.  See groth16.hpp.
#include "groth16.hpp"

#include <string.h>

namespace bellman {
namespace {
const uint64_t *const FR_R2 = fr_detail::R2;
const uint64_t *const FR_MOD = fr_detail::MOD;
}  // namespace

// little-endian 512-bit integer -> Fr (ff's wide reduction behind Field::random): lo + hi * 2^256 mod q
Fr Fr::from_u512(const uint64_t limbs[8]) {
  Fr lo, hi, r2;
  memcpy(lo.l, limbs, 32);
  memcpy(hi.l, limbs + 4, 32);
  memcpy(r2.l, FR_R2, 32);
  // mont_mul(R^2, x) = x * R mod q for any 256-bit x: the Montgomery form of x mod q (the FIRST operand of the product has
  // to be below q, the second may be any 256-bit value: groth16.hpp)
  const Fr lo_m = r2 * lo, hi_m = r2 * hi;
  return lo_m + hi_m * r2;   // Montgomery form of 2^256 is R * R = R^2
}
Fr Fr::pow_vartime(uint64_t e) const {
  Fr acc = Fr::one();
  for (int i = 63; i >= 0; i--) {
    acc = acc * acc;
    if ((e >> i) & 1) acc = acc * *this;
  }
  return acc;
}
Fr Fr::invert() const {   // a^(q-2)
  uint64_t e[4] = {FR_MOD[0] - 2, FR_MOD[1], FR_MOD[2], FR_MOD[3]};   // no borrow: the low limb ends in ...00000001 + 0xffffffff00000000
  Fr acc = Fr::one();
  for (int i = 255; i >= 0; i--) {
    acc = acc * acc;
    if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * *this;
  }
  return acc;
}
}  // namespace bellman
