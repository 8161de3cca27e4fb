// Where a multiexp over several contexts (bh_msm_sharded_async, api.hip) cuts its exponents; shared with the test
// library's hook (test_hooks.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace bh {
// index of the t-th (0-based) set bit of an LSB0 bitmap of n bits; n when there are not that many
inline size_t select_bit(const uint64_t *words, size_t n, size_t t) {
  const size_t nw = (n + 63) / 64;
  for (size_t w = 0; w < nw; w++) {
    uint64_t x = words[w];
    if (w == nw - 1 && (n & 63)) x &= (((uint64_t)1 << (n & 63)) - 1);
    const size_t pc = (size_t)__builtin_popcountll(x);
    if (t < pc) {
      for (;; x &= x - 1, t--)
        if (t == 0) return w * 64 + (size_t)__builtin_ctzll(x);
    }
    t -= pc;
  }
  return n;
}
// scalar index at which shard k starts (cut[k]) and the shard's first base index (off[k]): shard k computes the
// scalars [cut[k], cut[k+1]) - the dense entries whose base index skip + rank falls into [off[k], off[k+1]) plus the
// non-dense entries between them; the last shard takes everything left (it is the one that can run out of bases)
inline void shard_cuts(const size_t *lens, size_t n_shards, size_t skip, const uint64_t *density_words, size_t n_scalars,
                       std::vector<size_t> &cut, std::vector<size_t> &off) {
  cut.assign(n_shards + 1, 0);
  off.assign(n_shards + 1, 0);
  for (size_t k = 0; k < n_shards; k++) off[k + 1] = off[k] + lens[k];
  for (size_t k = 1; k < n_shards; k++) {
    if (off[k] <= skip) { cut[k] = 0; continue; }
    const size_t t = off[k] - skip;   // dense entries that precede the shard
    cut[k] = density_words ? select_bit(density_words, n_scalars, t) : (t < n_scalars ? t : n_scalars);
  }
  cut[n_shards] = n_scalars;
}
}  // namespace bh
