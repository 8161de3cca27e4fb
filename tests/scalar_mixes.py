"""Seeded scalar vectors in the shapes real witnesses have ("boolean-heavy vectors (~50 % zeros/ones, as
real witnesses produce - the reason Exponent::Zero/One exist)", src/multiexp.rs:172-182,245-252; the SHA-256 gadget's
aux assignment is almost entirely booleans, src/gadgets/sha256.rs:307-331).  Shared by the GPU parity tests, bench.py and
the profiling tools: inputs only, no expected values."""

import numpy as np

MIXES = ("uniform", "bool50", "bool90", "ones", "small90")


def _splitmix(n, seed):
    from bench import splitmix_scalars

    return splitmix_scalars(n, seed)


def scalars(mix, n, seed):
    """[n,4] uint64 canonical scalars < q:
      uniform  SplitMix64 limbs (the bench's generator)
      bool50   25 % zeros, 25 % ones, 50 % uniform
      bool90   45 % zeros, 45 % ones, 10 % uniform
      ones     every scalar = 1 (one bucket of window 0 holds the whole vector)
      small90  90 % below 2^8 (uniform bytes, zero included), 10 % uniform"""
    sc = _splitmix(n, seed)
    if mix == "uniform":
        return sc
    rnd = np.random.default_rng(seed ^ 0xB001)
    u = rnd.random(n)
    if mix in ("bool50", "bool90"):
        p = 0.25 if mix == "bool50" else 0.45
        zero, one = u < p, (u >= p) & (u < 2 * p)
        sc[zero | one] = 0
        sc[one, 0] = 1
    elif mix == "ones":
        sc[:] = 0
        sc[:, 0] = 1
    elif mix == "small90":
        small = u < 0.9
        sc[small] = 0
        sc[small, 0] = rnd.integers(0, 256, int(small.sum()), dtype=np.uint64)
    else:
        raise ValueError(mix)
    return np.ascontiguousarray(sc)
