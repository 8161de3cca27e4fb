#!/usr/bin/env python3
"""bh_msm_wait_stats against a host count: the additions a multiexp executes = non-zero digits - (chunk, bucket) segments.
usage: stats_check.py [log_n=20] [mix=uniform]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bellman_amd
from bellman_amd import _lib
from bench import G1_GEN_MONT, splitmix_scalars
from tests import scalar_mixes

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mix = sys.argv[2] if len(sys.argv) > 2 else "uniform"
n = 1 << log_n
lib = _lib.load()
w = bellman_amd.Worker(0)
t = splitmix_scalars(n, 5)
dt, dout = w.alloc(n * 32), w.alloc(n * 96)
w.upload(dt, t)
assert lib.bh_fixed_base_mul_dev(w.ctx, 1, G1_GEN_MONT.ctypes.data_as(ctypes.c_void_p), dt, n, 0, dout, None) == 0
w.synchronize()
bases = bellman_amd.Bases.copy_device(w, 1, dout, n)
sc = scalar_mixes.scalars(mix, n, 9)
w.upload(dt, sc)
for it in range(3):
    got, ms, st = bellman_amd.multiexp(w, bases, bellman_amd.FullDensity(), None, scalars_dev=dt, n=n, stats=True).wait()
print(mix, "2^%d" % log_n, ms, st)
c, K, W = st["window_bits"], st["chunk"], st["bucket_sets"]
if W > 1:
    # host recount: signed c-bit digits with carry (csrc/msm_stages.hip msm_digits_kernel)
    v = [int.from_bytes(sc[i].tobytes(), "little") for i in range(min(n, 1 << 16))]
    nz = 0
    for x in v:
        carry = 0
        for j in range(W):
            d = ((x >> (c * j)) & ((1 << c) - 1)) + carry
            carry = 0
            if d > (1 << (c - 1)):
                d -= 1 << c
                carry = 1
            nz += d != 0
    print("non-zero digits in the first %d scalars: %d of %d (device, all scalars: %d of %d)" %
          (len(v), nz, len(v) * W, st["sorted_entries"] - st["zero_digits"], st["sorted_entries"]))
live = st["sorted_entries"] - st["zero_digits"]
print("live entries %d, additions %d, copies (bucket / chunk openers) %d = %.2f %% of the live entries" %
      (live, st["mixed_additions"], live - st["mixed_additions"], 100.0 * (live - st["mixed_additions"]) / max(1, live)))
