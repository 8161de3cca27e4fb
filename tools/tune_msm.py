"""Sweep the MSM tuning knobs (window bits c, chunk K) on the GPU and print per-stage device ms."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bellman_amd
from bellman_amd import _lib
from bench import splitmix_scalars, G1_GEN_MONT

def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    cs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [13, 14, 15, 16]
    ks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [16, 32, 64]
    lib = _lib.load()
    w = bellman_amd.Worker(0)
    n = 1 << log_n
    t = splitmix_scalars(n, 1)
    dt, dout = w.alloc(n * 32), w.alloc(n * 96)
    w.upload(dt, t)
    assert lib.bh_fixed_base_mul_dev(w.ctx, 1, G1_GEN_MONT.ctypes.data_as(ctypes.c_void_p), dt, n, 0, dout, None) == 0
    w.synchronize()
    bases = bellman_amd.Bases.wrap_device(w, 1, dout, n)
    s = splitmix_scalars(n, 2)
    ds = w.alloc(n * 32)
    w.upload(ds, s)
    ref = None
    for c in cs:
        for k in ks:
            lib.bh_msm_set_window_bits(w.ctx, c)
            lib.bh_msm_set_chunk(w.ctx, k)
            best = None
            for it in range(4):
                r, ms = bellman_amd.multiexp(w, bases, bellman_amd.FullDensity(), None, scalars_dev=ds, n=n, timed=True).wait()
                if best is None or ms[0] < best[0]:
                    best = ms
            if ref is None:
                ref = r
            assert np.array_equal(r, ref)
            print("log_n=%d c=%2d K=%3d  total %.3f ms  sort %.3f  accumulate %.3f  reduce %.3f" % (log_n, c, k, *best), flush=True)

main()
