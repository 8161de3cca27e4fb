"""Sweep the MSM tuning knobs (window bits c, chunk K) on the GPU and print per-stage device ms."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bellman_amd
from bellman_amd import _lib
from bench import splitmix_scalars, G1_GEN_MONT

G2_GEN_MONT = np.array([0xF5F28FA202940A10, 0xB3F5FB2687B4961A, 0xA1A893B53E2AE580, 0x9894999D1A3CAEE9, 0x6F67B7631863366B, 0x058191924350BCD7,
                        0xA5A9C0759E23F606, 0xAAA0C59DBCCD60C3, 0x3BB17E18E2867806, 0x1B1AB6CC8541B367, 0xC2B6ED0EF2158547, 0x11922A097360EDF3,
                        0x4C730AF860494C4A, 0x597CFA1F5E369C5A, 0xE7E6856CAA0A635A, 0xBBEFB5E96E0D495F, 0x07D3A975F0EF25A2, 0x0083FD8E7E80DAE5,
                        0xADC0FC92DF64B05D, 0x18AA270A2B1461DC, 0x86ADAC6A3BE4EBA0, 0x79495C4EC93DA33A, 0xE7175850A43CCAED, 0x0B2BC2A163DE1BF2], dtype=np.uint64)

def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    cs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [13, 14, 15, 16]
    ks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [16, 32, 64]
    group = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    words = 12 if group == 1 else 24
    lib = _lib.load()
    w = bellman_amd.Worker(0)
    n = 1 << log_n
    t = splitmix_scalars(n, 1)
    dt, dout = w.alloc(n * 32), w.alloc(n * 8 * words)
    w.upload(dt, t)
    gen = G1_GEN_MONT if group == 1 else G2_GEN_MONT
    assert lib.bh_fixed_base_mul_dev(w.ctx, group, gen.ctypes.data_as(ctypes.c_void_p), dt, n, 0, dout, None) == 0
    w.synchronize()
    bases = bellman_amd.Bases.copy_device(w, group, dout, n)
    s = splitmix_scalars(n, 2)
    ds = w.alloc(n * 32)
    w.upload(ds, s)
    ref = None
    table = os.environ.get("BH_TABLE", "0") == "1"   # BH_TABLE=1: sweep with a window table built for each c
    for c in cs:
        if table:
            import time
            t0 = time.time()
            bases.precompute(c)
            print("  window table c=%d: %s built in %.2f s" % (c, bases.table_info(), time.time() - t0), flush=True)
        for k in ks:
            best = None
            for it in range(4):   # BH_ACC=1 / 2: register / LDS accumulator; BH_FLAGS: any other bh_msm_opts.flags (4 no table, 16 / 32 G2 kernel bundle)
                r, ms = bellman_amd.multiexp(w, bases, bellman_amd.FullDensity(), None, scalars_dev=ds, n=n, timed=True,
                                             window_bits=c, chunk=k, flags=int(os.environ.get('BH_ACC', '0')) | int(os.environ.get('BH_FLAGS', '0'))).wait()
                if best is None or ms[0] < best[0]:
                    best = ms
            if ref is None:
                ref = r
            assert np.array_equal(r, ref)
            print("G%d " % group + "log_n=%d c=%2d K=%3d  total %.3f ms  sort %.3f  accumulate %.3f  reduce %.3f" % (log_n, c, k, *best), flush=True)

main()
