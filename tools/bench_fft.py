"""Time the device-resident FFT entry points (wall clock around synchronised batches)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bellman_amd
from bellman_amd import _lib
from bench import splitmix_scalars

def main():
    lib = _lib.load()
    w = bellman_amd.Worker(0)
    for log_n in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["20", "22", "24"])]:
        n = 1 << log_n
        data = splitmix_scalars(n, 3)
        d = w.alloc(n * 32)
        w.upload(d, data)
        for mode, name in [(0, "fft"), (1, "ifft"), (2, "coset_fft"), (3, "icoset_fft")]:
            for _ in range(2):
                assert lib.bh_fft_fr_dev(w.ctx, d, log_n, mode, None) == 0
            w.synchronize()
            iters = 10
            t0 = time.perf_counter()
            for _ in range(iters):
                lib.bh_fft_fr_dev(w.ctx, d, log_n, mode, None)
            w.synchronize()
            dt = (time.perf_counter() - t0) / iters
            print("log_n=%d %-10s %.3f ms  algorithmic %.1f GB/s (64 B/elem)  %.1f Gbutterfly/s" %
                  (log_n, name, dt * 1e3, 64.0 * n / dt / 1e9, n / 2 * log_n / dt / 1e9), flush=True)
        w.free(d)

main()
