"""all four transforms == the restated best_fft at the sizes where the pass plan / table kind switches (the library under
test is whatever BELLMAN_HIP_LIB names): python tools/r5/fft_parity_quick.py [log_n ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bellman_amd
from oracle import cref

w = bellman_amd.Worker(0)
sizes = [int(x) for x in sys.argv[1:]] or [3, 10, 11, 12, 13, 17, 21, 22]
threads = cref.lib().orc_max_threads()
bad = 0
for log_n in sizes:
    n = 1 << log_n
    data = cref.random_fr(n, 5500 + log_n)
    for mode in (0, 1, 2, 3):
        d = bellman_amd.EvaluationDomain.from_coeffs(w, data)
        [d.fft, d.ifft, d.coset_fft, d.icoset_fft][mode]()
        ok = np.array_equal(d.into_coeffs(), cref.fft(data, mode, threads=threads))
        bad += not ok
        if not ok:
            print("MISMATCH", log_n, mode, flush=True)
print("parity", "ok" if not bad else "FAILED", sizes, flush=True)
