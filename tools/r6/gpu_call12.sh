#!/bin/bash
# round 6, call 12: UPPER BOUND of what keeping FFT tile elements in 30-bit limbs could save, by timing-only
# diagnostic builds (results are wrong by construction): no operand slicing and no repack in the products (lib_noslice),
# carry-free limb-wise additions / subtractions (lib_cheap), both (lib_noslice_cheap); same box, alternating, 3 rounds
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6c12
mkdir -p $OUT
run() { python - <<'PY'
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bellman_amd
from bellman_amd import _lib
from bench import splitmix_scalars
lib = _lib.load(); w = bellman_amd.Worker(0)
for log_n in (20, 22, 24):
    n = 1 << log_n
    d = w.alloc(n * 32); w.upload(d, splitmix_scalars(n, 3))
    for i in range(40): lib.bh_fft_fr_dev(w.ctx, d, log_n, i & 3, None)
    w.synchronize()
    res = []
    for mode in range(4):
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(10): lib.bh_fft_fr_dev(w.ctx, d, log_n, mode, None)
            w.synchronize(); ts.append((time.perf_counter() - t0) / 10 * 1e3)
        res.append(sorted(ts)[2])
    print("%-16s 2^%d fft %.4f ifft %.4f coset %.4f icoset %.4f ms" % (os.environ.get("VARIANT", "shipped"), log_n, *res), flush=True)
    w.free(d)
PY
}
for rep in 1 2 3; do
  VARIANT=shipped run
  for v in noslice cheap noslice_cheap; do
    VARIANT=$v BELLMAN_HIP_ALLOW_LIB_OVERRIDE=1 BELLMAN_HIP_LIB=$GRAFT_REPO_ROOT/bellman_amd/lib_$v/libbellman_hip.so run
  done
done 2>&1 | tee $OUT/fft_limb_upper_bound.txt
