#!/usr/bin/env python3
"""Soak of the last-finisher fold of big bucket runs (csrc/msm_ec.cuh merge_big_runs: piece results published behind
__threadfence + an atomic counter, read by whichever workgroup - on whichever XCD - finishes last): the same boolean-heavy
multiexps issued over and over from several host threads at once, every result compared with the first one (itself checked
against [sum s_i t_i]G).  A lost or stale piece result shows as a mismatch.   usage: soak_long_runs.py [seconds=60] [threads=4]"""
import ctypes, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import bellman_amd
from bellman_amd import _lib
from bench import G1_GEN_MONT, G2_GEN_MONT, splitmix_scalars
from oracle import cref
from tests import scalar_mixes

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lib = _lib.load()
w = bellman_amd.Worker(0)
cases = []
for group, log_n, table in ((1, 20, False), (1, 18, True), (2, 17, True), (1, 22, False)):
    n = 1 << log_n
    words = 12 if group == 1 else 24
    gen = G1_GEN_MONT if group == 1 else G2_GEN_MONT
    t = splitmix_scalars(n, 0x50A + group + log_n)
    dt, dout = w.alloc(n * 32), w.alloc(n * 8 * words)
    w.upload(dt, t)
    assert lib.bh_fixed_base_mul_dev(w.ctx, group, gen.ctypes.data_as(ctypes.c_void_p), dt, n, 0, dout, None) == 0
    w.synchronize()
    bases = bellman_amd.Bases.copy_device(w, group, dout, n)
    if table and bases.table_info()[1] == 0:
        bases.precompute()
    for mix in ("ones", "small90", "bool90"):
        sc = scalar_mixes.scalars(mix, n, 0xABC + log_n)
        ds = w.alloc(n * 32)
        w.upload(ds, sc)
        want = bellman_amd.multiexp(w, bases, bellman_amd.FullDensity(), None, scalars_dev=ds, n=n).wait()
        k = cref.fr_dot(sc, t)
        assert np.array_equal(want, cref.point_mul(group, cref.g1_generator() if group == 1 else cref.g2_generator(), k)), (group, log_n, mix)
        cases.append((group, log_n, mix, bases, ds, n, want))
    w.free(dt)
print("%d cases verified; soaking for %.0f s on %d host threads" % (len(cases), seconds, nthreads), flush=True)
stop = time.time() + seconds
counts, bad = [0] * nthreads, []


def run(tid):
    i = tid
    while time.time() < stop and not bad:
        group, log_n, mix, bases, ds, n, want = cases[i % len(cases)]
        jobs = [bellman_amd.multiexp(w, bases, bellman_amd.FullDensity(), None, scalars_dev=ds, n=n) for _ in range(2)]
        for j in jobs:
            got = j.wait()
            if not np.array_equal(got, want):
                bad.append((group, log_n, mix, counts[tid]))
        counts[tid] += 2
        i += 1


threads = [threading.Thread(target=run, args=(t,)) for t in range(nthreads)]
for t in threads:
    t.start()
for t in threads:
    t.join()
print("multiexps: %d, mismatches: %d %s" % (sum(counts), len(bad), bad[:3]))
sys.exit(1 if bad else 0)
