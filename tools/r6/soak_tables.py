#!/usr/bin/env python3
"""Register / multiexp / release churn of vectors that get the automatic 20-bit window tables (2^19 ... 2^22 points, 0.9-7 GB each):
device memory must come back, the context's table accounting must return to zero, and results must repeat.
usage: soak_tables.py [rounds=24]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import bellman_amd
from bellman_amd import _lib
from bench import G1_GEN_MONT, splitmix_scalars

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
lib = _lib.load()
w = bellman_amd.Worker(0)
nmax = 1 << 22
t = splitmix_scalars(nmax, 7)
dt, dout = w.alloc(nmax * 32), w.alloc(nmax * 96)
w.upload(dt, t)
assert lib.bh_fixed_base_mul_dev(w.ctx, 1, G1_GEN_MONT.ctypes.data_as(ctypes.c_void_p), dt, nmax, 0, dout, None) == 0
w.synchronize()
w.upload(dt, splitmix_scalars(nmax, 8))
free = lambda: torch.cuda.mem_get_info()[0] >> 20
ref, marks = {}, []
rnd = np.random.default_rng(3)
for it in range(rounds):
    lg = int(rnd.integers(19, 23))
    n = 1 << lg
    handles = [bellman_amd.Bases.copy_device(w, 1, dout, n) for _ in range(int(rnd.integers(1, 4)))]
    for h in handles:
        assert h.table_info()[:2] == (20, 13), h.table_info()
    assert w.info()["table_bytes"] == sum(h.table_info()[2] for h in handles)
    jobs = [bellman_amd.multiexp(w, h, bellman_amd.FullDensity(), None, scalars_dev=dt, n=n) for h in handles for _ in range(2)]
    for j in jobs:
        r = j.wait()
        assert np.array_equal(ref.setdefault(lg, r), r), lg
    for h in handles:
        h.release()
    w.synchronize()
    assert w.info()["table_bytes"] == 0
    if it % 4 == 3:
        w.trim()
        marks.append(int(free()))
print("free MiB after every 4th round (pool trimmed):", marks)
assert max(marks) - min(marks) <= 256, "device memory does not come back"
print("soak ok (window tables: %d rounds of register / multiexp / release)" % rounds)
