#!/usr/bin/env python3
"""Stage times of a multiexp per scalar mix (tests/scalar_mixes.py) - uniform against boolean-heavy vectors - with the
size-independent check sum_i s_i [t_i]G == [sum_i s_i t_i]G on every result.
usage: boolean_mix.py [group=1] [log_n=20] [table=0|1] [density=0|1] [iters=7]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import bellman_amd  # noqa: E402
from bellman_amd import _lib  # noqa: E402
from bench import G1_GEN_MONT, G2_GEN_MONT, splitmix_scalars  # noqa: E402
from oracle import cref  # noqa: E402
from tests import scalar_mixes  # noqa: E402


def main():
    group = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    table = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    dens = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 7
    n = 1 << log_n
    lib = _lib.load()
    w = bellman_amd.Worker(0)
    words = 12 if group == 1 else 24
    gen = G1_GEN_MONT if group == 1 else G2_GEN_MONT
    t = splitmix_scalars(n, 0x7157 + group)
    dt, dout = w.alloc(n * 32), w.alloc(n * 8 * words)
    w.upload(dt, t)
    assert lib.bh_fixed_base_mul_dev(w.ctx, group, gen.ctypes.data_as(ctypes.c_void_p), dt, n, 0, dout, None) == 0
    w.synchronize()
    bases = bellman_amd.Bases.copy_device(w, group, dout, n)
    if table and bases.table_info()[1] == 0:
        bases.precompute()
    print("# G%d 2^%d table=%s density=%d  (table c, rows, bytes = %s)" % (group, log_n, table, dens, bases.table_info()))
    bits = None
    dmap = bellman_amd.FullDensity()
    tt = t
    if dens:
        bits = np.random.default_rng(11).random(n) < 0.5
        dmap = bellman_amd.DensityTracker()
        dmap.bv = bits
    print("%-8s %9s %9s %9s %9s %9s %9s  %s" % ("mix", "wall", "pipeline", "sort", "accum", "reduce", "M/s", "x uniform"))
    base_rate = None
    only = os.environ.get("MIX")
    for mix in ([only] if only else scalar_mixes.MIXES):
        sc = scalar_mixes.scalars(mix, n, 0x5EED + group)
        walls, stages = [], []
        got = None
        for it in range(iters + 2):
            t0 = time.perf_counter()
            if dens:
                got, ms = bellman_amd.multiexp(w, bases, dmap, sc, timed=True, flags=0 if table else bellman_amd.multiexp.__globals__["NO_TABLE"]).wait()
            else:
                if it == 0:
                    w.upload(dt, sc)
                got, ms = bellman_amd.multiexp(w, bases, dmap, None, scalars_dev=dt, n=n, timed=True,
                                               flags=0 if table else bellman_amd.multiexp.__globals__["NO_TABLE"]).wait()
            if it >= 2:
                walls.append((time.perf_counter() - t0) * 1e3)
                stages.append(ms)
        if dens:
            k = cref.fr_dot(sc[bits], tt[: int(bits.sum())])
        else:
            k = cref.fr_dot(sc, tt)
        ok = np.array_equal(got, cref.point_mul(group, gen, k))
        st = np.median(np.array(stages), axis=0)
        wall = float(np.median(walls))
        rate = n / wall / 1e3
        if base_rate is None:
            base_rate = rate
        print("%-8s %9.3f %9.3f %9.3f %9.3f %9.3f %9.1f  %.2f  %s" % (mix, wall, st[0], st[1], st[2], st[3], rate, rate / base_rate,
                                                                       "ok" if ok else "MISMATCH"))
        assert ok, mix
    bases.release()
    w.close()


if __name__ == "__main__":
    main()
