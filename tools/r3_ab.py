"""Round-3 A/B measurements on the MI355X (per-stage device ms, median of `iters` runs):
  python tools/r3_ab.py g1flags            G1 2^20, default registers accumulator vs BH_MSM_ACC_LDS (3 wavefronts per SIMD)
  python tools/r3_ab.py halfdense <group>  2^20 scalars of which every other one is dense (the b_g1 / b_g2 shape of a
                                           2^20-constraint proof) next to the same 2^19 terms handed over densely"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bellman_amd  # noqa: E402
from bellman_amd import _lib  # noqa: E402
from bench import splitmix_scalars  # noqa: E402
from tools.profile_suite import make_bases  # noqa: E402


def med_stage(w, bases, dens, sc_dev, n, flags=0, iters=8, dens_dev=None, skip=0, chunk=0):
    out = []
    for it in range(iters + 2):
        _, ms = bellman_amd.multiexp(w, bases, dens, None, scalars_dev=sc_dev, n=n, timed=True, flags=flags, density_dev=dens_dev,
                                     skip=skip, chunk=chunk).wait()
        if it >= 2:
            out.append(ms)
    return [round(float(x), 3) for x in np.median(np.array(out), axis=0)]


def main():
    lib = _lib.load()
    w = bellman_amd.Worker(0)
    what = sys.argv[1]
    if what == "g1flags":
        n = 1 << 20
        dout = make_bases(w, lib, 1, n)
        bases = bellman_amd.Bases.copy_device(w, 1, dout, n)
        ds = w.alloc(n * 32)
        w.upload(ds, splitmix_scalars(n, 2))
        for name, fl in (("registers (default)", 0), ("ACC_LDS", 2), ("registers", 1)):
            print("G1 2^20 %-20s [pipeline, sort, accumulate, reduce] ms = %s" % (name, med_stage(w, bases, bellman_amd.FullDensity(), ds, n, fl)), flush=True)
    elif what == "g1table":
        # a window table for a 2^20-point G1 query (c = 16: 1.5 GB, one bucket set of 2^15 buckets, 15-step host tail)
        n = 1 << 20
        dout = make_bases(w, lib, 1, n)
        bases = bellman_amd.Bases.copy_device(w, 1, dout, n)
        ds = w.alloc(n * 32)
        w.upload(ds, splitmix_scalars(n, 2))
        import time
        for c in (0, 16, 20):
            if c:
                bases.precompute(c)
            for chunk in ((0,) if not c else (0, 64, 128, 256)):
                walls = []
                for it in range(8):
                    t0 = time.perf_counter()
                    bellman_amd.multiexp(w, bases, bellman_amd.FullDensity(), None, scalars_dev=ds, n=n, chunk=chunk).wait()
                    if it >= 2:
                        walls.append((time.perf_counter() - t0) * 1e3)
                print("G1 2^20 table c=%d K=%d: wall median %.3f ms; stages %s" %
                      (c, chunk, float(np.median(walls)), med_stage(w, bases, bellman_amd.FullDensity(), ds, n, chunk=chunk)), flush=True)
    elif what == "halfdense":
        group = int(sys.argv[2])
        n = 1 << 20
        nb = n // 2
        dout = make_bases(w, lib, group, nb + 2)
        bases = bellman_amd.Bases.copy_device(w, group, dout, nb + 2)
        sc = splitmix_scalars(n, 3)
        ds = w.alloc(n * 32)
        w.upload(ds, sc)
        bits = (np.arange(n) % 2 == 0)
        dt = bellman_amd.DensityTracker()
        dt.bv = bits
        words = dt.words()
        dd = w.alloc(words.nbytes)
        w.upload(dd, words)
        print("G%d 2^20 scalars, 2^19 dense [pipeline, sort, accumulate, reduce] ms = %s" %
              (group, med_stage(w, bases, dt, ds, n, dens_dev=dd, skip=1)), flush=True)
        ds2 = w.alloc(nb * 32)
        w.upload(ds2, np.ascontiguousarray(sc[bits]))
        print("G%d 2^19 scalars, all dense    [pipeline, sort, accumulate, reduce] ms = %s" %
              (group, med_stage(w, bases, bellman_amd.FullDensity(), ds2, nb, skip=1)), flush=True)
    w.close()


main()
