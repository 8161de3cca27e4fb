"""Isolated workloads for rocprofv3 / hand timing on the MI355X (one workload per process so the kernel
statistics of a run belong to exactly one configuration).

  python tools/profile_suite.py msm <group 1|2> <log_n> [iters] [c] [K]   device-resident scalars, default plan
  python tools/profile_suite.py fft <log_n> [iters]                       fft, ifft, coset_fft, icoset_fft
  python tools/profile_suite.py mimc [iters]                              create_proof on MiMC-322 (config C1)
  python tools/profile_suite.py sizes <group> <lo> <hi>                   per-stage device ms for 2^lo..2^hi
  python tools/profile_suite.py proof [log_n] [iters] [threads]           create_proof, chain circuit, R1CS resident

Prints host-side timings; run it under `rocprofv3 --kernel-trace --stats` for the per-kernel view."""
import ctypes
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bellman_amd  # noqa: E402
from bellman_amd import _lib  # noqa: E402
from bench import G1_GEN_MONT, G2_GEN_MONT, splitmix_scalars  # noqa: E402


SUITE_FLAGS = int(os.environ.get("BH_SUITE_FLAGS", "0"), 0)   # bh_msm_opts.flags for `msm` / `sizes` (A/B of kernel variants)


def make_bases(w, lib, group, n):
    words = 12 if group == 1 else 24
    t = splitmix_scalars(n, 1)
    dt, dout = w.alloc(n * 32), w.alloc(n * 8 * words)
    w.upload(dt, t)
    gen = G1_GEN_MONT if group == 1 else G2_GEN_MONT
    assert lib.bh_fixed_base_mul_dev(w.ctx, group, gen.ctypes.data_as(ctypes.c_void_p), dt, n, 0, dout, None) == 0
    w.synchronize()
    w.free(dt)
    return dout


def run_msm(args):
    group, log_n = int(args[0]), int(args[1])
    iters = int(args[2]) if len(args) > 2 else 10
    c = int(args[3]) if len(args) > 3 else 0
    k = int(args[4]) if len(args) > 4 else 0
    lib = _lib.load()
    w = bellman_amd.Worker(0)
    n = 1 << log_n
    dout = make_bases(w, lib, group, n)
    bases = bellman_amd.Bases.copy_device(w, group, dout, n)
    s = splitmix_scalars(n, 2)
    ds = w.alloc(n * 32)
    w.upload(ds, s)
    walls, best = [], None
    for it in range(iters + 2):
        t0 = time.perf_counter()
        r, ms = bellman_amd.multiexp(w, bases, bellman_amd.FullDensity(), None, scalars_dev=ds, n=n, timed=True,
                                     window_bits=c, chunk=k, flags=SUITE_FLAGS).wait()
        wall = (time.perf_counter() - t0) * 1e3
        if it >= 2:
            walls.append(wall)
            if best is None or ms[0] < best[0]:
                best = ms
    walls.sort()
    print("G%d MSM 2^%d c=%d K=%d: wall median %.3f ms (min %.3f); device best total %.3f = sort %.3f + accumulate %.3f + reduce %.3f" %
          (group, log_n, c, k, walls[len(walls) // 2], walls[0], *best), flush=True)


def run_sizes(args):
    group, lo, hi = int(args[0]), int(args[1]), int(args[2])
    lib = _lib.load()
    w = bellman_amd.Worker(0)
    nmax = 1 << hi
    dout = make_bases(w, lib, group, nmax)
    s = splitmix_scalars(nmax, 2)
    ds = w.alloc(nmax * 32)
    w.upload(ds, s)
    for log_n in range(lo, hi + 1):
        n = 1 << log_n
        bases = bellman_amd.Bases.copy_device(w, group, dout, n)
        best, walls = None, []
        for it in range(7):
            t0 = time.perf_counter()
            r, ms = bellman_amd.multiexp(w, bases, bellman_amd.FullDensity(), None, scalars_dev=ds, n=n, timed=True,
                                         flags=SUITE_FLAGS).wait()
            walls.append((time.perf_counter() - t0) * 1e3)
            if it and (best is None or ms[0] < best[0]):
                best = ms
        walls = sorted(walls[1:])
        print("G%d log_n=%d  wall median %.3f ms  device total %.3f ms  sort %.3f  accumulate %.3f  reduce %.3f" %
              (group, log_n, walls[len(walls) // 2], *best), flush=True)


def run_sweep(args):
    """sweep <group> <log_n> <c> <K list> <flags list> [reps]: every (K, flags) combination in ONE process, `reps` times
    round-robin, so that variants are compared on the same box under the same conditions."""
    group, log_n, c = int(args[0]), int(args[1]), int(args[2])
    ks = [int(x) for x in args[3].split(",")]
    fls = [int(x, 0) for x in args[4].split(",")]
    reps = int(args[5]) if len(args) > 5 else 2
    lib = _lib.load()
    w = bellman_amd.Worker(0)
    n = 1 << log_n
    dout = make_bases(w, lib, group, n)
    bases = bellman_amd.Bases.copy_device(w, group, dout, n)
    s = splitmix_scalars(n, 2)
    ds = w.alloc(n * 32)
    w.upload(ds, s)
    res = {}
    for rep in range(reps):
        for k in ks:
            for fl in fls:
                best, walls = None, []
                for it in range(6):
                    t0 = time.perf_counter()
                    r, ms = bellman_amd.multiexp(w, bases, bellman_amd.FullDensity(), None, scalars_dev=ds, n=n, timed=True,
                                                 window_bits=c, chunk=k, flags=fl).wait()
                    walls.append((time.perf_counter() - t0) * 1e3)
                    if it and (best is None or ms[0] < best[0]):
                        best = ms
                res.setdefault((k, fl), []).append((sorted(walls[1:])[2], best))
    for (k, fl), v in res.items():
        for wall, best in v:
            print("G%d 2^%d c=%d K=%3d flags=%4d: wall median %.3f ms; device best total %.3f = sort %.3f + accumulate %.3f + reduce %.3f" %
                  (group, log_n, c, k, fl, wall, *best), flush=True)


def run_tsweep(args):
    """tsweep <group> <lo> <hi> <table c list, 0 = no table (classic plan)>: per-stage device ms of 2^lo..2^hi with the
    window table rebuilt for each c (bh_bases_precompute), to re-check the plan boundaries of msm_stages.hip."""
    group, lo, hi = int(args[0]), int(args[1]), int(args[2])
    cs = [int(x) for x in args[3].split(",")]
    ks = [int(x) for x in args[4].split(",")] if len(args) > 4 else [0]   # chunk lengths K (0 = the plan's)
    from bellman_amd.multiexp import NO_TABLE
    lib = _lib.load()
    w = bellman_amd.Worker(0)
    nmax = 1 << hi
    dout = make_bases(w, lib, group, nmax)
    s = splitmix_scalars(nmax, 2)
    ds = w.alloc(nmax * 32)
    w.upload(ds, s)
    for log_n in range(lo, hi + 1):
        n = 1 << log_n
        for c in cs:
            bases = bellman_amd.Bases.copy_device(w, group, dout, n)
            flags = (NO_TABLE if c == 0 else 0) | SUITE_FLAGS
            if c:
                bases.precompute(c)
            for k in ks:
                best, walls = None, []
                for it in range(7):
                    t0 = time.perf_counter()
                    r, ms = bellman_amd.multiexp(w, bases, bellman_amd.FullDensity(), None, scalars_dev=ds, n=n, timed=True, flags=flags, chunk=k).wait()
                    walls.append((time.perf_counter() - t0) * 1e3)
                    if it and (best is None or ms[0] < best[0]):
                        best = ms
                walls = sorted(walls[1:])
                print("G%d log_n=%d table c=%2d%s  wall median %.3f ms  device total %.3f ms  sort %.3f  accumulate %.3f  reduce %.3f" %
                      (group, log_n, c, " K=%3d" % k if k else "", walls[len(walls) // 2], *best), flush=True)
            bases.release()


def run_fft(args):
    log_n = int(args[0])
    iters = int(args[1]) if len(args) > 1 else 10
    lib = _lib.load()
    w = bellman_amd.Worker(0)
    n = 1 << log_n
    data = splitmix_scalars(n, 3)
    d = w.alloc(n * 32)
    w.upload(d, data)
    for i in range(40):   # untimed: the clock of an idle device ramps up over the first ~25 ms of load (r5_call2_fft_wave_local.txt)
        assert lib.bh_fft_fr_dev(w.ctx, d, log_n, i & 3, None) == 0
    w.synchronize()
    for mode, name in enumerate(("fft", "ifft", "coset_fft", "icoset_fft")):
        ts = []
        for it in range(iters + 2):
            w.synchronize()
            t0 = time.perf_counter()
            assert lib.bh_fft_fr_dev(w.ctx, d, log_n, mode, None) == 0
            w.synchronize()
            if it >= 2:
                ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        t = ts[len(ts) // 2]
        print("log_n=%d %-10s median %.3f ms (min %.3f)  algorithmic %.1f GB/s (64 B/elem)  %.1f Gbutterfly/s" %
              (log_n, name, t, ts[0], n * 64 / t / 1e6, n / 2 * log_n / t / 1e6), flush=True)


def run_mimc(args):
    iters = int(args[0]) if args else 20
    from bellman_amd import groth16 as pg
    from tests import circuits
    from oracle.pyref import bls12_381 as bls

    w = bellman_amd.Worker(0)
    rnd = random.Random(322)
    cons = [rnd.randrange(bls.Q) for _ in range(circuits.MIMC_ROUNDS)]
    xl, xr = rnd.randrange(bls.Q), rnd.randrange(bls.Q)
    r, s = rnd.randrange(bls.Q), rnd.randrange(bls.Q)
    r1cs = pg.R1CS.from_demo(w, 0, circuits.MIMC_ROUNDS, 0, cons)
    cons_mont = pg.fr_to_mont_array(cons)   # the round constants are fixed: converted once, not per proof
    params = pg.Parameters.generate(w, r1cs, G1_GEN_MONT, G2_GEN_MONT, alpha=48577, beta=22580, gamma=53332, delta=5481, tau=3673)
    walls, tms = [], []
    for it in range(iters + 3):
        tm = [0, 0, 0, 0]
        t0 = time.perf_counter()
        pg.create_proof_demo(params, 0, circuits.MIMC_ROUNDS, 0, [xl + it, xr], cons_mont, r + it, s, tm)   # full-size r, s: the host's blinding multiplications are part of a proof
        wall = (time.perf_counter() - t0) * 1e3
        if it >= 3:
            walls.append(wall)
            tms.append(tm)
    walls.sort()
    med = [sorted(t[i] for t in tms)[len(tms) // 2] for i in range(4)]
    print("MiMC-322 create_proof: wall median %.3f ms (min %.3f); host ms [synthesis %.3f, h %.3f, msm %.3f, total %.3f]" %
          (walls[len(walls) // 2], walls[0], *med), flush=True)


def run_proof(args):
    """create_proof on the 2^log_n chain circuit with the constraint matrices resident in HBM: single-stream latency
    (per-phase host ms) and throughput with `threads` host threads on one context."""
    log_n = int(args[0]) if args else 20
    iters = int(args[1]) if len(args) > 1 else 5
    threads = int(args[2]) if len(args) > 2 else 12
    from concurrent.futures import ThreadPoolExecutor
    from bellman_amd import groth16 as pg

    w = bellman_amd.Worker(0)
    rounds, seed = (1 << log_n) - 3, 2020
    r1cs = pg.R1CS.from_demo(w, 1, rounds, seed)
    params = pg.Parameters.generate(w, r1cs, G1_GEN_MONT, G2_GEN_MONT, alpha=48577, beta=22580, gamma=53332, delta=5481, tau=3673)
    walls, tms = [], []
    for it in range(iters + 2):
        tm = [0, 0, 0, 0]
        t0 = time.perf_counter()
        pg.create_proof_demo_r1cs(params, r1cs, 1, rounds, seed, [987654321 + it], None, 0xABCDEF0123 + it, 0x123456789AB, tm)
        wall = (time.perf_counter() - t0) * 1e3
        if it >= 2:
            walls.append(wall)
            tms.append(tm)
    walls.sort()
    med = [sorted(t[i] for t in tms)[len(tms) // 2] for i in range(4)]

    def one(i):
        pg.create_proof_demo_r1cs(params, r1cs, 1, rounds, seed, [1234567 + i], None, 0x55AA + i, 0x77, None)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(one, range(threads * 3)))
    conc = threads * 3 / (time.perf_counter() - t0)
    print("create_proof 2^%d (R1CS resident): wall median %.2f ms (min %.2f); host ms [witness %.2f, issue+h %.2f, "
          "h multiexp + waits %.2f, total %.2f]; %d threads: %.2f proofs/s" %
          (log_n, walls[len(walls) // 2], walls[0], *med, threads, conc), flush=True)


if __name__ == "__main__":
    {"msm": run_msm, "sweep": run_sweep, "tsweep": run_tsweep, "fft": run_fft, "mimc": run_mimc, "sizes": run_sizes, "proof": run_proof}[sys.argv[1]](sys.argv[2:])
