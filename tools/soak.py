"""Soak: many MSMs / FFTs / proofs of varying sizes through one context; reports device-memory drift."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bellman_amd
from bellman_amd import _lib
from bench import splitmix_scalars, G1_GEN_MONT

def free_mb():
    return torch.cuda.mem_get_info()[0] / 2**20

def main():
    lib = _lib.load()
    w = bellman_amd.Worker(0)
    n = 1 << 18
    t = splitmix_scalars(n, 1)
    dt, dout = w.alloc(n * 32), w.alloc(n * 96)
    w.upload(dt, t)
    assert lib.bh_fixed_base_mul_dev(w.ctx, 1, G1_GEN_MONT.ctypes.data_as(ctypes.c_void_p), dt, n, 0, dout, None) == 0
    w.synchronize()
    bases = bellman_amd.Bases.wrap_device(w, 1, dout, n)
    rnd = np.random.default_rng(5)
    ref = {}
    marks = []
    t0 = time.time()
    for it in range(300):
        m = int(rnd.choice([1, 77, 1000, 4096, 30000, 1 << 16, 1 << 18]))
        sc = splitmix_scalars(m, 100 + m)
        jobs = [bellman_amd.multiexp(w, bases, bellman_amd.FullDensity(), sc) for _ in range(int(rnd.integers(1, 4)))]
        for j in jobs:
            r = j.wait()
            assert np.array_equal(ref.setdefault(m, r), r)
        if it % 10 == 0:
            lg = int(rnd.integers(4, 19))
            d = bellman_amd.EvaluationDomain.from_coeffs(w, splitmix_scalars(1 << lg, it))
            d.coset_fft(); d.icoset_fft(); d.into_coeffs()
        if it % 50 == 0:
            w.synchronize()
            marks.append(round(free_mb()))
    w.synchronize()
    marks.append(round(free_mb()))
    print("free MiB over time:", marks, "elapsed %.1fs" % (time.time() - t0))
    assert marks[-1] >= marks[1] - 64, "device memory keeps shrinking"
    print("soak ok (multiexp / fft)")
    soak_fft_cache(w)
    soak_proofs(w)


def soak_fft_cache(w):
    """[r5] the FFT table cache under pressure and from four host threads at once: a 96 MiB budget (the one-level set of
    2^19 points is 64 MiB, of 2^20 points 128 MiB: two-level), random sizes 2^12 ... 2^21 - every table request may have to
    evict another size's tables while other threads are between their table lookup and their launches (fft.hip fft_evict:
    exclusive lock + device synchronise).  Every transform is checked by its round trip; the cache never exceeds its budget."""
    from concurrent.futures import ThreadPoolExecutor

    budget = 96 << 20
    w.set_limits(fft_table_budget_bytes=budget)
    w.trim() if hasattr(w, "trim") else None
    t0 = time.time()
    peak = [0]

    def worker_thread(tid):
        rnd = np.random.default_rng(900 + tid)
        for it in range(40):
            lg = int(rnd.integers(12, 22))
            data = splitmix_scalars(1 << lg, 1000 * tid + it)
            d = bellman_amd.EvaluationDomain.from_coeffs(w, data)
            if it & 1:
                d.coset_fft(); d.icoset_fft()
            else:
                d.fft(); d.ifft()
            assert np.array_equal(d.into_coeffs(), data), (tid, it, lg)
            peak[0] = max(peak[0], w.info()["fft_table_bytes"])
        return tid

    with ThreadPoolExecutor(max_workers=4) as ex:
        assert sorted(ex.map(worker_thread, range(4))) == [0, 1, 2, 3]
    assert peak[0] <= budget, (peak[0], budget)
    print("soak ok (fft table cache: 160 transforms from 4 threads under a %d MiB budget, peak %d MiB, %.1fs)" %
          (budget >> 20, peak[0] >> 20, time.time() - t0))
    w.set_limits(fft_table_budget_bytes=w.info()["hbm_bytes"] // 8)


def soak_proofs(w):
    """create / prove / release cycles of every handle type of the proof path, from four host threads"""
    from concurrent.futures import ThreadPoolExecutor
    from bellman_amd import groth16 as pg
    from bench import G2_GEN_MONT

    rounds, seed = (1 << 14) - 3, 9
    marks = []
    t0 = time.time()
    ref = {}
    for cycle in range(int(os.environ.get("SOAK_CYCLES", "12"))):
        r1cs = pg.R1CS.from_demo(w, 1, rounds, seed)
        params = pg.Parameters.generate(w, r1cs, G1_GEN_MONT, G2_GEN_MONT, 48577, 22580, 53332, 5481, 3673)
        if cycle % 3 == 0:   # through the serialized form as well
            blob = params.write()
            params.release()
            params = pg.Parameters.read(w, blob, cycle % 2 == 0)

        def one(i):
            if i % 3 == 2:   # a proof assembled from three slices
                tot = None
                for part in range(3):
                    sm = pg.prove_demo_part(params, r1cs, 1, rounds, seed, [1000 + i % 5], None, part, 3)
                    tot = sm if tot is None else pg.sums_add(tot, sm)
                p = pg.assemble(params, tot, 77, 88)
            elif i % 3 == 1:
                p = pg.create_proof_demo_r1cs(params, r1cs, 1, rounds, seed, [1000 + i % 5], None, 77, 88)
            else:
                p = pg.create_proof_demo(params, 1, rounds, seed, [1000 + i % 5], None, 77, 88)
            return i % 5, p.a.tobytes() + p.b.tobytes() + p.c.tobytes()

        with ThreadPoolExecutor(max_workers=4) as ex:
            for k, proof in ex.map(one, range(20)):
                assert ref.setdefault(k, proof) == proof, "proof changed between paths / cycles"
        r1cs.release()
        params.release()
        w.synchronize()
        marks.append(round(free_mb()))
    print("free MiB after each proof cycle:", marks, "elapsed %.1fs" % (time.time() - t0))
    # the pools reach their steady state after a few cycles (how many depends on how many jobs really overlap)
    assert marks[-1] >= marks[len(marks) // 2] - 16, "device memory keeps shrinking"
    print("soak ok (proof path)")
    before = free_mb()
    w.trim()
    print("trim: free MiB %d -> %d" % (before, free_mb()))
    assert free_mb() >= before

main()
