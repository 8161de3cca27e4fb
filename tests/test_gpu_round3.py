"""GPU parity of what round 3 added to the boundary (run with `pytest -m gpu` on a MI355X):

  * the reference's call sites as the patched Rust issues them (shim/patches/bellman-hip.patch, transcribed in
    csrc/groth16_callsites.cpp): with groth16/src/prover.rs patched and without -> the same proof as
    bh_groth16_prove_assignment / the oracle (groth16/src/prover.rs:217-360);
  * scalar vectors registered once and shared by several multiexps (prover.rs:267-318);
  * ONE multiexp over several contexts of one process (src/multicore.rs:21-92 is a single process), incl. density,
    skip and the EOF / identity precedence of src/multiexp.rs:295-300 across shard boundaries;
  * back-pressure: 64 concurrent 2^20-term multiexps under a capped workspace pool / a job cap all complete
    (src/multicore.rs:47-73);
  * wrapped device buffers are live views; the RCCL code path of bellman_amd/sharding.py on one GPU (world size 1).
Integer work: every limb equal, no tolerances."""

import ctypes
import os
import random
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cref  # noqa: E402
from oracle.cengine import CBls12  # noqa: E402
from oracle.pyref.generator import generate_parameters  # noqa: E402
from tests import circuits  # noqa: E402
from tests.test_gpu_groth16 import TOXIC, _chain_setup, _product_params, _same, worker  # noqa: E402,F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q = circuits.Q


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ---------------------------------------------------------------------------------------------------------------------
# the drop-in as patched
# ---------------------------------------------------------------------------------------------------------------------
def test_call_sites_mimc_322(worker):
    """C1 through the call sequences of the patched bellman (both patch levels, the weaker one in both its forms - [r5]
    "resident": the EvaluationDomain stays in HBM between its calls) == create_proof == the oracle"""
    from bellman_amd import groth16 as pg
    from oracle.pyref.prover import create_proof as oracle_create_proof

    rnd = random.Random(3220)
    cons = [rnd.randrange(Q) for _ in range(circuits.MIMC_ROUNDS)]
    xl, xr, r, s = (rnd.randrange(Q) for _ in range(4))
    p = generate_parameters(CBls12, circuits.mimc_circuit(0, 0, cons), CBls12.G1.gen, CBls12.G2.gen, **TOXIC)
    want = oracle_create_proof(CBls12, circuits.mimc_circuit(xl, xr, cons), p, r, s)
    pp = _product_params(worker, p)
    asg = pg.demo_assignment(0, circuits.MIMC_ROUNDS, 0, [xl, xr], cons)
    assert asg["a"].shape[0] == 646 and asg["aux_assignment"].shape[0] == 645
    for patched in (True, False, "resident"):
        tm = [0, 0]
        got = pg.prove_via_call_sites(pp, asg, r, s, patched, tm)
        assert _same(got, want.a, want.b, want.c), patched
    assert _same(pg.prove_assignment_arrays(pp, asg, r, s), want.a, want.b, want.c)


@pytest.mark.parametrize("rounds", [1, 61, 4093, (1 << 16) - 3])
def test_call_sites_chain_circuit(worker, rounds):
    from bellman_amd import groth16 as pg

    seed, x0, r, s = 11 + rounds, 424242, 0x1234567 + rounds, 0x7654321
    pp, vk, _ = _chain_setup(worker, rounds, seed)
    asg = pg.demo_assignment(1, rounds, seed, [x0])
    want = pg.create_proof_demo(pp, 1, rounds, seed, [x0], None, r, s)   # pinned to the oracle in test_gpu_groth16.py
    got_ref = pg.prove_assignment_arrays(pp, asg, r, s)
    assert _same(got_ref, want.a, want.b, want.c)
    for patched in (True, False, "resident"):
        got = pg.prove_via_call_sites(pp, asg, r, s, patched)
        assert _same(got, want.a, want.b, want.c), patched


def test_call_sites_error_paths(worker):
    """a short query -> UnexpectedEof through both call sequences, every job still waited on"""
    from bellman_amd import UnexpectedEof
    from bellman_amd import groth16 as pg

    rounds, seed, x0 = 200, 5, 77
    pp, vk, (h, l, a, b1, b2) = _chain_setup(worker, rounds, seed)
    short = pg.Parameters(worker, vk["alpha_g1"], vk["beta_g1"], vk["beta_g2"], vk["delta_g1"], vk["delta_g2"], h, l[:-3], a, b1, b2)
    asg = pg.demo_assignment(1, rounds, seed, [x0])
    for patched in (True, False, "resident"):
        with pytest.raises(UnexpectedEof):
            pg.prove_via_call_sites(short, asg, 5, 6, patched)
    assert worker.info()["jobs_in_flight"] == 0
    # an identity delta is reported before any multiexp is waited on (prover.rs:320-324): it wins over the EOF
    from bellman_amd import UnexpectedIdentity

    zero1 = np.zeros(12, dtype=np.uint64)
    both = pg.Parameters(worker, vk["alpha_g1"], vk["beta_g1"], vk["beta_g2"], zero1, vk["delta_g2"], h, l[:-3], a, b1, b2)
    with pytest.raises(UnexpectedIdentity):
        pg.create_proof_demo(both, 1, rounds, seed, [x0], None, 5, 6)
    for patched in (True, False, "resident"):
        with pytest.raises(UnexpectedIdentity):
            pg.prove_via_call_sites(both, asg, 5, 6, patched)
    assert worker.info()["jobs_in_flight"] == 0


# ---------------------------------------------------------------------------------------------------------------------
# scalars registered once
# ---------------------------------------------------------------------------------------------------------------------
def test_registered_scalars_shared_by_multiexps(worker):
    import bellman_amd

    n = 5000
    sc = cref.random_fr(n, 31)
    sc[3] = 0
    sc[4] = cref.ints_to_arr([1], 4)[0]
    rnd = np.random.default_rng(32)
    bits = rnd.random(n) < 0.6
    g1 = cref.gen_bases(1, n + 7, a=3, b=5)
    g2 = cref.gen_bases(2, int(bits.sum()) + 2, a=4, b=9)
    hb1, hb2 = bellman_amd.Bases(worker, 1, g1), bellman_amd.Bases(worker, 2, g2)
    for mont in (False, True):
        host = cref.fr_to_mont(sc) if mont else sc
        reg = bellman_amd.Scalars(worker, host, mont=mont)
        dt = bellman_amd.DensityTracker()
        dt.bv = bits
        jobs = [bellman_amd.multiexp_scalars(worker, hb1, bellman_amd.FullDensity(), reg, skip=7),
                bellman_amd.multiexp_scalars(worker, hb2, dt, reg, skip=2),
                bellman_amd.multiexp_scalars(worker, hb1, bellman_amd.FullDensity(), reg, skip=0, first=1000, n=3000)]
        want = [cref.multiexp(1, g1, 7, None, sc), cref.multiexp(2, g2, 2, cref.density_bitmap(bits), sc),
                cref.multiexp(1, g1, 0, None, sc[1000:4000])]
        for j, (rc, w) in zip(jobs, want):
            assert rc == 0 and np.array_equal(j.wait(), w)
        reg.release()
    # out-of-range slices are refused
    reg = bellman_amd.Scalars(worker, sc)
    with pytest.raises(Exception):
        bellman_amd.multiexp_scalars(worker, hb1, bellman_amd.FullDensity(), reg, first=n - 1, n=2)
    reg.release()


# ---------------------------------------------------------------------------------------------------------------------
# one multiexp over several contexts of this process
# ---------------------------------------------------------------------------------------------------------------------
def _split(arr, cuts):
    return [arr[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1)]


@pytest.mark.parametrize("group,log_n,shards", [(1, 20, 2), (2, 16, 2), (1, 14, 3)])
def test_sharded_multiexp_contexts_in_one_process(worker, group, log_n, shards):
    """TWO (three) contexts on device 0, each holding a contiguous shard of the bases: the fold == ONE multiexp over
    the whole vector - with FullDensity, and with a DensityTracker + skip whose dense entries cross the shard borders."""
    import bellman_amd

    n = 1 << log_n
    workers = [worker] + [bellman_amd.Worker(0) for _ in range(shards - 1)]
    try:
        bases = cref.gen_bases(group, n + 5, a=7, b=3)
        cuts = [0] + sorted(int(x) for x in np.random.default_rng(log_n).integers(1, n, shards - 1)) + [n + 5]
        hs = [bellman_amd.Bases(w, group, piece) for w, piece in zip(workers, _split(bases, cuts))]
        sc = cref.random_fr(n, 50 + log_n)
        sc[1] = 0
        threads = cref.lib().orc_max_threads()
        # (1) FullDensity, skip = 5: issued through the C entry point from this thread
        got = bellman_amd.multiexp_sharded(workers, hs, bellman_amd.FullDensity(), sc, skip=5).wait()
        rc, want = cref.multiexp(group, bases, 5, None, sc, threads=threads)
        assert rc == 0 and np.array_equal(got, want)
        # (2) density + skip
        bits = np.random.default_rng(log_n + 1).random(n) < 0.5
        dt = bellman_amd.DensityTracker()
        dt.bv = bits
        got = bellman_amd.multiexp_sharded(workers, hs, dt, sc, skip=3).wait()
        rc, want = cref.multiexp(group, bases, 3, cref.density_bitmap(bits), sc, threads=threads)
        assert rc == 0 and np.array_equal(got, want)
        # (3) the documented N-context pattern by hand: one host thread per context, each a plain multiexp over its
        # shard, folded with bh_point_add
        parts = [None] * shards
        sc_cuts = [0] + [c - 5 for c in cuts[1:-1]] + [n]          # FullDensity, skip = 5: scalar i uses base 5 + i

        def run(k):
            sk = 5 if k == 0 else 0
            parts[k] = bellman_amd.multiexp(workers[k], hs[k], bellman_amd.FullDensity(), sc[sc_cuts[k]:sc_cuts[k + 1]], skip=sk).wait()

        ts = [threading.Thread(target=run, args=(k,)) for k in range(shards)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        total = parts[0]
        for p in parts[1:]:
            total = bellman_amd.point_add(group, total, p)
        rc, want = cref.multiexp(group, bases, 5, None, sc, threads=threads)
        assert np.array_equal(total, want)
        for h in hs:
            h.release()
    finally:
        for w in workers[1:]:
            w.close()


def test_sharded_multiexp_error_semantics(worker):
    """EOF can only come from the last shard; an identity base consumed in the reference's top window by ANY shard wins
    over it (src/multiexp.rs:295-300); an identity under a zero scalar is skipped unseen - each case == the oracle's rc."""
    import bellman_amd
    from bellman_amd import UnexpectedEof, UnexpectedIdentity

    w2 = bellman_amd.Worker(0)
    try:
        n = 600
        bases = cref.gen_bases(1, n, a=2, b=11)
        sc = cref.random_fr(n, 77)

        def run(b, s, skip=0):
            hs = [bellman_amd.Bases(worker, 1, b[:250]), bellman_amd.Bases(w2, 1, b[250:])]
            rc, want = cref.multiexp(1, b, skip, None, s)
            try:
                got = bellman_amd.multiexp_sharded([worker, w2], hs, bellman_amd.FullDensity(), s, skip=skip).wait()
                assert rc == 0 and np.array_equal(got, want)
                return 0
            except UnexpectedIdentity:
                assert rc == 1
                return 1
            except UnexpectedEof:
                assert rc == 2
                return 2
            finally:
                for h in hs:
                    h.release()

        assert run(bases, sc) == 0
        assert run(bases[:-10], sc) == 2                           # EOF in the last shard
        b = bases.copy()
        b[100] = 0                                                  # identity in the FIRST shard, full-size scalar
        assert run(b, sc) == 1
        assert run(b[:-10], sc) == 1                                # ... with an EOF later: top-window identity first
        s2 = sc.copy()
        s2[100] = cref.ints_to_arr([5], 4)[0]                       # small scalar: not in the top window -> EOF wins
        assert run(b[:-10], s2) == 2
        s3 = sc.copy()
        s3[100] = 0                                                 # zero scalar skips the identity unseen
        assert run(b, s3) == 0
        assert run(bases, sc, skip=1) == 2                          # skip pushes the last entry past the end
    finally:
        w2.close()


# ---------------------------------------------------------------------------------------------------------------------
# back-pressure
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["pool_cap", "job_cap"])
def test_64_concurrent_jobs_under_back_pressure(mode):
    """64 multiexps of 2^20 G1 terms issued back to back from one thread (+ 2 more threads doing the same): with the
    workspace pool capped at 3 GiB (~4 workspaces) or 4 jobs in flight, issuing completes the oldest job inline instead
    of failing; every result equals the oracle's."""
    import bellman_amd

    w = bellman_amd.Worker(0)
    try:
        if mode == "pool_cap":
            w.set_limits(max_jobs_in_flight=1000, pool_cap_bytes=3 << 30)
        else:
            w.set_limits(max_jobs_in_flight=4)
        n = 1 << 20
        t = cref.random_fr(n, 500)
        gen = cref.g1_generator()
        dt, dout = w.alloc(n * 32), w.alloc(n * 96)
        w.upload(dt, t)
        lib = w._lib
        assert lib.bh_fixed_base_mul_dev(w.ctx, 1, _p(gen), dt, n, 0, dout, None) == 0
        w.synchronize()
        bases = bellman_amd.Bases.copy_device(w, 1, dout, n)
        scs = [cref.random_fr(n, 600 + i) for i in range(2)]
        want = [cref.point_mul(1, gen, cref.fr_dot(s, t)) for s in scs]
        regs = [bellman_amd.Scalars(w, s) for s in scs]
        peak = [0]

        def burst(count):
            jobs = []
            for i in range(count):
                jobs.append((i & 1, bellman_amd.multiexp_scalars(w, bases, bellman_amd.FullDensity(), regs[i & 1])))
                peak[0] = max(peak[0], w.info()["jobs_in_flight"])
            for which, j in jobs:
                assert np.array_equal(j.wait(), want[which])

        ts = [threading.Thread(target=burst, args=(16,)) for _ in range(2)]
        [x.start() for x in ts]
        burst(64)
        [x.join() for x in ts]
        info = w.info()
        assert info["jobs_in_flight"] == 0
        if mode == "pool_cap":
            assert info["pool_bytes_held"] <= 3 << 30
        else:
            assert peak[0] <= 4
        print(mode, "peak jobs in flight", peak[0], "pool held %.2f GiB" % (info["pool_bytes_held"] / 2**30))
        for r in regs:
            r.release()
        bases.release()
    finally:
        w.close()


# ---------------------------------------------------------------------------------------------------------------------
# wrapped buffers, context info, RCCL
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [5, 3000])
def test_wrapped_device_bases_are_a_live_view(worker, n):
    """bh_bases_wrap_dev snapshots nothing: after the caller rewrites the buffer a multiexp over the same handle uses
    the new contents - for a handful of terms (formerly answered from a host mirror) and for a table-eligible size."""
    import bellman_amd

    a, b = cref.gen_bases(1, n, a=3, b=4), cref.gen_bases(1, n, a=9, b=2)
    sc = cref.random_fr(n, 12)
    d = worker.alloc(n * 96)
    worker.upload(d, a)
    hb = bellman_amd.Bases.wrap_device(worker, 1, d, n)
    assert hb.table_info() == (0, 0, 0)
    assert np.array_equal(bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc).wait(), cref.multiexp(1, a, 0, None, sc)[1])
    worker.upload(d, b)
    assert np.array_equal(bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc).wait(), cref.multiexp(1, b, 0, None, sc)[1])
    hb.release()
    worker.free(d)


def test_fft_when_the_scratch_vector_cannot_be_had():
    """a multi-pass transform needs a scratch vector of the same size: when the pool cannot provide it the entry point
    returns BH_ERR_HIP (no partial transform, no CPU path) and the context stays usable"""
    import bellman_amd

    w = bellman_amd.Worker(0)
    try:
        data = cref.random_fr(1 << 20, 9)
        d = bellman_amd.EvaluationDomain.from_coeffs(w, data)    # 32 MiB from the pool
        w.set_limits(pool_cap_bytes=48 << 20)
        with pytest.raises(Exception):
            d.fft()
        assert np.array_equal(d.as_ref(), data)                  # untouched
        w.set_limits(pool_cap_bytes=0)
        d.fft()
        assert np.array_equal(d.into_coeffs(), cref.fft(data, cref.FFT, threads=16))
    finally:
        w.close()


def test_ctx_info_and_table_budget(worker):
    import bellman_amd

    info = worker.info()
    assert info["num_cus"] == 256 and info["hbm_bytes"] > 200 << 30
    assert info["hw_queues_requested"] == 16 and info["hw_queues_set_before_hip_init"] == 1   # the loader asked in time
    assert info["max_jobs_in_flight"] >= 8 and info["table_budget"] > 0
    w = bellman_amd.Worker(0)
    try:
        pts = cref.gen_bases(2, 1 << 12, a=5, b=6)
        hb = bellman_amd.Bases(w, 2, pts)
        c, rows, nbytes = hb.table_info()
        assert rows > 0 and w.info()["table_bytes"] == nbytes
        w.set_limits(table_budget_bytes=0)
        hb2 = bellman_amd.Bases(w, 2, pts)
        assert hb2.table_info() == (0, 0, 0)                     # over budget: registered without a table
        sc = cref.random_fr(1 << 12, 5)
        want = cref.multiexp(2, pts, 0, None, sc)[1]
        assert np.array_equal(bellman_amd.multiexp(w, hb, bellman_amd.FullDensity(), sc).wait(), want)
        assert np.array_equal(bellman_amd.multiexp(w, hb2, bellman_amd.FullDensity(), sc).wait(), want)
        w.trim()                                                  # drops the automatic table, the handle stays usable
        assert hb.table_info() == (0, 0, 0) and w.info()["table_bytes"] == 0
        assert np.array_equal(bellman_amd.multiexp(w, hb, bellman_amd.FullDensity(), sc).wait(), want)
        hb.release()
        hb2.release()
    finally:
        w.close()


def test_rccl_code_path_world_size_one():
    """The collective legs of bellman_amd/sharding.py with backend "nccl" (= RCCL) and CUDA tensors, on the one GPU of
    this box: all_gather + fold of a partial result and of a proof's 960-byte sums, and bench.py's N = 1 launch through
    torch.distributed.run with the collective forced on.  (N > 1 over xGMI is NOT measured in this repository.)"""
    code = r"""
import os, numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import bellman_amd
from bellman_amd import sharding, groth16 as pg
from oracle import cref
w = bellman_amd.Worker(0)
n = 4096
bases, sc = cref.gen_bases(1, n, a=1, b=1), cref.random_fr(n, 1)
hb = bellman_amd.Bases(w, 1, bases)
got = sharding.sharded_multiexp(w, hb, bellman_amd.FullDensity(), sc, 1, device="cuda")
assert np.array_equal(got, cref.multiexp(1, bases, 0, None, sc)[1])
sums = np.arange(120, dtype=np.uint64)
assert np.array_equal(sharding.fold_sums(sums, device="cuda"), sums)
t = torch.ones(4, device="cuda"); dist.all_reduce(t); assert float(t.sum()) == 4.0
dist.barrier(); dist.destroy_process_group(); w.close()
print("RCCL_PATH_OK", torch.cuda.nccl.version())
"""
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_PATH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29534", "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--log-n", "12", "--no-proof",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, BENCH_FORCE_COLLECTIVE="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    assert '"collective": "nccl' in line


def test_async_proofs_two_deep_equal_the_synchronous_ones(worker):
    """bh_groth16_prove_demo_async / bh_groth16_proof_wait (one caller, synthesis of proof k+1 beside the device part of
    proof k): every proof equals create_proof's for the same circuit, r, s - with the constraints evaluated on the host
    (as in the reference) and on the device; an error (short query) surfaces at the wait."""
    from bellman_amd import UnexpectedEof
    from bellman_amd import groth16 as pg

    rounds, seed = 4093, 17
    pp, vk, (h, l, a, b1, b2) = _chain_setup(worker, rounds, seed)
    r1cs = pg.R1CS.from_demo(worker, 1, rounds, seed)
    want = [pg.create_proof_demo(pp, 1, rounds, seed, [1000 + i], None, 50 + i, 60 + i) for i in range(5)]
    for rc in (r1cs, None):
        waits, got = [], []
        for i in range(5):
            waits.append(pg.create_proof_demo_async(pp, rc, 1, rounds, seed, [1000 + i], None, 50 + i, 60 + i))
            if len(waits) == 2:
                got.append(waits.pop(0)())
        got += [w() for w in waits]
        for g, w in zip(got, want):
            assert _same(g, w.a, w.b, w.c)
    short = pg.Parameters(worker, vk["alpha_g1"], vk["beta_g1"], vk["beta_g2"], vk["delta_g1"], vk["delta_g2"], h[:-2], l, a, b1, b2)
    wait = pg.create_proof_demo_async(short, r1cs, 1, rounds, seed, [1], None, 2, 3)
    with pytest.raises(UnexpectedEof):
        wait()
    assert worker.info()["jobs_in_flight"] == 0
    r1cs.release()


def test_async_proofs_from_an_assignment_and_from_a_witness(worker):
    """[r4] bh_groth16_prove_assignment_async / bh_groth16_prove_witness_async + bh_groth16_proof_wait: what a C or Rust
    host that synthesised by itself calls to keep two proofs in flight (prover.rs:182-215 on its side, :217-360 here).
    Every proof equals the synchronous entry point's and create_proof's for the same circuit, r, s; the witness call has
    copied its inputs when it returns (the buffers are overwritten right away); an error surfaces at the wait."""
    from bellman_amd import UnexpectedEof
    from bellman_amd import groth16 as pg

    rounds, seed = 4093, 23
    pp, vk, (h, l, a, b1, b2) = _chain_setup(worker, rounds, seed)
    r1cs = pg.R1CS.from_demo(worker, 1, rounds, seed)
    want = [pg.create_proof_demo(pp, 1, rounds, seed, [2000 + i], None, 70 + i, 80 + i) for i in range(4)]
    asgs = [pg.demo_assignment(1, rounds, seed, [2000 + i]) for i in range(4)]
    # the assignment path, two deep
    waits, got = [], []
    for i in range(4):
        waits.append(pg.prove_assignment_arrays_async(pp, asgs[i], 70 + i, 80 + i))
        if len(waits) == 2:
            got.append(waits.pop(0)())
    got += [w() for w in waits]
    for g, w, asg, i in zip(got, want, asgs, range(4)):
        assert _same(g, w.a, w.b, w.c)
        sync = pg.prove_assignment_arrays(pp, asg, 70 + i, 80 + i)
        assert _same(sync, w.a, w.b, w.c)
    # the witness path (constraints evaluated on the device), inputs overwritten as soon as the call returns
    waits = []
    for i in range(4):
        ia, aa = asgs[i]["input_assignment"].copy(), asgs[i]["aux_assignment"].copy()
        waits.append(pg.prove_witness_async(r1cs, pp, ia, aa, 70 + i, 80 + i))
        ia[:] = 0
        aa[:] = 0
    for wt, w in zip(waits, want):
        assert _same(wt(), w.a, w.b, w.c)
    short = pg.Parameters(worker, vk["alpha_g1"], vk["beta_g1"], vk["beta_g2"], vk["delta_g1"], vk["delta_g2"], h[:-2], l, a, b1, b2)
    wait = pg.prove_assignment_arrays_async(short, asgs[0], 2, 3)
    with pytest.raises(UnexpectedEof):
        wait()
    assert worker.info()["jobs_in_flight"] == 0
    r1cs.release()


def test_held_jobs_two_phase_issue(worker):
    """BH_MSM_HOLD / bh_msm_start: jobs issued held (digit + sort stage only), started in another order or not at all
    (the wait starts them) - the results are those of the plain call, whatever the plan (classic, window table, G2)."""
    import bellman_amd
    from bellman_amd.multiexp import HOLD

    cases = []
    for i, (g, n) in enumerate([(1, 1 << 17), (2, 1 << 13), (1, 3000), (2, 1 << 16), (1, 5)]):
        bases = cref.gen_bases(g, n, a=i + 3, b=7)
        sc = cref.random_fr(n, 4000 + i)
        cases.append((bellman_amd.Bases(worker, g, bases), sc, cref.multiexp(g, bases, 0, None, sc)[1]))
    for order in ("reverse", "none", "forward"):
        jobs = [bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc, flags=HOLD) for hb, sc, _ in cases]
        if order == "reverse":
            for j in reversed(jobs):
                j.start()
        elif order == "forward":
            for j in jobs:
                j.start()
                j.start()   # idempotent
        for j, (_, _, want) in zip(jobs, cases):
            assert np.array_equal(j.wait(), want), order
    assert worker.info()["jobs_in_flight"] == 0


@pytest.mark.parametrize("n", [9, 64, 322, 645, 1023])
def test_small_multiexp_single_launch_path(worker, n):
    """the one-launch path of small multiexps over a window table (msm_small_fill_kernel: MiMC-sized jobs) against the
    oracle AND against the full pipeline (BH_MSM_NO_SMALL_PATH), with: a density map + skip, Montgomery scalars, the
    scalars 0 / 1 / q-1, a bucket far fuller than its list (every other scalar equal: the table-scan fallback), EOF and
    identity errors with the reference's precedence."""
    import bellman_amd
    from bellman_amd import UnexpectedEof, UnexpectedIdentity
    from bellman_amd.multiexp import NO_SMALL_PATH

    rnd = np.random.default_rng(n)
    bases = cref.gen_bases(1, n + 3, a=n, b=3)
    hb = bellman_amd.Bases(worker, 1, bases)
    assert hb.table_info()[1] > 0          # registered with its window table: the fused path applies
    sc = cref.random_fr(n, 9000 + n)
    sc[0] = 0
    sc[1] = cref.ints_to_arr([1], 4)[0]
    sc[2] = cref.ints_to_arr([cref.Q - 1], 4)[0]
    sc[3::2] = cref.ints_to_arr([0x123456789ABCDEF], 4)[0]      # ~n/2 equal scalars: their buckets overflow the lists
    bits = rnd.random(n) < 0.7
    dt = bellman_amd.DensityTracker()
    dt.bv = bits
    for dens, dens_c, skip in ((bellman_amd.FullDensity(), None, 3), (dt, cref.density_bitmap(bits), 2)):
        rc, want = cref.multiexp(1, bases, skip, dens_c, sc)
        assert rc == 0
        got = bellman_amd.multiexp(worker, hb, dens, sc, skip=skip).wait()
        full = bellman_amd.multiexp(worker, hb, dens, sc, skip=skip, flags=NO_SMALL_PATH).wait()
        mont = bellman_amd.multiexp(worker, hb, dens, cref.fr_to_mont(sc), skip=skip, mont=True).wait()
        assert np.array_equal(got, want) and np.array_equal(full, want) and np.array_equal(mont, want)
    # errors: identity under a non-zero scalar; EOF; both (top-window precedence); identity under a zero scalar is unseen
    b2 = bases.copy()
    b2[5] = 0
    hb2 = bellman_amd.Bases(worker, 1, b2)
    short = bellman_amd.Bases(worker, 1, b2[: n - 1])
    for handle, arr, s in ((hb2, b2, sc), (short, b2[: n - 1], sc), (short, bases[: n - 1], sc)):
        rc, want = cref.multiexp(1, arr, 0, None, s)
        try:
            got = bellman_amd.multiexp(worker, handle, bellman_amd.FullDensity(), s).wait()
            assert rc == 0 and np.array_equal(got, want)
        except UnexpectedIdentity:
            assert rc == 1
        except UnexpectedEof:
            assert rc == 2
    s0 = sc.copy()
    s0[5] = 0
    rc, want = cref.multiexp(1, b2, 0, None, s0)
    assert rc == 0 and np.array_equal(bellman_amd.multiexp(worker, hb2, bellman_amd.FullDensity(), s0).wait(), want)
    for h in (hb, hb2, short):
        h.release()


@pytest.mark.parametrize("group", [1, 2])
def test_reductions_with_equal_and_opposite_partial_sums(worker, group):
    """every base the SAME point and small scalars k, 2^c - k: different buckets then hold equal points (P, P, ...) and
    opposite points (P, -P), so the row / column / bit sums of the reduction add equal points (the doubling case - on lane
    pairs the one-lane fallback) and opposite points (-> identity) at every tree level.  Window-table and classic plans."""
    import bellman_amd
    from bellman_amd.multiexp import NO_TABLE, NO_SMALL_PATH

    gen = cref.g1_generator() if group == 1 else cref.g2_generator()
    pt = cref.point_mul(group, gen, 0xC0FFEE)
    for n, c in ((200, 13), (1500, 13), (40000, 16)):
        bases = np.repeat(pt[None, :], n, axis=0)
        ks = list(range(1, n // 2 + 1)) + [(1 << c) - k for k in range(1, n - n // 2 + 1)]
        sc = cref.ints_to_arr(ks, 4)
        rc, want = cref.multiexp(group, bases, 0, None, sc)
        assert rc == 0
        hb = bellman_amd.Bases(worker, group, bases)
        for flags in (0, NO_SMALL_PATH, NO_TABLE | NO_SMALL_PATH):
            got = bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc, flags=flags).wait()
            assert np.array_equal(got, want), (n, flags)
        hb.release()


@pytest.mark.parametrize("group", [1, 2])
def test_bucket_runs_of_every_length_class(worker, group):
    """[r4] Bucket runs that straddle chunk boundaries are folded by the owner lane of the merge kernel (up to four
    further chunks), by the wavefront-parallel merge (queued runs) or by the workgroup-parallel one (csrc/msm_ec.cuh).
    Scalars drawn from a small set of values give every window a few dozen buckets whose runs are 1 ... 60 chunks long,
    so with chunks of 8 / 16 / 32 entries every route and every kernel bundle (one lane, lane pairs, lane triples per G2
    point) sees runs of every length class.  Against the restated multiexp (src/multiexp.rs:210-301).  (Written for the
    round-4 experiment that folded runs inside the accumulation workgroup - measured slower, profiles/archive/r4_call5_fold_ab.txt,
    and removed - and kept because no other test forces these shapes.)"""
    import random

    import bellman_amd
    from bellman_amd.multiexp import NO_TABLE, NO_SMALL_PATH

    rnd = random.Random(4041)
    n = 6000
    bases = cref.gen_bases(group, n, a=11, b=29)
    hb = bellman_amd.Bases(worker, group, bases)
    for distinct in (3, 40, 700):
        vals = [rnd.randrange(cref.Q) for _ in range(distinct)]
        sc = cref.ints_to_arr([vals[min(int(rnd.expovariate(4.0 / distinct)), distinct - 1)] for _ in range(n)], 4)
        rc, want = cref.multiexp(group, bases, 0, None, sc)
        assert rc == 0
        bundles = (0,) if group == 1 else (16, 32, 256, 256 | 16)
        for chunk in (8, 16, 32):
            for fl in bundles:
                for c in (13, 16):
                    got = bellman_amd.multiexp(worker, hb, bellman_amd.FullDensity(), sc, window_bits=c, chunk=chunk,
                                               flags=fl | NO_TABLE | NO_SMALL_PATH).wait()
                    assert np.array_equal(got, want), (distinct, chunk, fl, c)
    hb.release()
