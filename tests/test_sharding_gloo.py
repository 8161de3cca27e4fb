"""world_size-2 gloo test (CPU) of the multi-GPU MSM path: contiguous base shards + ONE all-gather
of the per-rank partial points + local fold with the product's host group law.  The per-rank MSM
itself needs a GPU, so here each rank's partial comes from the oracle; what is under test is
bellman_amd.sharding (split, exchange, fold) exactly as bench.py --gpus N uses it."""

import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, group, n, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from bellman_amd import sharding
    from oracle import cref

    dist.init_process_group("gloo", rank=rank, world_size=world)
    bases = cref.gen_bases(group, n, a=3, b=7)
    sc = cref.random_fr(n, 5)
    lo, hi = sharding.shard_bounds(n, world, rank)
    rc, part = cref.multiexp(group, bases[lo:hi], 0, None, sc[lo:hi])  # stand-in for the GPU shard MSM
    assert rc == 0
    total = sharding.fold_partials(part, group)
    rc, want = cref.multiexp(group, bases, 0, None, sc)
    q.put((rank, bool(np.array_equal(total, want))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("group", [1, 2])
def test_sharded_msm_fold_world2(group):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world, n = 2, 301  # odd n: uneven shards
    procs = [ctx.Process(target=_worker, args=(r, world, port, group, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]


def test_shard_bounds_cover_everything():
    from bellman_amd.sharding import shard_bounds

    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _sums_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from bellman_amd import sharding
    from oracle import cref

    dist.init_process_group("gloo", rank=rank, world_size=world)

    def record(seed):   # a_in, a_aux, b1_in, b1_aux (G1) | b2_in, b2_aux (G2) | h, l (G1); one slot is the identity
        g1 = cref.gen_bases(1, 6, a=seed, b=3)
        g2 = cref.gen_bases(2, 2, a=seed + 1, b=5)
        if seed % 2:
            g1[1] = 0
        return np.concatenate([g1[:4].reshape(-1), g2.reshape(-1), g1[4:].reshape(-1)])

    total = sharding.fold_sums(record(10 + rank))
    recs = [record(10 + k) for k in range(world)]
    want = recs[0].copy()
    for other in recs[1:]:
        for lo, hi, group in ((0, 12, 1), (12, 24, 1), (24, 36, 1), (36, 48, 1), (48, 72, 2), (72, 96, 2), (96, 108, 1), (108, 120, 1)):
            want[lo:hi] = cref.point_add(group, want[lo:hi], other[lo:hi])
    q.put((rank, bool(np.array_equal(total, want))))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_proof_sums_fold_world2():
    """The exchange step of a proof spread over ranks (bellman_amd.sharding.fold_sums): all-gather of
    the 960-byte multiexp-result records and slot-wise group addition (host code, no GPU needed)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_sums_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]
