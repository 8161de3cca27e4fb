"""One Groth16 proof spread over several ranks: each rank computes the eight multiexps
over its slice of the scalar indices, the 960-byte result records are summed slot-wise and every rank
assembles the proof - which must equal the single-GPU proof bit for bit, for every number of parts."""

import os
import random
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.cengine import CBls12  # noqa: E402
from oracle.pyref.generator import generate_parameters  # noqa: E402
from oracle.pyref.prover import create_proof as oracle_create_proof  # noqa: E402
from tests import circuits  # noqa: E402
from tests.test_gpu_groth16 import TOXIC, _product_params, _same, worker  # noqa: E402,F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q = circuits.Q
G1_GEN = None


def _witness(circ):
    from bellman_amd import groth16 as pg

    w = pg.WitnessAssignment()
    w.alloc_input(lambda: 1)
    circ(w)
    return w.input_assignment, w.aux_assignment


@pytest.mark.parametrize("parts", [1, 2, 3, 8])
def test_mimc_proof_from_parts_equals_oracle(worker, parts):
    from bellman_amd import groth16 as pg

    rnd = random.Random(44)
    cons = [rnd.randrange(Q) for _ in range(circuits.MIMC_ROUNDS)]
    xl, xr, r, s = (rnd.randrange(Q) for _ in range(4))
    circ = circuits.mimc_circuit(xl, xr, cons)
    p = generate_parameters(CBls12, circuits.mimc_circuit(0, 0, cons), CBls12.G1.gen, CBls12.G2.gen, **TOXIC)
    want = oracle_create_proof(CBls12, circ, p, r, s)
    pp = _product_params(worker, p)
    r1cs = pg.R1CS.from_circuit(worker, circuits.mimc_circuit(0, 0, cons))
    ia, aa = _witness(circ)
    total = None
    for part in range(parts):   # what `parts` ranks would compute, here one after the other on one GPU
        sums = pg.prove_witness_part(r1cs, pp, ia, aa, part, parts)
        total = sums if total is None else pg.sums_add(total, sums)
    assert _same(pg.assemble(pp, total, r, s), want.a, want.b, want.c)
    with pytest.raises(AssertionError):
        pg.prove_witness_part(r1cs, pp, ia, aa, parts, parts)   # part out of range


@pytest.mark.parametrize("log_n,parts", [(12, 5), (16, 2), (16, 8)])
def test_chain_proof_from_parts_equals_single_gpu(worker, log_n, parts):
    """densities that cut mid-vector (B uses every other aux variable), slices at multiples of 64"""
    from bellman_amd import groth16 as pg
    from tests.test_gpu_groth16 import _chain_setup

    rounds, seed, x0, r, s = (1 << log_n) - 3, 77, 31337, 0xDEADBEEF, 0xFEEDFACE
    pp, vk, _ = _chain_setup(worker, rounds, seed)
    r1cs = pg.R1CS.from_demo(worker, 1, rounds, seed)
    want = pg.create_proof_demo_r1cs(pp, r1cs, 1, rounds, seed, [x0], None, r, s)
    total = None
    for part in range(parts):
        sums = pg.prove_demo_part(pp, r1cs, 1, rounds, seed, [x0], None, part, parts)
        total = sums if total is None else pg.sums_add(total, sums)
    got = pg.assemble(pp, total, r, s)
    assert _same(got, want.a.tobytes(), want.b.tobytes(), want.c.tobytes())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import bellman_amd
    from bellman_amd import groth16 as pg
    from bellman_amd import sharding
    from tests.test_gpu_groth16 import _chain_setup

    dist.init_process_group("gloo", rank=rank, world_size=world)   # both ranks share GPU 0 on the test box
    w = bellman_amd.Worker(0)
    rounds, seed, x0, r, s = 4093, 9, 4242, 1234567, 7654321
    pp, vk, _ = _chain_setup(w, rounds, seed)
    r1cs = pg.R1CS.from_demo(w, 1, rounds, seed)
    whole = pg.create_proof_demo_r1cs(pp, r1cs, 1, rounds, seed, [x0], None, r, s)
    got = sharding.create_proof_sharded(lambda rk, wd: pg.prove_demo_part(pp, r1cs, 1, rounds, seed, [x0], None, rk, wd), pp, r, s)
    ok = got.a.tobytes() == whole.a.tobytes() and got.b.tobytes() == whole.b.tobytes() and got.c.tobytes() == whole.c.tobytes()
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()
    w.close()


def test_two_processes_one_proof():
    """the whole multi-rank path: 2 processes (gloo rendezvous; both on GPU 0 here), sliced multiexps,
    all-gather, fold, assembly - same proof as a single process"""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]
