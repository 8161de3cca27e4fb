"""CPU-side consistency of the prover layers:
  * the product's Python ProvingAssignment (bellman_amd/groth16.py) == the oracle's restated one,
  * the three oracle provers agree (pure-Python BLS / C-backed engine / C multiexp+FFT fast path),
  * the closed-form chain assignment == generic synthesis."""

import random

import numpy as np

from oracle import cref
from oracle.cengine import CBls12
from oracle import cprover
from oracle.pyref import bls12_381 as bls
from oracle.pyref.core import INPUT, Variable
from oracle.pyref.engines import Bls12
from oracle.pyref.generator import generate_parameters
from oracle.pyref.prover import ProvingAssignment, create_proof
from tests import circuits

Q = bls.Q
TOXIC = dict(alpha=48577, beta=22580, gamma=53332, delta=5481, tau=3673)


def _synth(PA, circuit):
    pa = PA() if PA is not ProvingAssignment else ProvingAssignment(Q)
    pa.alloc_input(lambda: 1)
    circuit(pa)
    for i in range(len(pa.input_assignment)):
        var = Variable(INPUT, i) if PA is ProvingAssignment else type(pa.one())(0, i)
        pa.enforce(lambda lc, var=var: lc + var, lambda lc: lc, lambda lc: lc)
    return pa


def test_product_proving_assignment_matches_oracle():
    from bellman_amd import groth16 as pg

    rnd = random.Random(1)
    cons = [rnd.randrange(Q) for _ in range(7)]
    for circ in (circuits.mimc_circuit(rnd.randrange(Q), rnd.randrange(Q), cons), circuits.chain_circuit(9, 5, 12345)):
        o = _synth(ProvingAssignment, circ)
        p = _synth(pg.ProvingAssignment, circ)
        assert (o.a, o.b, o.c) == (p.a, p.b, p.c)
        assert o.input_assignment == p.input_assignment and o.aux_assignment == p.aux_assignment
        assert o.a_aux_density.bv == p.a_aux_density.bv
        assert o.b_input_density.bv == p.b_input_density.bv and o.b_aux_density.bv == p.b_aux_density.bv


def test_chain_assignment_closed_form():
    for rounds in (1, 2, 7, 40):
        o = _synth(ProvingAssignment, circuits.chain_circuit(rounds, 99, 777))
        f = circuits.chain_assignment_fast(rounds, 99, 777)
        assert (o.a, o.b, o.c) == (f["a"], f["b"], f["c"])
        assert o.input_assignment == f["input_assignment"] and o.aux_assignment == f["aux_assignment"]
        assert o.a_aux_density.bv == f["a_aux_density"]
        assert o.b_input_density.bv == f["b_input_density"] and o.b_aux_density.bv == f["b_aux_density"]


def test_oracle_provers_agree_on_small_mimc():
    rnd = random.Random(2)
    cons = [rnd.randrange(Q) for _ in range(3)]
    xl, xr = rnd.randrange(Q), rnd.randrange(Q)
    circ = circuits.mimc_circuit(xl, xr, cons)
    r, s = rnd.randrange(Q), rnd.randrange(Q)
    # pure Python big-int engine (KAT-pinned code path, affine formulas)
    p_py = generate_parameters(Bls12, circ, bls.G1_GEN, bls.G2_GEN, **TOXIC)
    proof_py = create_proof(Bls12, circ, p_py, r, s)
    # same generic code, group law executed by the C oracle
    G1, G2 = CBls12.G1, CBls12.G2
    p_c = generate_parameters(CBls12, circ, G1.gen, G2.gen, **TOXIC)
    proof_c = create_proof(CBls12, circ, p_c, r, s)
    assert cref.g1_to_py(np.frombuffer(proof_c.a, dtype=np.uint64))[0] == proof_py.a
    assert cref.g2_to_py(np.frombuffer(proof_c.b, dtype=np.uint64))[0] == proof_py.b
    assert cref.g1_to_py(np.frombuffer(proof_c.c, dtype=np.uint64))[0] == proof_py.c
    # fast path: C multiexp + C FFT pipeline on the synthesised assignment
    pa = _synth(ProvingAssignment, circ)
    vk = {k: np.frombuffer(getattr(p_c.vk, k), dtype=np.uint64) for k in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2")}
    got = cprover.prove_assignment(pa.a, pa.b, pa.c, pa.input_assignment, pa.aux_assignment, pa.a_aux_density.bv,
                                   pa.b_input_density.bv, pa.b_aux_density.bv, vk, G1.to_array(p_c.h), G1.to_array(p_c.l),
                                   G1.to_array(p_c.a), G1.to_array(p_c.b_g1), G2.to_array(p_c.b_g2), r, s)
    assert got[0].tobytes() == proof_c.a and got[1].tobytes() == proof_c.b and got[2].tobytes() == proof_c.c


def test_cprover_concurrent_issue_gives_the_same_proof():
    """oracle/cprover.py: issuing the eight multiexps together (prover.rs:244-318) is unobservable."""
    from oracle import cprover, cref
    from tests import circuits

    rounds, seed, x0 = 70, 5, 424242
    f = circuits.chain_assignment_fast(rounds, seed, x0)
    m = 128
    h, l = cref.gen_bases(1, m - 1, a=11, b=3), cref.gen_bases(1, rounds + 1, a=5, b=7)
    a, b1, b2 = cref.gen_bases(1, rounds + 3, a=2, b=9), cref.gen_bases(1, 40, a=13, b=4), cref.gen_bases(2, 40, a=17, b=6)
    g1, g2 = cref.g1_generator(), cref.g2_generator()
    vk = dict(alpha_g1=cref.point_mul(1, g1, 101), beta_g1=cref.point_mul(1, g1, 102), beta_g2=cref.point_mul(2, g2, 102),
              delta_g1=cref.point_mul(1, g1, 103), delta_g2=cref.point_mul(2, g2, 103))
    args = (f["a"], f["b"], f["c"], f["input_assignment"], f["aux_assignment"], f["a_aux_density"], f["b_input_density"],
            f["b_aux_density"], vk, h, l, a, b1, b2, 77, 88)
    tm = {}
    p1 = cprover.prove_assignment(*args)
    p2 = cprover.prove_assignment(*args, threads=4, concurrent=True, timing=tm)
    assert all(x.tobytes() == y.tobytes() for x, y in zip(p1, p2)) and tm["total_s"] >= tm["multiexp_and_h_s"] > 0
