"""GPU parity of generate_parameters (groth16/src/generator.rs:163-510): every query
vector, the verifying key and the serialized bytes against the oracle's generic restatement."""

import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cref  # noqa: E402
from oracle.cengine import CBls12  # noqa: E402
from oracle.pyref import bls12_381 as bls  # noqa: E402
from oracle.pyref import params_io as pio  # noqa: E402
from oracle.pyref.generator import generate_parameters  # noqa: E402
from oracle.pyref.prover import create_proof as oracle_create_proof  # noqa: E402
from tests import circuits  # noqa: E402
from tests.test_gpu_groth16 import TOXIC, _same, worker  # noqa: E402,F401

Q = bls.Q


def _recs(group, pts):
    w = 12 if group == 1 else 24
    return np.frombuffer(b"".join(bytes(p) for p in pts), dtype=np.uint64).reshape(-1, w)


def _check_against_oracle(worker, shape_circuit, toxic):
    from bellman_amd import groth16 as pg

    want = generate_parameters(CBls12, shape_circuit, CBls12.G1.gen, CBls12.G2.gen, **toxic)
    r1cs = pg.R1CS.from_circuit(worker, shape_circuit)
    got = pg.Parameters.generate(worker, r1cs, _recs(1, [CBls12.G1.gen])[0], _recs(2, [CBls12.G2.gen])[0], toxic["alpha"],
                                 toxic["beta"], toxic["gamma"], toxic["delta"], toxic["tau"])
    for name, group in (("h", 1), ("l", 1), ("a", 1), ("b_g1", 1), ("b_g2", 2)):
        w = _recs(group, getattr(want, name))
        g = got.query(name)
        assert g.shape == w.shape and (g == w).all(), name
    vk = got.vk()
    for g, name in zip(vk, ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2")):
        assert g.tobytes() == bytes(getattr(want.vk, name)), name
    gamma, ic = got.vk_ext()
    assert gamma.tobytes() == bytes(want.vk.gamma_g2)
    assert (ic == _recs(1, want.vk.ic)).all()
    return want, got, r1cs


def test_generate_mimc_then_write_read_prove(worker):
    from bellman_amd import groth16 as pg

    rnd = random.Random(7)
    rounds = 25
    cons = [rnd.randrange(Q) for _ in range(rounds)]
    want, got, r1cs = _check_against_oracle(worker, circuits.mimc_circuit(0, 0, cons), TOXIC)
    # Parameters::write bytes against the oracle's writer
    py = lambda group, pts: (cref.g1_to_py if group == 1 else cref.g2_to_py)(_recs(group, pts))  # noqa: E731
    vk = dict(alpha_g1=py(1, [want.vk.alpha_g1])[0], beta_g1=py(1, [want.vk.beta_g1])[0], beta_g2=py(2, [want.vk.beta_g2])[0],
              gamma_g2=py(2, [want.vk.gamma_g2])[0], delta_g1=py(1, [want.vk.delta_g1])[0], delta_g2=py(2, [want.vk.delta_g2])[0],
              ic=py(1, want.vk.ic))
    blob = pio.parameters_write(vk, py(1, want.h), py(1, want.l), py(1, want.a), py(1, want.b_g1), py(2, want.b_g2))
    assert got.write() == blob
    # round trip through the product's reader, then a proof with the generated parameters (device R1CS path)
    again = pg.Parameters.read(worker, got.write(), True)
    assert again.write() == blob
    xl, xr, r, s = (rnd.randrange(Q) for _ in range(4))
    circ = circuits.mimc_circuit(xl, xr, cons)
    proof_want = oracle_create_proof(CBls12, circ, want, r, s)
    for params in (got, again):
        assert _same(pg.create_proof_r1cs(circ, r1cs, params, r, s), proof_want.a, proof_want.b, proof_want.c)


@pytest.mark.parametrize("rounds", [1, 2, 9, 130, 1300])
def test_generate_chain_circuit(worker, rounds):
    """zero-coefficient terms, variables absent from B (identities filtered out of the B queries),
    a domain that is not full (n_cons < m)"""
    toxic = dict(alpha=3 + rounds, beta=Q - 1000003, gamma=123456789, delta=Q // 3, tau=987654321 + rounds)
    want, got, _ = _check_against_oracle(worker, circuits.chain_circuit(rounds, 11 + rounds, 0), toxic)
    assert len(want.b_g1) < len(want.a) or rounds == 1


def test_generate_unconstrained_by_cancellation(worker):
    """alpha = -beta makes (at*beta + bt*alpha + ct) vanish for a variable used identically in A and B:
    the oracle reports UnconstrainedVariable (generator.rs:464-470) and so must the device path."""
    import bellman_amd
    from bellman_amd import groth16 as pg
    from oracle.pyref import errors as oerr

    toxic = dict(alpha=5, beta=Q - 5, gamma=123456789, delta=Q // 3, tau=987654323)
    shape = circuits.chain_circuit(2, 13, 0)
    with pytest.raises(oerr.UnconstrainedVariable):
        generate_parameters(CBls12, shape, CBls12.G1.gen, CBls12.G2.gen, **toxic)
    with pytest.raises(bellman_amd.UnconstrainedVariable):
        pg.Parameters.generate(worker, pg.R1CS.from_circuit(worker, shape), _recs(1, [CBls12.G1.gen])[0],
                               _recs(2, [CBls12.G2.gen])[0], toxic["alpha"], toxic["beta"], toxic["gamma"], toxic["delta"], toxic["tau"])


def test_generate_errors(worker):
    import bellman_amd
    from bellman_amd import groth16 as pg

    g1, g2 = _recs(1, [CBls12.G1.gen])[0], _recs(2, [CBls12.G2.gen])[0]
    r1cs = pg.R1CS.from_circuit(worker, circuits.chain_circuit(5, 1, 0))
    for bad in ("gamma", "delta"):   # generator.rs:227-243
        t = dict(TOXIC)
        t[bad] = 0
        with pytest.raises(bellman_amd.UnexpectedIdentity):
            pg.Parameters.generate(worker, r1cs, g1, g2, t["alpha"], t["beta"], t["gamma"], t["delta"], t["tau"])

    def unconstrained(cs):   # a variable that appears in no constraint: generator.rs:464-470
        cs.alloc(lambda: 1)
        x = cs.alloc(lambda: 2)
        cs.enforce(lambda lc: lc + x, lambda lc: lc + cs.one(), lambda lc: lc + x)

    with pytest.raises(bellman_amd.UnconstrainedVariable):
        pg.Parameters.generate(worker, pg.R1CS.from_circuit(worker, unconstrained), g1, g2, *[TOXIC[k] for k in ("alpha", "beta", "gamma", "delta", "tau")])


def test_mimc_322_generate_prove_verify_like_the_reference_test(worker):
    """groth16/tests/mimc.rs:38-101 end to end on the product: random trapdoors -> generate_parameters
    (device), create_random_proof (device, R1CS resident), Proof::write = 192 bytes, and the proof
    satisfies the verification equation (oracle/pyref/pairing.py restates verify_proof) for the right
    image only."""
    from bellman_amd import groth16 as pg
    from oracle.pyref import pairing

    rnd = random.Random(31415)
    cons = [rnd.randrange(Q) for _ in range(circuits.MIMC_ROUNDS)]
    xl, xr = rnd.randrange(Q), rnd.randrange(Q)
    image = circuits.mimc_hash(xl, xr, cons)
    r1cs = pg.R1CS.from_demo(worker, 0, circuits.MIMC_ROUNDS, 0, cons)
    g1, g2 = _recs(1, [CBls12.G1.gen])[0], _recs(2, [CBls12.G2.gen])[0]
    params = pg.Parameters.generate(worker, r1cs, g1, g2, *[rnd.randrange(1, Q) for _ in range(5)])
    proof = pg.create_random_proof(circuits.mimc_circuit(xl, xr, cons), params, rng=rnd, r1cs=r1cs)
    assert len(proof.write()) == 192
    alpha_g1, _, beta_g2, _, delta_g2 = params.vk()
    gamma_g2, ic = params.vk_ext()
    vk = dict(alpha_g1=cref.g1_to_py(alpha_g1)[0], beta_g2=cref.g2_to_py(beta_g2)[0], gamma_g2=cref.g2_to_py(gamma_g2)[0],
              delta_g2=cref.g2_to_py(delta_g2)[0], ic=cref.g1_to_py(ic))
    pr = (cref.g1_to_py(proof.a)[0], cref.g2_to_py(proof.b)[0], cref.g1_to_py(proof.c)[0])
    assert pairing.verify_proof(vk, pr, [image])
    assert not pairing.verify_proof(vk, pr, [(image + 1) % Q])
