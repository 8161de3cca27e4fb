"""The oracle's prover and generator pinned on BLS12-381 by the verification equation - the check the
reference's own MiMC test makes (groth16/tests/mimc.rs:38-101: generate parameters, prove, and
`verify_proof(&pvk, &proof, &[image])` must succeed).  oracle/pyref/pairing.py restates verify_proof
(groth16/src/verifier.rs:23-58) on Python integers."""

import random

import numpy as np

from oracle import cref
from oracle.cengine import CBls12
from oracle.pyref import bls12_381 as bls
from oracle.pyref import pairing
from oracle.pyref.generator import generate_parameters
from oracle.pyref.prover import create_proof
from tests import circuits

Q = bls.Q


def _py(group, recs):
    return (cref.g1_to_py if group == 1 else cref.g2_to_py)(np.frombuffer(b"".join(bytes(r) for r in recs), dtype=np.uint64))


def vk_to_py(vk):
    return dict(alpha_g1=_py(1, [vk.alpha_g1])[0], beta_g2=_py(2, [vk.beta_g2])[0], gamma_g2=_py(2, [vk.gamma_g2])[0],
                delta_g2=_py(2, [vk.delta_g2])[0], ic=_py(1, vk.ic))


def test_pairing_is_bilinear_and_non_degenerate():
    G1, G2 = bls.G1, bls.G2
    a, b = 0x1234567, 0x89ABCDEF
    assert pairing.pairing_product_is_one([(G1.mul(G1.gen, a), G2.mul(G2.gen, b)), (G1.neg(G1.mul(G1.gen, a * b % Q)), G2.gen)])
    assert pairing.pairing_product_is_one([(G1.mul(G1.gen, a), G2.gen), (G1.neg(G1.gen), G2.mul(G2.gen, a))])
    assert not pairing.pairing_product_is_one([(G1.gen, G2.gen)])
    assert pairing.pairing_product_is_one([(None, G2.gen), (G1.gen, None)])


def test_oracle_mimc_proof_verifies():
    rnd = random.Random(2718)
    rounds = 12
    cons = [rnd.randrange(Q) for _ in range(rounds)]
    xl, xr = rnd.randrange(Q), rnd.randrange(Q)
    image = circuits.mimc_hash(xl, xr, cons)
    toxic = {k: rnd.randrange(1, Q) for k in ("alpha", "beta", "gamma", "delta", "tau")}   # like mimc.rs: random trapdoors
    params = generate_parameters(CBls12, circuits.mimc_circuit(0, 0, cons), CBls12.G1.gen, CBls12.G2.gen, **toxic)
    proof = create_proof(CBls12, circuits.mimc_circuit(xl, xr, cons), params, rnd.randrange(Q), rnd.randrange(Q))
    vk = vk_to_py(params.vk)
    pr = (_py(1, [proof.a])[0], _py(2, [proof.b])[0], _py(1, [proof.c])[0])
    assert pairing.verify_proof(vk, pr, [image])
    assert not pairing.verify_proof(vk, pr, [(image + 1) % Q])                     # wrong public input
    assert not pairing.verify_proof(vk, (pr[0], pr[1], bls.G1.double(pr[2])), [image])   # tampered proof
