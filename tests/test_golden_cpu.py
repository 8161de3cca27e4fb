"""Golden fixtures (tests/golden/*.json): the C oracle reproduces the frozen pyref vectors, the
xordemo KAT file matches the reference-derived constants used by the KAT test, and the Zcash proof
encoding has the reference's size (192 bytes, groth16/src/lib.rs:559)."""

import json
import os

import numpy as np

from oracle import cprover, cref
from oracle.cengine import CBls12
from oracle.pyref.core import INPUT, Variable
from oracle.pyref.generator import generate_parameters
from oracle.pyref.prover import ProvingAssignment
from tests import circuits

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "bls12_381_small.json")))
Q = cref.Q


def ints(xs):
    return [int(x, 16) for x in xs]


def g1(p):
    return None if p is None else (int(p[0], 16), int(p[1], 16))


def g2(p):
    return None if p is None else ((int(p[0][0], 16), int(p[0][1], 16)), (int(p[1][0], 16), int(p[1][1], 16)))


def test_xordemo_kat_file():
    kat = json.load(open(os.path.join(HERE, "golden", "xordemo_kat.json")))
    assert kat["modulus"] == 64513 and kat["h_coefficients"] == [5040, 11763, 10755, 63633, 128, 9747, 8739]
    assert kat["root_of_unity_2^3"] == 20201 and kat["u_i"] == [59158, 48317, 21767, 10402]


def test_c_oracle_fft_matches_golden():
    mont = cref.fr_to_mont(cref.ints_to_arr(ints(G["fft8"]["input"]), 4))
    for mode, name in enumerate(("fft", "ifft", "coset_fft", "icoset_fft")):
        got = cref.arr_to_ints(cref.fr_from_mont(cref.fft(mont, mode, threads=8)))
        assert got == ints(G["fft8"][name]), name


def test_c_oracle_msm_matches_golden():
    for gname, group, conv, dec in (("g1", 1, cref.g1_to_py, g1), ("g2", 2, cref.g2_to_py, g2)):
        m = G["msm_" + gname]
        gen = cref.g1_generator() if group == 1 else cref.g2_generator()
        bases = np.stack([cref.point_mul(group, gen, k) for k in ints(m["base_scalars"])])
        rc, got = cref.multiexp(group, bases, m["skip"], cref.density_bitmap(m["density"]), cref.ints_to_arr(ints(m["scalars"]), 4))
        assert rc == 0 and conv(got)[0] == dec(m["result"])


def test_c_prover_matches_golden_proof():
    m = G["mimc3_proof"]
    cons, xl, xr, r, s = ints(m["constants"]), int(m["xl"], 16), int(m["xr"], 16), int(m["r"], 16), int(m["s"], 16)
    assert circuits.mimc_hash(xl, xr, cons) == int(m["image"], 16)
    circ = circuits.mimc_circuit(xl, xr, cons)
    G1, G2 = CBls12.G1, CBls12.G2
    p = generate_parameters(CBls12, circ, G1.gen, G2.gen, **m["toxic"])
    pa = ProvingAssignment(Q)
    pa.alloc_input(lambda: 1)
    circ(pa)
    for i in range(len(pa.input_assignment)):
        pa.enforce(lambda lc, i=i: lc + Variable(INPUT, i), lambda lc: lc, lambda lc: lc)
    vk = {k: np.frombuffer(getattr(p.vk, k), dtype=np.uint64) for k in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2")}
    a, b, c = cprover.prove_assignment(pa.a, pa.b, pa.c, pa.input_assignment, pa.aux_assignment, pa.a_aux_density.bv,
                                       pa.b_input_density.bv, pa.b_aux_density.bv, vk, G1.to_array(p.h), G1.to_array(p.l),
                                       G1.to_array(p.a), G1.to_array(p.b_g1), G2.to_array(p.b_g2), r, s)
    assert cref.g1_to_py(a)[0] == g1(m["a"]) and cref.g2_to_py(b)[0] == g2(m["b"]) and cref.g1_to_py(c)[0] == g1(m["c"])
    assert len(bytes.fromhex(m["proof_bytes_zcash"])) == 192


def test_golden_parameters_bytes_and_assignment():
    """frozen Parameters::write bytes and ProvingAssignment of the 3-round MiMC fixture: the oracle's reader
    accepts the bytes (checked), re-serialising gives them back, the C-engine generator reproduces them, and
    the product's host-side R1CS capture evaluates to the frozen a/b/c and densities."""
    from bellman_amd import groth16 as pg
    from oracle.pyref import params_io as pio
    from tests import circuits

    m = G["mimc3_proof"]
    blob = bytes.fromhex(m["parameters_bytes"])
    got = pio.parameters_read(blob, True)
    assert pio.parameters_write(got["vk"], got["h"], got["l"], got["a"], got["b_g1"], got["b_g2"]) == blob
    ints = lambda xs: [int(x, 16) for x in xs]  # noqa: E731
    cons, xl, xr = ints(m["constants"]), int(m["xl"], 16), int(m["xr"], 16)
    cs = pg.ShapeAssembly.capture(circuits.mimc_circuit(0, 0, cons))
    matrices, table = cs.csr()
    w = ints(m["assignment"]["inputs"]) + ints(m["assignment"]["aux"])
    wit = pg.WitnessAssignment()
    wit.alloc_input(lambda: 1)
    circuits.mimc_circuit(xl, xr, cons)(wit)
    assert wit.input_assignment + wit.aux_assignment == w
    for (row_ptr, var, coeff), name in zip(matrices, "abc"):
        vals = [sum(table[coeff[t]] * w[var[t]] for t in range(row_ptr[i], row_ptr[i + 1])) % pg.Q for i in range(len(row_ptr) - 1)]
        assert vals == ints(m["assignment"][name]), name
