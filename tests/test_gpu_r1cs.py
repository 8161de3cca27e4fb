"""GPU parity of the device-resident R1CS path: a = A.w, b = B.w, c = C.w on the
device against Python integers, the structure-derived densities against the reference's
DensityTracker rule (prover.rs:31-44), and whole proofs made from the witness alone against the
oracle's create_proof - bit-exact, like every other proof test."""

import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cprover, cref  # noqa: E402
from oracle.cengine import CBls12  # noqa: E402
from oracle.pyref.generator import generate_parameters  # noqa: E402
from oracle.pyref.prover import create_proof as oracle_create_proof  # noqa: E402
from tests import circuits  # noqa: E402
from tests.test_gpu_groth16 import TOXIC, _chain_setup, _product_params, _same, worker  # noqa: E402,F401

Q = circuits.Q
R = (1 << 256) % Q
RINV = pow(R, -1, Q)


def _from_mont(arr):
    return [int.from_bytes(row.tobytes(), "little") * RINV % Q for row in arr]


@pytest.mark.parametrize("n_in,n_aux,n_cons,seed", [(1, 0, 1, 1), (1, 1, 1, 2), (3, 40, 64, 3), (2, 500, 1000, 4), (5, 3000, 4097, 5)])
def test_r1cs_eval_random_matrices(worker, n_in, n_aux, n_cons, seed):
    from bellman_amd import groth16 as pg

    rnd = random.Random(seed)
    table = [1, 0, Q - 1, 2] + [rnd.randrange(Q) for _ in range(12)]   # includes a zero coefficient (slot 1)
    nv = n_in + n_aux
    matrices = []
    for _ in range(3):
        row_ptr, var, coeff = [0], [], []
        for i in range(n_cons):
            k = rnd.choice([0, 0, 1, 1, 2, 3, 5, 40 if i % 97 == 0 else 2])
            if n_cons > 4000 and i in (7, 4000):
                k = 1025 + 2000 * (i == 7)   # rows above the lane-per-row limit: summed by a workgroup
            for _ in range(k):
                var.append(rnd.randrange(nv))
                coeff.append(rnd.choice([0, 0, 0, 1, 2] + list(range(len(table)))))
            row_ptr.append(len(var))
        matrices.append((row_ptr, var, coeff))
    r1cs = pg.R1CS.from_csr(worker, n_in, n_aux, matrices, table)
    assert (r1cs.num_inputs, r1cs.num_aux, r1cs.num_constraints) == (n_in, n_aux, n_cons)
    w_in = [1] + [rnd.randrange(Q) for _ in range(n_in - 1)]
    w_aux = [rnd.choice([0, 1, rnd.randrange(Q)]) for _ in range(n_aux)]
    w = w_in + w_aux
    got = r1cs.eval(w_in, w_aux)
    m = 1
    while m < n_cons:
        m *= 2
    for (row_ptr, var, coeff), out in zip(matrices, got):
        assert out.shape[0] == m
        want = [sum(table[coeff[t]] * w[var[t]] for t in range(row_ptr[i], row_ptr[i + 1])) % Q for i in range(n_cons)]
        assert _from_mont(out[:n_cons]) == want
        assert not out[n_cons:].any()   # EvaluationDomain::from_coeffs padding (domain.rs:68)
    # densities: only non-zero coefficients count; A tracks aux only, B inputs and aux, C nothing
    want_d = [np.zeros(n_aux, bool), np.zeros(n_in, bool), np.zeros(n_aux, bool)]
    for mat, (row_ptr, var, coeff) in enumerate(matrices[:2]):
        for v, ci in zip(var, coeff):
            if table[ci] == 0:
                continue
            if v >= n_in:
                want_d[0 if mat == 0 else 2][v - n_in] = True
            elif mat == 1:
                want_d[1][v] = True
    for which in range(3):
        bits, total = r1cs.density(which)
        assert (bits == want_d[which]).all() and total == int(want_d[which].sum())
    r1cs.release()


def test_r1cs_create_rejects_bad_input(worker):
    from bellman_amd import groth16 as pg

    ok = ([0, 1], [0], [0])
    with pytest.raises(AssertionError):      # coefficient slot 0 must be the constant 1
        pg.R1CS.from_csr(worker, 1, 0, [ok, ok, ok], [2])
    with pytest.raises(AssertionError):      # variable out of range
        pg.R1CS.from_csr(worker, 1, 0, [([0, 1], [1], [0]), ok, ok], [1])
    with pytest.raises(AssertionError):      # coefficient index out of range
        pg.R1CS.from_csr(worker, 1, 0, [([0, 1], [0], [1]), ok, ok], [1])


def test_mimc_322_from_witness_only(worker):
    """Config C1 through the R1CS path: matrices captured from the Python circuit and from the C++
    MiMCDemo; both give the oracle's proof."""
    from bellman_amd import groth16 as pg

    rnd = random.Random(322)
    cons = [rnd.randrange(Q) for _ in range(circuits.MIMC_ROUNDS)]
    xl, xr = rnd.randrange(Q), rnd.randrange(Q)
    r, s = rnd.randrange(Q), rnd.randrange(Q)
    circ = circuits.mimc_circuit(xl, xr, cons)
    p = generate_parameters(CBls12, circuits.mimc_circuit(0, 0, cons), CBls12.G1.gen, CBls12.G2.gen, **TOXIC)
    want = oracle_create_proof(CBls12, circ, p, r, s)
    pp = _product_params(worker, p)
    r1cs = pg.R1CS.from_circuit(worker, circuits.mimc_circuit(0, 0, cons))   # shape only: no witness needed
    assert (r1cs.num_inputs, r1cs.num_aux, r1cs.num_constraints) == (2, 645, 646)
    assert r1cs.density(0)[1] == 644 and r1cs.density(1)[1] == 1 and r1cs.density(2)[1] == 322
    got = pg.create_proof_r1cs(circ, r1cs, pp, r, s)
    assert _same(got, want.a, want.b, want.c)
    r1cs_cpp = pg.R1CS.from_demo(worker, 0, circuits.MIMC_ROUNDS, 0, cons)
    tm = [0, 0, 0, 0]
    got2 = pg.create_proof_demo_r1cs(pp, r1cs_cpp, 0, circuits.MIMC_ROUNDS, 0, [xl, xr], cons, r, s, tm)
    assert _same(got2, want.a, want.b, want.c)
    print("MiMC-322 create_proof (R1CS on device) host ms [witness, h, msm, total]:", tm)
    got3 = pg.create_proof_demo_r1cs(pp, r1cs, 0, circuits.MIMC_ROUNDS, 0, [xl, xr], cons, r, s)   # Python-captured matrices, C++ witness
    assert _same(got3, want.a, want.b, want.c)
    with pytest.raises(AssertionError):     # a witness of the wrong shape is refused, not mis-proved
        pg.prove_witness(r1cs, pp, [1, 2, 3], [0] * 645, r, s)


@pytest.mark.parametrize("rounds", [1, 2, 61, 4093])
def test_chain_from_witness_only(worker, rounds):
    from bellman_amd import groth16 as pg

    seed, x0 = 7 + rounds, 123456789
    r, s = 0x1234567 + rounds, 0x7654321
    pp, vk, (h, l, a, b1, b2) = _chain_setup(worker, rounds, seed)
    f = circuits.chain_assignment_fast(rounds, seed, x0)
    want = cprover.prove_assignment(f["a"], f["b"], f["c"], f["input_assignment"], f["aux_assignment"], f["a_aux_density"],
                                    f["b_input_density"], f["b_aux_density"], vk, h, l, a, b1, b2, r, s)
    r1cs = pg.R1CS.from_demo(worker, 1, rounds, seed)
    got = pg.create_proof_demo_r1cs(pp, r1cs, 1, rounds, seed, [x0], None, r, s)
    assert _same(got, want[0].tobytes(), want[1].tobytes(), want[2].tobytes())
    if rounds <= 61:
        r1cs_py = pg.R1CS.from_circuit(worker, circuits.chain_circuit(rounds, seed, 0))
        got_py = pg.create_proof_r1cs(circuits.chain_circuit(rounds, seed, x0), r1cs_py, pp, r, s)
        assert _same(got_py, want[0].tobytes(), want[1].tobytes(), want[2].tobytes())


def test_chain_2_20_from_witness_only(worker):
    """Config C4 with the constraint evaluation on the device: same proof as the host-synthesis path
    (which test_gpu_groth16.py pins against the oracle at this size)."""
    from bellman_amd import groth16 as pg

    rounds = (1 << 20) - 3
    seed, x0, r, s = 2020, 987654321, 0xABCDEF0123, 0x123456789AB
    pp, vk, _ = _chain_setup(worker, rounds, seed)
    r1cs = pg.R1CS.from_demo(worker, 1, rounds, seed)
    assert r1cs.num_constraints == 1 << 20
    tm, tm0 = [0, 0, 0, 0], [0, 0, 0, 0]
    got = pg.create_proof_demo_r1cs(pp, r1cs, 1, rounds, seed, [x0], None, r, s, tm)
    want = pg.create_proof_demo(pp, 1, rounds, seed, [x0], None, r, s, tm0)
    print("2^20 create_proof host ms [synthesis|witness, h, msm, total]: host-eval", tm0, " device-eval", tm)
    assert _same(got, want.a.tobytes(), want.b.tobytes(), want.c.tobytes())
