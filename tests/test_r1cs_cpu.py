"""Host logic of the R1CS capture (bellman_amd.groth16.ShapeAssembly): the captured matrices,
evaluated with Python integers, must reproduce what the reference's ProvingAssignment computes while
it synthesises (groth16/src/prover.rs:19-55,105-145,208-215) - evaluations and the three densities."""

import random

from bellman_amd import groth16 as pg
from tests import circuits

Q = pg.Q


def _reference_assignment(circuit):
    prover = pg.ProvingAssignment()
    prover.alloc_input(lambda: 1)
    circuit(prover)
    for i in range(len(prover.input_assignment)):
        prover.enforce(lambda lc, i=i: lc + pg.Variable(pg.INPUT, i), lambda lc: lc, lambda lc: lc)
    return prover


def _spmv(matrix, table, w):
    row_ptr, var, coeff = matrix
    return [sum(table[coeff[t]] * w[var[t]] for t in range(row_ptr[i], row_ptr[i + 1])) % Q for i in range(len(row_ptr) - 1)]


def _check(circuit_with_witness, circuit_shape_only):
    want = _reference_assignment(circuit_with_witness)
    cs = pg.ShapeAssembly.capture(circuit_shape_only)
    matrices, table = cs.csr()
    assert table[0] == 1 and 0 not in table
    assert (cs.num_inputs, cs.num_aux) == (len(want.input_assignment), len(want.aux_assignment))
    wit = pg.WitnessAssignment()
    wit.alloc_input(lambda: 1)
    circuit_with_witness(wit)
    assert wit.input_assignment == want.input_assignment and wit.aux_assignment == want.aux_assignment
    w = wit.input_assignment + wit.aux_assignment
    assert _spmv(matrices[0], table, w) == want.a
    assert _spmv(matrices[1], table, w) == want.b
    assert _spmv(matrices[2], table, w) == want.c
    # densities from structure alone
    n_in = cs.num_inputs
    a_aux = [False] * cs.num_aux
    b_in, b_aux = [False] * n_in, [False] * cs.num_aux
    for v in matrices[0][1]:
        if v >= n_in:
            a_aux[v - n_in] = True
    for v in matrices[1][1]:
        if v >= n_in:
            b_aux[v - n_in] = True
        else:
            b_in[v] = True
    assert a_aux == want.a_aux_density.bv and b_in == want.b_input_density.bv and b_aux == want.b_aux_density.bv


def test_capture_mimc():
    rnd = random.Random(5)
    cons = [rnd.randrange(Q) for _ in range(20)]
    _check(circuits.mimc_circuit(rnd.randrange(Q), rnd.randrange(Q), cons), circuits.mimc_circuit(0, 0, cons))


def test_capture_chain_with_zero_coefficient_terms():
    _check(circuits.chain_circuit(37, 11, 123456789), circuits.chain_circuit(37, 11, 0))
