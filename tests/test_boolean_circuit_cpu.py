"""The boolean-heavy demo circuit (BoolMixCircuit, csrc/demo_circuits.cpp kind 5 - the shape of src/gadgets/boolean.rs):
the C++ mirror's ProvingAssignment == the oracle's ProvingAssignment (oracle/pyref/prover.py, prover.rs:57-162) on
tests.circuits.boolmix_circuit == the direct computation boolmix_assignment_fast the large GPU tests feed the restated
prover with; and the aux assignment is what the circuit is for: almost only zeros and ones.  Host code only."""

import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.mark.parametrize("rounds", [0, 1, 5, 63, 64, 200, 1000])
def test_boolmix_mirror_oracle_and_fast_form_agree(rounds):
    from bellman_amd import _lib
    from bellman_amd import groth16 as pg
    from oracle import cref
    from oracle.pyref import prover as oprover
    from oracle.pyref.core import INPUT, Variable
    from tests import circuits

    seed, x0 = 31 + rounds, 0xFEDCBA9876543210F0F0
    asg = pg.demo_assignment(5, rounds, seed, [x0])
    pa = oprover.ProvingAssignment(circuits.Q)
    pa.alloc_input(lambda: 1)
    circuits.boolmix_circuit(rounds, seed, x0)(pa)
    for i in range(len(pa.input_assignment)):   # prover.rs:208-215
        pa.enforce(lambda lc: lc + Variable(INPUT, i), lambda lc: lc, lambda lc: lc)
    f = circuits.boolmix_assignment_fast(rounds, seed, x0)
    for key in ("a", "b", "c", "input_assignment", "aux_assignment"):
        want = [v % circuits.Q for v in getattr(pa, key)]
        assert cref.arr_to_ints(cref.fr_from_mont(asg[key])) == want, key
        assert [v % circuits.Q for v in f[key]] == want, key
    for key in ("a_aux_density", "b_input_density", "b_aux_density"):
        want = [bool(b) for b in getattr(pa, key).bv]
        bits = np.unpackbits(asg[key].view(np.uint8), bitorder="little")[:len(want)].astype(bool)
        assert list(bits) == want, key
        assert [bool(b) for b in f[key]] == want, key
    # every constraint holds (the circuit is satisfiable, unlike the fixtures of test_round3_cpu.py)
    for a, b, c in zip(pa.a, pa.b, pa.c):
        assert a * b % circuits.Q == c % circuits.Q
    # the structure capture of the same circuit reproduces its ProvingAssignment
    lib = _lib.load()
    lib.bh_test_capture_check.restype = ctypes.c_double
    out4 = (ctypes.c_size_t * 4)()
    ms = lib.bh_test_capture_check(5, ctypes.c_size_t(rounds), ctypes.c_uint64(seed), out4)
    assert ms >= 0 and out4[0] == len(pa.a) and out4[3] == 0, list(out4)


def test_boolmix_aux_assignment_is_boolean_heavy():
    from tests import circuits

    rounds = circuits.boolmix_rounds(14)
    f = circuits.boolmix_assignment_fast(rounds, 5, 0x123456789ABCDEF)
    assert len(f["a"]) <= 1 << 14 < len(f["a"]) + 70
    aux = f["aux_assignment"]
    booleans = sum(1 for v in aux if v in (0, 1))
    ones = sum(1 for v in aux if v == 1)
    assert booleans >= 0.98 * len(aux)
    assert 0.3 * len(aux) < ones < 0.6 * len(aux)
    a_dense, b_dense = sum(f["a_aux_density"]), sum(f["b_aux_density"])
    assert 0.3 * len(aux) < a_dense < len(aux) and 0.3 * len(aux) < b_dense < len(aux)
