#!/usr/bin/env python3
"""Fuzz of the C++ mirror's host side against the oracle (no GPU): random fixture circuits (kind 2: every form of linear
combination, kind 3: structure drawn from the seed) of random sizes, seeds and witnesses - ProvingAssignment evaluations,
assignments and density maps == oracle/pyref/prover.py, and the structure capture reproduces its own assignment.
    [r5] one draw in eight is a MisuseCircuit (kind 4: a closure that discards terms it added to a copy of its argument,
    returns one of two branches, or touches its argument and returns a stored combination): the checking build of the test
    library must refuse it, in ProvingAssignment::enforce and in the structure capture.
    python tools/fuzz_mirror.py [rng seed] [circuits]        (1200 circuits ran clean at the end of round 4)"""
import ctypes, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bellman_amd import _lib, groth16 as pg
from oracle import cref
from oracle.pyref import prover as oprover
from oracle.pyref.core import INPUT, Variable
from tests import circuits
lib = _lib.load()
lib.bh_test_capture_check.restype = ctypes.c_double
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_ok = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 200):
    kind = rnd.choice([2, 3, 3, 3, 2, 3, 3, 4])
    rounds = rnd.choice([1, 2, 3, 5, 17, 64, 65, 129, 300])
    seed = rnd.getrandbits(64)
    x0 = rnd.randrange(circuits.Q)
    if kind == 4:
        try:
            pg.demo_assignment(kind, rounds, seed, [x0])
            raise SystemExit("misuse not detected: rounds %d seed %d" % (rounds, seed))
        except AssertionError:
            pass
        out4 = (ctypes.c_size_t * 4)()
        assert lib.bh_test_capture_check(kind, ctypes.c_size_t(rounds), ctypes.c_uint64(seed), out4) < 0, (rounds, seed)
        n_ok += 1
        continue
    asg = pg.demo_assignment(kind, rounds, seed, [x0])
    pa = oprover.ProvingAssignment(circuits.Q)
    pa.alloc_input(lambda: 1)
    (circuits.forms_circuit if kind == 2 else circuits.random_circuit)(rounds, seed, x0)(pa)
    for i in range(len(pa.input_assignment)):
        pa.enforce(lambda lc: lc + Variable(INPUT, i), lambda lc: lc, lambda lc: lc)
    for key in ("a", "b", "c", "input_assignment", "aux_assignment"):
        assert cref.arr_to_ints(cref.fr_from_mont(asg[key])) == [v % circuits.Q for v in getattr(pa, key)], (kind, rounds, seed, key)
    for key in ("a_aux_density", "b_input_density", "b_aux_density"):
        want = getattr(pa, key).bv
        bits = np.unpackbits(asg[key].view(np.uint8), bitorder="little")[:len(want)].astype(bool)
        assert list(bits) == [bool(b) for b in want], (kind, rounds, seed, key)
    out4 = (ctypes.c_size_t * 4)()
    ms = lib.bh_test_capture_check(kind, ctypes.c_size_t(rounds), ctypes.c_uint64(seed), out4)
    assert ms >= 0 and out4[0] == len(pa.a) and out4[3] == 0, (kind, rounds, seed, list(out4))
    n_ok += 1
print("fuzz ok:", n_ok, "circuits")
