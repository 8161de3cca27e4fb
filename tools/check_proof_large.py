"""One-off parity check of a large proof: device (R1CS resident, bh_groth16_prove_demo_r1cs) against the C
restatement of the prover (oracle/cprover.py) on the synthetic chain circuit with 2^log_n constraints and a
synthetic CRS of distinct prime-order points.  Usage: python tools/check_proof_large.py [log_n=22]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bellman_amd
from bellman_amd import groth16 as pg
from oracle import cprover, cref
from tests import circuits


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    rounds = (1 << log_n) - 3
    seed, x0, r, s = 4242, 1234567, 0xABCDEF0123456789, 0x1234567890ABCDEF
    m = 1 << log_n
    n_aux, nb = rounds + 1, (rounds + 1) // 2 + 2
    t0 = time.time()
    h, l = cref.gen_bases(1, m - 1, a=11, b=3), cref.gen_bases(1, n_aux, a=5, b=7)
    a, b1, b2 = cref.gen_bases(1, n_aux + 2, a=2, b=9), cref.gen_bases(1, nb, a=13, b=4), cref.gen_bases(2, nb, a=17, b=6)
    g1, g2 = cref.g1_generator(), cref.g2_generator()
    vk = dict(alpha_g1=cref.point_mul(1, g1, 101), beta_g1=cref.point_mul(1, g1, 102), beta_g2=cref.point_mul(2, g2, 102),
              delta_g1=cref.point_mul(1, g1, 103), delta_g2=cref.point_mul(2, g2, 103))
    print("CRS made in %.1f s" % (time.time() - t0), flush=True)
    w = bellman_amd.Worker(0)
    pp = pg.Parameters(w, vk["alpha_g1"], vk["beta_g1"], vk["beta_g2"], vk["delta_g1"], vk["delta_g2"], h, l, a, b1, b2)
    r1cs = pg.R1CS.from_demo(w, 1, rounds, seed)
    tm = [0, 0, 0, 0]
    got = pg.create_proof_demo_r1cs(pp, r1cs, 1, rounds, seed, [x0], None, r, s, tm)
    print("device proof: host ms [witness, h, msm, total] =", [round(x, 1) for x in tm], flush=True)
    t0 = time.time()
    f = circuits.chain_assignment_fast(rounds, seed, x0)
    print("assignment (python) %.1f s" % (time.time() - t0), flush=True)
    tc = {}
    want = cprover.prove_assignment(f["a"], f["b"], f["c"], f["input_assignment"], f["aux_assignment"], f["a_aux_density"],
                                    f["b_input_density"], f["b_aux_density"], vk, h, l, a, b1, b2, r, s,
                                    threads=cref.lib().orc_max_threads(), concurrent=True, timing=tc)
    print("C restatement: %.1f s" % tc["total_s"], flush=True)
    ok = got.a.tobytes() == want[0].tobytes() and got.b.tobytes() == want[1].tobytes() and got.c.tobytes() == want[2].tobytes()
    print("2^%d-constraint proof bit-identical to the oracle:" % log_n, ok)
    sys.exit(0 if ok else 1)


main()
