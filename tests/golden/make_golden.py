#!/usr/bin/env python3
"""Generates tests/golden/bls12_381_small.json from the KAT-pinned pure-Python oracle
(oracle/pyref, affine big-int formulas).  The reference itself cannot run here (Rust, crates not
vendored), so these are ORACLE outputs frozen for regression - they do not pin BLS12-381 parity to the
reference; the reference-pinned vectors are the test_xordemo values in
xordemo_kat.json, copied from /root/reference/groth16/src/tests/mod.rs.

Run:  python tests/golden/make_golden.py   (deterministic; rewrites the JSON files)"""

import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.pyref import bls12_381 as bls  # noqa: E402
from oracle.pyref import multiexp as pm  # noqa: E402
from oracle.pyref.domain import EvaluationDomain  # noqa: E402
from oracle.pyref.engines import Bls12  # noqa: E402
from oracle.pyref.generator import generate_parameters  # noqa: E402
from oracle.pyref.multicore import Worker  # noqa: E402
from oracle.pyref.prover import create_proof  # noqa: E402
from tests import circuits  # noqa: E402

Q = bls.Q
HERE = os.path.dirname(os.path.abspath(__file__))


def hx(v):
    return hex(v)


def pt1(p):
    return None if p is None else [hx(p[0]), hx(p[1])]


def pt2(p):
    return None if p is None else [[hx(p[0][0]), hx(p[0][1])], [hx(p[1][0]), hx(p[1][1])]]


def main():
    rnd = random.Random(0xBE11)
    out = {}
    # FFT: 8 points, all four transforms (domain.rs:81-125)
    vals = [rnd.randrange(Q) for _ in range(8)]
    fft = {"input": [hx(v) for v in vals]}
    for name in ("fft", "ifft", "coset_fft", "icoset_fft"):
        d = EvaluationDomain.from_coeffs(Bls12.Fr, vals)
        getattr(d, name)(Worker(8))
        fft[name] = [hx(v) for v in d.coeffs]
    out["fft8"] = fft
    # MSM: 24 scalars, density map, skip 2 (multiexp.rs:305-332)
    n = 24
    scalars = [rnd.randrange(Q) for _ in range(n)]
    scalars[1], scalars[2], scalars[3] = 0, 1, Q - 1
    density = [rnd.random() < 0.6 for _ in range(n)]
    nb = sum(density) + 2
    for gname, curve, ser in (("g1", bls.G1, pt1), ("g2", bls.G2, pt2)):
        ks = [rnd.randrange(1, Q) for _ in range(nb)]
        bases = [curve.mul(curve.gen, k) for k in ks]
        d = pm.DensityTracker()
        d.bv = list(density)
        res = pm.multiexp(Worker(), curve, Bls12.Fr, bases, 2, d, [pm.exponent_from(s) for s in scalars]).wait()
        full = pm.multiexp(Worker(), curve, Bls12.Fr, bases[:n] + [curve.gen] * max(0, n - nb), 0, pm.FullDensity(),
                           [pm.exponent_from(s) for s in scalars]).wait() if nb >= n else None
        out["msm_" + gname] = {"base_scalars": [hx(k) for k in ks], "scalars": [hx(s) for s in scalars],
                               "density": density, "skip": 2, "result": ser(res),
                               "result_full_density_first_n": ser(full) if full is not None else None}
    # a tiny Groth16 proof: 3-round MiMC, fixed toxic waste (generator.rs / prover.rs restated)
    cons = [rnd.randrange(Q) for _ in range(3)]
    xl, xr, r, s = (rnd.randrange(Q) for _ in range(4))
    toxic = dict(alpha=48577, beta=22580, gamma=53332, delta=5481, tau=3673)
    circ = circuits.mimc_circuit(xl, xr, cons)
    params = generate_parameters(Bls12, circ, bls.G1_GEN, bls.G2_GEN, **toxic)
    proof = create_proof(Bls12, circ, params, r, s)
    out["mimc3_proof"] = {"constants": [hx(c) for c in cons], "xl": hx(xl), "xr": hx(xr), "r": hx(r), "s": hx(s),
                          "toxic": toxic, "image": hx(circuits.mimc_hash(xl, xr, cons)),
                          "a": pt1(proof.a), "b": pt2(proof.b), "c": pt1(proof.c),
                          "proof_bytes_zcash": (bls.g1_compress(proof.a) + bls.g2_compress(proof.b) + bls.g1_compress(proof.c)).hex()}
    # the same circuit's serialized Parameters (groth16/src/lib.rs:258-287) and the constraint
    # evaluations / query densities the prover derives from the witness (prover.rs:19-55,105-145,208-215)
    from oracle.pyref import params_io as pio
    from oracle.pyref.core import INPUT, Variable
    from oracle.pyref.prover import ProvingAssignment

    vk = dict(alpha_g1=params.vk.alpha_g1, beta_g1=params.vk.beta_g1, beta_g2=params.vk.beta_g2, gamma_g2=params.vk.gamma_g2,
              delta_g1=params.vk.delta_g1, delta_g2=params.vk.delta_g2, ic=list(params.vk.ic))
    out["mimc3_proof"]["parameters_bytes"] = pio.parameters_write(vk, params.h, params.l, params.a, params.b_g1, params.b_g2).hex()
    pa = ProvingAssignment(Q)
    pa.alloc_input(lambda: 1)
    circ(pa)
    for i in range(len(pa.input_assignment)):
        pa.enforce(lambda lc, i=i: lc + Variable(INPUT, i), lambda lc: lc, lambda lc: lc)
    out["mimc3_proof"]["assignment"] = {
        "a": [hx(v) for v in pa.a], "b": [hx(v) for v in pa.b], "c": [hx(v) for v in pa.c],
        "inputs": [hx(v) for v in pa.input_assignment], "aux": [hx(v) for v in pa.aux_assignment],
        "a_aux_density": list(pa.a_aux_density.bv), "b_input_density": list(pa.b_input_density.bv),
        "b_aux_density": list(pa.b_aux_density.bv)}
    json.dump(out, open(os.path.join(HERE, "bls12_381_small.json"), "w"), indent=1)
    # the reference's own KAT (groth16/src/tests/mod.rs:91-373), as data
    kat = {
        "source": "/root/reference/groth16/src/tests/mod.rs:91-373 (test_xordemo over DummyEngine F_64513)",
        "modulus": 64513, "toxic": {"alpha": 48577, "beta": 22580, "gamma": 53332, "delta": 5481, "tau": 3673},
        "root_of_unity_2^3": 20201, "u_i": [59158, 48317, 21767, 10402], "v_i": [0, 0, 60619, 30791],
        "w_i": [0, 23320, 41193, 41193], "r": 27134, "s": 17146,
        "h_coefficients": [5040, 11763, 10755, 63633, 128, 9747, 8739],
    }
    json.dump(kat, open(os.path.join(HERE, "xordemo_kat.json"), "w"), indent=1)
    print("wrote golden fixtures")


if __name__ == "__main__":
    main()
