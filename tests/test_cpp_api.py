"""The C++ mirror of bellman's API (bellman_amd/csrc/groth16.hpp) used from a standalone C++ program
with its own Circuit (tests/cpp/prove_cubic.cpp): builds everywhere; on a GPU it proves and the proof is
compared with the oracle's and with the Python mirror's."""

import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "prove_cubic.bin")
Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def _build():
    from bellman_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])


def cubic_circuit(x):
    def synth(cs):
        x2v, x3v = x * x % Q, x * x % Q * x % Q
        outv = (x3v + x + 5) % Q
        xv = cs.alloc(lambda: x)
        x2 = cs.alloc(lambda: x2v)
        x3 = cs.alloc(lambda: x3v)
        out = cs.alloc_input(lambda: outv)
        cs.enforce(lambda lc: lc + xv, lambda lc: lc + xv, lambda lc: lc + x2)
        cs.enforce(lambda lc: lc + x2, lambda lc: lc + xv, lambda lc: lc + x3)
        cs.enforce(lambda lc: lc + x3 + xv + (5, cs.one()), lambda lc: lc + cs.one(), lambda lc: lc + out)

    return synth


def test_cpp_example_builds_and_refuses_without_gpu(tmp_path):
    import torch

    _build()
    assert os.path.exists(BIN)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    # a well-formed (dummy) input file: the program must stop at bh_ctx_create, not fall back to a CPU path
    f = tmp_path / "in.bin"
    with open(f, "wb") as fh:
        fh.write(bytes(96 + 96 + 192 + 96 + 192))
        for _ in range(5):
            fh.write(struct.pack("<Q", 0))
        fh.write(bytes(96))
    r = subprocess.run([BIN, str(f)], capture_output=True, text=True)
    assert r.returncode == 4 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_cpp_api_proof_matches_oracle_and_python_mirror(tmp_path):
    import bellman_amd
    from bellman_amd import groth16 as pg
    from oracle import cref
    from oracle.cengine import CBls12
    from oracle.pyref.generator import generate_parameters
    from oracle.pyref.prover import create_proof as oracle_create_proof

    _build()
    x, r, s = 0x1234567890ABCDEF, 0x1111222233334444, 0x5555666677778888
    toxic = dict(alpha=48577, beta=22580, gamma=53332, delta=5481, tau=3673)
    circ = cubic_circuit(x)
    p = generate_parameters(CBls12, circ, CBls12.G1.gen, CBls12.G2.gen, **toxic)
    want = oracle_create_proof(CBls12, circ, p, r, s)
    G1, G2 = CBls12.G1, CBls12.G2
    f = tmp_path / "in.bin"
    with open(f, "wb") as fh:
        for k in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2"):
            fh.write(getattr(p.vk, k))
        for arr in (G1.to_array(p.h), G1.to_array(p.l), G1.to_array(p.a), G1.to_array(p.b_g1), G2.to_array(p.b_g2)):
            fh.write(struct.pack("<Q", arr.shape[0]))
            fh.write(arr.tobytes())
        fh.write(pg.fr_to_mont_array([x, r, s]).tobytes())
    out = subprocess.run([BIN, str(f)], capture_output=True, text=True, check=True).stdout.strip()
    got = bytes.fromhex(out)
    assert got == want.a + want.b + want.c
    # and the Python mirror agrees
    w = bellman_amd.Worker(0)
    pp = pg.Parameters(w, *(np.frombuffer(getattr(p.vk, k), dtype=np.uint64) for k in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2")),
                       G1.to_array(p.h), G1.to_array(p.l), G1.to_array(p.a), G1.to_array(p.b_g1), G2.to_array(p.b_g2))
    pr = pg.create_proof(circ, pp, r, s)
    assert pr.a.tobytes() + pr.b.tobytes() + pr.c.tobytes() == got
    pp.release()
    w.close()
