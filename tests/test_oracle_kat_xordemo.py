"""Pins the pure-Python oracle to the reference's only known-answer test,
`test_xordemo` (/root/reference/groth16/src/tests/mod.rs:91-373), over the
toy DummyEngine (F_64513).  Every literal below is copied from that test's
assertions (they are DATA the reference asserts, not code)."""

from oracle.pyref.core import ConstraintSystem
from oracle.pyref.engines import DummyEngine
from oracle.pyref.generator import generate_parameters
from oracle.pyref.prover import create_proof

R = 64513


def xordemo(a, b):
    """groth16/src/tests/mod.rs:20-89"""

    def synth(cs):
        a_var = cs.alloc(lambda: 1 if a else 0)
        cs.enforce(lambda lc: lc + cs.one() - a_var, lambda lc: lc + a_var, lambda lc: lc)
        b_var = cs.alloc(lambda: 1 if b else 0)
        cs.enforce(lambda lc: lc + cs.one() - b_var, lambda lc: lc + b_var, lambda lc: lc)
        c_var = cs.alloc_input(lambda: 1 if (a ^ b) else 0)
        cs.enforce(
            lambda lc: lc + a_var + a_var,
            lambda lc: lc + b_var,
            lambda lc: lc + a_var + b_var - c_var,
        )

    return synth


ALPHA, BETA, GAMMA, DELTA, TAU = 48577, 22580, 53332, 5481, 3673


def _params():
    return generate_parameters(
        DummyEngine, xordemo(False, False), 1, 1, ALPHA, BETA, GAMMA, DELTA, TAU
    )


def test_root_of_unity():
    # mod.rs:126-134
    w = DummyEngine.Fr.ROOT_OF_UNITY
    assert pow(w, 1 << 10, R) == 1
    w8 = pow(w, 1 << 7, R)
    assert pow(w8, 8, R) == 1
    assert w8 == 20201


def test_parameters_match_reference_kat():
    p = _params()
    assert len(p.h) == 7  # mod.rs:122
    t_at_tau = (pow(TAU, 8, R) - 1) % R
    dinv = pow(DELTA, R - 2, R)
    ginv = pow(GAMMA, R - 2, R)
    coeff = dinv * t_at_tau % R
    cur = 1
    for h in p.h:  # mod.rs:159-174
        assert h == cur * coeff % R
        cur = cur * TAU % R
    assert len(p.vk.ic) == 2 and len(p.l) == 2 and len(p.a) == 4  # mod.rs:177-184
    assert len(p.b_g1) == 2 and len(p.b_g2) == 2
    u_i = [59158, 48317, 21767, 10402]  # mod.rs:213-224
    v_i = [0, 0, 60619, 30791]
    w_i = [0, 23320, 41193, 41193]
    assert p.a == u_i
    assert p.b_g1 == [v for v in v_i if v]
    assert p.b_g2 == [v for v in v_i if v]
    for i in range(4):  # mod.rs:238-259
        t = (BETA * u_i[i] + ALPHA * v_i[i] + w_i[i]) % R
        if i < 2:
            assert p.vk.ic[i] == t * ginv % R
        else:
            assert p.l[i - 2] == t * dinv % R
    assert (p.vk.alpha_g1, p.vk.beta_g1, p.vk.beta_g2) == (ALPHA, BETA, BETA)
    assert (p.vk.gamma_g2, p.vk.delta_g1, p.vk.delta_g2) == (GAMMA, DELTA, DELTA)


def test_proof_matches_reference_kat():
    p = _params()
    r, s = 27134, 17146  # mod.rs:274-275
    trace = {}
    proof = create_proof(DummyEngine, xordemo(True, False), p, r, s, trace=trace)
    u_i = [59158, 48317, 21767, 10402]
    v_i = [0, 0, 60619, 30791]
    # H-polynomial coefficients, mod.rs:358
    assert trace["h_coeffs"] == [5040, 11763, 10755, 63633, 128, 9747, 8739]
    exp_a = (DELTA * r + ALPHA + u_i[0] + u_i[1] + u_i[2]) % R  # mod.rs:294-303
    assert proof.a == exp_a
    exp_b = (DELTA * s + BETA + v_i[0] + v_i[1] + v_i[2]) % R  # mod.rs:311-320
    assert proof.b == exp_b
    exp_c = (proof.a * s + proof.b * r - DELTA * r * s + p.l[0]) % R  # mod.rs:336-369
    for i, co in enumerate([5040, 11763, 10755, 63633, 128, 9747, 8739]):
        exp_c = (exp_c + p.h[i] * co) % R
    assert proof.c == exp_c
    # verify_proof for the dummy pairing e(a,b) = a*b  (verifier.rs:23-58)
    acc = (p.vk.ic[0] + p.vk.ic[1] * 1) % R
    lhs = proof.a * proof.b % R
    rhs = (ALPHA * BETA + acc * GAMMA + proof.c * DELTA) % R
    assert lhs == rhs


def test_zero_coeff_does_not_count_towards_density():
    """mod.rs:375-440 regression (prover.rs:31)."""

    def circuit(one_var):
        def synth(cs):
            a = cs.alloc(lambda: 10)
            b = cs.alloc(lambda: 9)
            product = cs.alloc(lambda: 90)
            if one_var:
                cs.enforce(lambda lc: lc + (0, cs.one()) + a, lambda lc: lc + b, lambda lc: lc + product)
            else:
                cs.enforce(lambda lc: lc + a, lambda lc: lc + (0, product) + b, lambda lc: lc + product)

        return synth

    for one_var in (True, False):
        p = generate_parameters(DummyEngine, circuit(one_var), 1, 1, ALPHA, BETA, GAMMA, DELTA, TAU)
        proof = create_proof(DummyEngine, circuit(one_var), p, 27134, 17146)
        acc = p.vk.ic[0]
        assert proof.a * proof.b % R == (ALPHA * BETA + acc * GAMMA + proof.c * DELTA) % R
