"""The kernel's pass/tile decomposition (modelled in tests/models/ntt_model.py, which mirrors
bellman_amd/csrc/fft.hip statement by statement) equals the restated serial_fft
(src/domain.rs:272-314) for every pass count L = 1..4, forward / inverse / coset variants."""

import random

import pytest

from oracle.pyref.domain import EvaluationDomain
from oracle.pyref.engines import Bls12, DummyEngine
from oracle.pyref.multicore import Worker
from tests.models.ntt_model import ntt_model, plan_passes


@pytest.mark.parametrize("log_n", range(0, 10))
@pytest.mark.parametrize("cfg", [(10, 8), (3, 2), (4, 3), (5, 3)])
def test_model_matches_serial_fft_toy_field(log_n, cfg):
    log_tile, max_r = cfg
    if len(plan_passes(log_n, log_tile, max_r)) > 4:
        pytest.skip("more than 4 passes")
    F = DummyEngine.Fr
    rnd = random.Random(log_n * 17 + log_tile)
    vals = [rnd.randrange(F.r) for _ in range(1 << log_n)]
    for name, inverse in [("fft", False), ("ifft", True), ("coset_fft", False), ("icoset_fft", True)]:
        d = EvaluationDomain.from_coeffs(F, vals)
        getattr(d, name)(Worker(1))
        n = 1 << log_n
        g, ginv, minv = F.MULTIPLICATIVE_GENERATOR, d.geninv, d.minv
        pre = [pow(g, i, F.r) for i in range(n)] if name == "coset_fft" else None
        post = [pow(ginv, i, F.r) * minv % F.r for i in range(n)] if name == "icoset_fft" else None
        pc = minv if name == "ifft" else None
        got = ntt_model(vals, F.r, d.omega, log_n, inverse=inverse, pre=pre, post=post, post_const=pc,
                        log_tile=log_tile, max_r=max_r)
        assert got == d.coeffs, (name, log_n, cfg)


def test_model_real_plan_bls_2_12():
    """One real-plan (tile 2^10, radix <= 2^8) two-pass case over BLS12-381 Fr."""
    F = Bls12.Fr
    rnd = random.Random(1)
    log_n = 11
    vals = [rnd.randrange(F.r) for _ in range(1 << log_n)]
    d = EvaluationDomain.from_coeffs(F, vals)
    d.fft(Worker(1))
    assert plan_passes(log_n) == [6, 5]
    assert ntt_model(vals, F.r, d.omega, log_n) == d.coeffs


def test_plan_shapes():
    assert plan_passes(10) == [10]
    assert plan_passes(20) == [7, 7, 6]
    assert plan_passes(22) == [8, 7, 7]
    assert plan_passes(24) == [8, 8, 8]
    assert plan_passes(26) == [7, 7, 6, 6]
    assert plan_passes(31) == [8, 8, 8, 7]
