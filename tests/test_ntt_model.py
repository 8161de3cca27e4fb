"""The kernel's pass / tile / register-step decomposition (modelled in tests/models/ntt_model.py, which mirrors
bellman_amd/csrc/fft.hip statement by statement) equals the restated serial_fft (src/domain.rs:272-314) for
every pass count L = 1..3, every stage-group pattern (radix 8 / 4 / 2 steps), forward / inverse / coset
variants, with the tile shrunk so that the multi-pass plans are reachable at Python speed."""

import random

import pytest

from oracle.pyref.domain import EvaluationDomain
from oracle.pyref.engines import Bls12, DummyEngine
from oracle.pyref.multicore import Worker
from tests.models.ntt_model import ntt_model, plan_passes, step_groups


def _check(F, vals, log_n, log_tile, max_r, threads, names=("fft", "ifft", "coset_fft", "icoset_fft")):
    for name in names:
        d = EvaluationDomain.from_coeffs(F, vals)
        getattr(d, name)(Worker(1))
        inverse = name in ("ifft", "icoset_fft")
        kw = {}
        if name == "coset_fft":
            kw["pre_g"] = F.MULTIPLICATIVE_GENERATOR
        if name == "icoset_fft":
            kw["post_g"], kw["post_scale"] = d.geninv, d.minv
        if name == "ifft":
            kw["post_const"] = d.minv
        got = ntt_model(vals, F.r, d.omega, log_n, inverse=inverse, log_tile=log_tile, max_r=max_r, threads=threads, **kw)
        assert got == d.coeffs, (name, log_n, log_tile, max_r)


# (log_tile, max_r, threads): the toy field F_64513 has 2-adicity 10 (domains up to 2^9) and the tile has to shrink for
# two- and three-pass plans; threads < tasks exercises the task loop, threads > tasks the idle lanes
@pytest.mark.parametrize("cfg", [(4, 4, 2), (5, 4, 4), (5, 5, 64), (6, 3, 8), (4, 2, 4), (7, 7, 16)])
@pytest.mark.parametrize("log_n", range(0, 10))
def test_model_matches_serial_fft_toy_field(log_n, cfg):
    log_tile, max_r, threads = cfg
    if len(plan_passes(log_n, max_r)) > 3:
        pytest.skip("more than 3 passes")
    F = DummyEngine.Fr
    rnd = random.Random(log_n * 17 + log_tile)
    vals = [rnd.randrange(F.r) for _ in range(1 << log_n)]
    _check(F, vals, log_n, log_tile, max_r, threads)


@pytest.mark.parametrize("log_n", [11, 12, 13])
def test_model_real_tile_bls(log_n):
    """The real tile (2^11 elements, 256 threads, sub-FFTs up to 2^11) over BLS12-381 Fr: one pass at 2^11
    (steps 3+3+3+2), two passes at 2^12 (6+6: 3+3) and 2^13 (7+6: 3+2+2 and 3+3)."""
    F = Bls12.Fr
    rnd = random.Random(log_n)
    vals = [rnd.randrange(F.r) for _ in range(1 << log_n)]
    _check(F, vals, log_n, 11, 11, 256, names=("fft", "icoset_fft") if log_n > 11 else ("fft", "ifft", "coset_fft", "icoset_fft"))


@pytest.mark.parametrize("log_n,threads,gmax", [(11, 512, 2), (12, 512, 2), (13, 512, 2), (11, 256, 3), (13, 256, 3)])
def test_model_real_tile_wave_local_steps(log_n, threads, gmax):
    """[r5] The 512-thread radix-4 kernel of the one-level tables (and the 256-thread radix-8 kernel) drop the s_barrier
    between register steps that stay inside a wavefront's own block of the tile: the model runs the same task -> thread
    mapping and asserts, for every pair of steps the kernel's rule leaves without a barrier, that each wavefront touches
    exactly the positions it touched in the step before - and that the transform is still the serial FFT."""
    F = Bls12.Fr
    rnd = random.Random(100 + log_n)
    vals = [rnd.randrange(F.r) for _ in range(1 << log_n)]
    d = EvaluationDomain.from_coeffs(F, vals)
    d.fft(Worker(1))
    skipped = []
    got = ntt_model(vals, F.r, d.omega, log_n, log_tile=11, max_r=11, threads=threads, gmax=gmax, barriers_skipped=skipped)
    assert got == d.coeffs
    assert skipped, "no barrier was dropped at all"
    if (log_n, threads) == (11, 512):
        assert sorted(set(skipped)) == [(11, 2), (11, 4), (11, 6)]     # steps at s = 0, 2, 4, 6 share one barrier
    if (log_n, threads) == (11, 256):
        assert sorted(set(skipped)) == [(11, 3), (11, 6)]


def test_plan_shapes():
    assert step_groups(11, 2) == [2, 2, 2, 2, 2, 1] and step_groups(8, 2) == [2, 2, 2, 2] and step_groups(7, 2) == [2, 2, 2, 1]
    assert plan_passes(10) == [10] and plan_passes(11) == [11]
    assert plan_passes(12) == [6, 6]
    assert plan_passes(20) == [10, 10]
    assert plan_passes(21) == [11, 10]
    assert plan_passes(22) == [11, 11]            # two passes up to 2^22
    assert plan_passes(23) == [8, 8, 7]
    assert plan_passes(24) == [8, 8, 8]
    assert plan_passes(31) == [11, 10, 10]
    assert step_groups(11) == [3, 3, 3, 2] and step_groups(10) == [3, 3, 2, 2] and step_groups(8) == [3, 3, 2]
    assert step_groups(7) == [3, 2, 2] and step_groups(4) == [2, 2] and step_groups(1) == [1] and step_groups(0) == []
