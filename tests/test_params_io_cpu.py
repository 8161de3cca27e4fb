"""CPU side of the byte formats around the proving path (groth16/src/lib.rs:38-46, 143-215, 258-398):
the oracle restatement's rules, and the product's host-side Proof::write against it."""

import random

import numpy as np
import pytest

from oracle import cref
from oracle.pyref import bls12_381 as bls
from oracle.pyref import params_io as pio
from tests import pointgen


def test_oracle_parameters_round_trip_and_lengths():
    vk, h, l, a, b1, b2 = pointgen.small_parameters()
    blob = pio.parameters_write(vk, h, l, a, b1, b2)
    # vk: 3*96 + 3*192 + 4 + 2*96; then 4 + n*rec per query (the reference's 2136-byte check, lib.rs:529, same rule)
    assert len(blob) == 864 + 4 + 2 * 96 + (4 + 3 * 96) + (4 + 4 * 96) + (4 + 5 * 96) + (4 + 3 * 96) + (4 + 3 * 192)
    for checked in (True, False):
        got = pio.parameters_read(blob, checked)
        assert got["vk"] == vk and (got["h"], got["l"], got["a"], got["b_g1"], got["b_g2"]) == (h, l, a, b1, b2)


def test_oracle_point_rules():
    g = bls.g1_uncompressed(bls.G1.gen)
    assert pio.from_uncompressed(1, g, True) == bls.G1.gen
    assert pio.from_uncompressed(1, bls.g1_uncompressed(None), True) is None
    for bit in (0x80, 0x20):
        with pytest.raises(pio.InvalidPoint):
            pio.from_uncompressed(1, bytes([g[0] | bit]) + g[1:], False)
    with pytest.raises(pio.InvalidPoint):   # infinity flag with a coordinate
        pio.from_uncompressed(1, bytes([g[0] | 0x40]) + g[1:], False)
    with pytest.raises(pio.InvalidPoint):   # x = p is not canonical
        pio.from_uncompressed(1, bls.P.to_bytes(48, "big") + g[48:], False)
    off = (bls.G1.gen[0], (bls.G1.gen[1] + 1) % bls.P)
    assert pio.from_uncompressed(1, bls.g1_uncompressed(off), False) == off   # unchecked accepts it
    with pytest.raises(pio.InvalidPoint):
        pio.from_uncompressed(1, bls.g1_uncompressed(off), True)
    tors = pointgen.g1_on_curve_not_in_subgroup(3)
    assert bls.G1.on_curve(tors)
    with pytest.raises(pio.InvalidPoint):
        pio.from_uncompressed(1, bls.g1_uncompressed(tors), True)
    tors2 = pointgen.g2_on_curve_not_in_subgroup(5)
    assert bls.G2.on_curve(tors2)
    assert pio.from_uncompressed(2, bls.g2_uncompressed(tors2), False) == tors2
    with pytest.raises(pio.InvalidPoint):
        pio.from_uncompressed(2, bls.g2_uncompressed(tors2), True)


def test_oracle_read_error_order():
    vk, h, l, a, b1, b2 = pointgen.small_parameters()
    blob = bytearray(pio.parameters_write(vk, h, l, a, b1, b2))
    with pytest.raises(pio.UnexpectedEof):
        pio.parameters_read(bytes(blob[:-1]), False)
    with pytest.raises(pio.UnexpectedEof):
        pio.parameters_read(bytes(blob[:866]), False)   # inside the ic count
    bad = bytearray(blob)
    bad[864 + 4 + 2 * 96 + 4] |= 0x80                   # first h point: compression flag
    with pytest.raises(pio.InvalidPoint):                # an earlier bad point wins over the truncation
        pio.parameters_read(bytes(bad[:-1]), False)
    l_with_inf = list(l)
    l_with_inf[1] = None
    with pytest.raises(pio.PointAtInfinity):
        pio.parameters_read(pio.parameters_write(vk, h, l_with_inf, a, b1, b2), False)
    vk_inf = dict(vk, ic=[vk["ic"][0], None])
    with pytest.raises(pio.PointAtInfinity):
        pio.parameters_read(pio.parameters_write(vk_inf, h, l, a, b1, b2), False)
    vk_ok_inf = dict(vk, gamma_g2=None)                  # identity allowed in the fixed vk slots
    assert pio.parameters_read(pio.parameters_write(vk_ok_inf, h, l, a, b1, b2), True)["vk"]["gamma_g2"] is None


def _rand_points(rnd, n):
    g1 = [bls.G1.mul(bls.G1.gen, rnd.randrange(1, bls.Q)) for _ in range(n)]
    g2 = [bls.G2.mul(bls.G2.gen, rnd.randrange(1, bls.Q)) for _ in range(n)]
    return g1, g2


def test_product_proof_write_matches_oracle():
    """bh_proof_write is host code (Montgomery -> canonical, sort flag): runs without a GPU."""
    from bellman_amd import groth16 as pg

    rnd = random.Random(17)
    g1, g2 = _rand_points(rnd, 6)
    # force both values of every sort flag and the c1 == 0 branch of the Fp2 ordering
    cases = [(g1[i], g2[i], g1[(i + 1) % 6]) for i in range(6)]
    cases += [(bls.G1.neg(a), bls.G2.neg(b), bls.G1.neg(c)) for a, b, c in cases[:3]]
    cases += [(None, None, None), (g1[0], None, None), (None, g2[0], g1[1])]
    seen_flags = set()
    for a, b, c in cases:
        raw = np.concatenate([cref.g1_from_py([a])[0], cref.g2_from_py([b])[0], cref.g1_from_py([c])[0]])
        got = pg.Proof(raw).write()
        want = pio.proof_write(a, b, c)
        assert got == want and len(got) == 192   # lib.rs:559
        seen_flags.add((got[0] >> 5, got[48] >> 5))
    assert len(seen_flags) >= 4
    # Fp2 ordering with c1 == 0: synthetic (not on the curve - the encoder does not care)
    for y0 in (1, bls.P - 1):
        b = ((5, 6), (y0, 0))
        raw = np.concatenate([cref.g1_from_py([g1[0]])[0], cref.g2_from_py([b])[0], cref.g1_from_py([g1[1]])[0]])
        assert pg.Proof(raw).write() == pio.proof_write(g1[0], b, g1[1])
