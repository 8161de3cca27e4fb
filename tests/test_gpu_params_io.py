"""GPU parity of `Parameters::read(reader, checked)` (groth16/src/lib.rs:289-398, with
VerifyingKey::read :159-215): decoding and point validation on the device against the oracle's
restatement - same points out, same error (and same first offending point) for every rule."""

import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cref  # noqa: E402
from oracle.cengine import CBls12  # noqa: E402
from oracle.pyref import bls12_381 as bls  # noqa: E402
from oracle.pyref import params_io as pio  # noqa: E402
from oracle.pyref.generator import generate_parameters  # noqa: E402
from oracle.pyref.prover import create_proof as oracle_create_proof  # noqa: E402
from tests import circuits, pointgen  # noqa: E402
from tests.test_gpu_groth16 import TOXIC, _same, worker  # noqa: E402,F401

Q = bls.Q


def _enc(group, pt):
    return bls.g1_uncompressed(pt) if group == 1 else bls.g2_uncompressed(pt)


def _expected_error(group, blobs, checked, forbid_identity):
    """what the reference's read loop reports for this stream of points: (exception type, index) or None"""
    for i, b in enumerate(blobs):
        try:
            p = pio.from_uncompressed(group, b, checked)
        except pio.InvalidPoint:
            return pio.InvalidPoint, i
        if p is None and forbid_identity:
            return pio.PointAtInfinity, i
    return None


@pytest.mark.parametrize("group", [1, 2])
def test_read_uncompressed_valid_points(worker, group):
    import bellman_amd

    n = 300
    arr = cref.gen_bases(group, n, a=3, b=5)
    pts = cref.g1_to_py(arr) if group == 1 else cref.g2_to_py(arr)
    pts[17] = None   # an identity, allowed here
    blob = b"".join(_enc(group, p) for p in pts)
    want = cref.g1_from_py(pts) if group == 1 else cref.g2_from_py(pts)
    for checked in (False, True):
        b = bellman_amd.Bases.read_uncompressed(worker, group, blob, checked=checked, forbid_identity=False)
        assert (b.download() == want).all()
        assert (b.download(10, 20) == want[10:30]).all()
    with pytest.raises(bellman_amd.PointAtInfinity) as e:
        bellman_amd.Bases.read_uncompressed(worker, group, blob, checked=True, forbid_identity=True)
    assert e.value.index == 17
    assert len(bellman_amd.Bases.read_uncompressed(worker, group, b"", checked=True)) == 0


@pytest.mark.parametrize("group", [1, 2])
def test_read_uncompressed_rules_match_oracle(worker, group):
    import bellman_amd

    curve = bls.G1 if group == 1 else bls.G2
    rec = 96 if group == 1 else 192
    good = [curve.mul(curve.gen, k) for k in (2, 3, 5, 7, 11, 13)]
    tors = pointgen.g1_on_curve_not_in_subgroup(9) if group == 1 else pointgen.g2_on_curve_not_in_subgroup(9)
    if group == 1:
        off = (good[0][0], (good[0][1] + 1) % bls.P)
    else:
        off = (good[0][0], ((good[0][1][0] + 1) % bls.P, good[0][1][1]))
    g = _enc(group, good[1])
    mutations = {
        "compressed flag": bytes([g[0] | 0x80]) + g[1:],
        "sort flag": bytes([g[0] | 0x20]) + g[1:],
        "infinity flag + coordinates": bytes([g[0] | 0x40]) + g[1:],
        "infinity flag + one low bit": bytes([0x40]) + bytes(rec - 2) + b"\x01",
        "infinity + sort": bytes([0x60]) + bytes(rec - 1),
        "x not canonical": bls.P.to_bytes(48, "big") + g[48:],
        "last coordinate not canonical": g[:-48] + (bls.P + 1).to_bytes(48, "big"),
        "all ones": bytes([0x1F]) + b"\xff" * (rec - 1),
        "off curve": _enc(group, off),
        "on curve, wrong subgroup": _enc(group, tors),
        "identity": _enc(group, None),
        "valid": g,
    }
    ok = [_enc(group, p) for p in good]
    for name, bad in mutations.items():
        for pos in (0, 3, 5):
            blobs = list(ok)
            blobs[pos] = bad
            for checked in (False, True):
                for forbid in (False, True):
                    want = _expected_error(group, blobs, checked, forbid)
                    try:
                        b = bellman_amd.Bases.read_uncompressed(worker, group, b"".join(blobs), checked=checked,
                                                                forbid_identity=forbid)
                        got = None
                    except bellman_amd.InvalidPoint as e:
                        got = (pio.InvalidPoint, e.index)
                    except bellman_amd.PointAtInfinity as e:
                        got = (pio.PointAtInfinity, e.index)
                    assert got == want, (name, pos, checked, forbid)
                    if got is None:
                        pts = [pio.from_uncompressed(group, x, False) for x in blobs]
                        ref = cref.g1_from_py(pts) if group == 1 else cref.g2_from_py(pts)
                        assert (b.download() == ref).all(), name
    # two bad points: the first in stream order is the one reported
    blobs = list(ok)
    blobs[4] = mutations["sort flag"]
    blobs[2] = mutations["identity"]
    with pytest.raises(bellman_amd.PointAtInfinity) as e:
        bellman_amd.Bases.read_uncompressed(worker, group, b"".join(blobs), checked=True)
    assert e.value.index == 2


def test_subgroup_check_at_scale(worker):
    """2^14 valid G1 and 2^12 valid G2 points pass `checked`; one planted torsion point is found."""
    import bellman_amd

    for group, n in ((1, 1 << 14), (2, 1 << 12)):
        arr = cref.gen_bases(group, n, a=7, b=9)
        canon = cref.fp_from_mont(arr.reshape(-1, 6)).reshape(n, -1)
        be = canon[:, :].reshape(n, -1, 6)[:, :, ::-1].astype(">u8").tobytes()   # limbs MSB first, big-endian bytes
        if group == 2:   # c1 before c0 on the wire
            raw = np.frombuffer(be, dtype=np.uint8).reshape(n, 4, 48)[:, [1, 0, 3, 2], :]
            be = raw.tobytes()
        b = bellman_amd.Bases.read_uncompressed(worker, group, be, checked=True)
        assert (b.download() == arr.reshape(n, -1)).all()
        tors = pointgen.g1_on_curve_not_in_subgroup(77) if group == 1 else pointgen.g2_on_curve_not_in_subgroup(77)
        rec = 96 if group == 1 else 192
        k = n - 5
        planted = be[:k * rec] + _enc(group, tors) + be[(k + 1) * rec:]
        with pytest.raises(bellman_amd.InvalidPoint) as e:
            bellman_amd.Bases.read_uncompressed(worker, group, planted, checked=True)
        assert e.value.index == k
        assert len(bellman_amd.Bases.read_uncompressed(worker, group, planted, checked=False)) == n


def _py(group, recs):
    return (cref.g1_to_py if group == 1 else cref.g2_to_py)(np.frombuffer(b"".join(bytes(r) for r in recs), dtype=np.uint64))


def test_parameters_read_then_prove_mimc(worker):
    """serialization test of the reference (lib.rs:486-567) turned into parity: write the CRS with the
    oracle, read it with the product (checked and unchecked), prove, compare with the oracle's proof
    and its Proof::write bytes."""
    from bellman_amd import groth16 as pg

    rounds = 40
    rnd = random.Random(99)
    cons = [rnd.randrange(Q) for _ in range(rounds)]
    xl, xr, r, s = (rnd.randrange(Q) for _ in range(4))
    p = generate_parameters(CBls12, circuits.mimc_circuit(0, 0, cons), CBls12.G1.gen, CBls12.G2.gen, **TOXIC)
    vk = dict(alpha_g1=_py(1, [p.vk.alpha_g1])[0], beta_g1=_py(1, [p.vk.beta_g1])[0], beta_g2=_py(2, [p.vk.beta_g2])[0],
              gamma_g2=_py(2, [p.vk.gamma_g2])[0], delta_g1=_py(1, [p.vk.delta_g1])[0], delta_g2=_py(2, [p.vk.delta_g2])[0],
              ic=_py(1, p.vk.ic))
    qs = dict(h=_py(1, p.h), l=_py(1, p.l), a=_py(1, p.a), b_g1=_py(1, p.b_g1), b_g2=_py(2, p.b_g2))
    blob = pio.parameters_write(vk, qs["h"], qs["l"], qs["a"], qs["b_g1"], qs["b_g2"])
    circ = circuits.mimc_circuit(xl, xr, cons)
    want = oracle_create_proof(CBls12, circ, p, r, s)
    for checked in (True, False):
        pp = pg.Parameters.read(worker, blob + b"trailing bytes are ignored", checked)
        for name in ("h", "l", "a", "b_g1"):
            assert (pp.query(name) == cref.g1_from_py(qs[name])).all()
        assert (pp.query("b_g2") == cref.g2_from_py(qs["b_g2"])).all()
        got_vk = pp.vk()
        assert got_vk[0].tobytes() == bytes(p.vk.alpha_g1) and got_vk[4].tobytes() == bytes(p.vk.delta_g2)
        got = pg.create_proof(circ, pp, r, s)
        assert _same(got, want.a, want.b, want.c)
        a_pt, b_pt, c_pt = _py(1, [want.a])[0], _py(2, [want.b])[0], _py(1, [want.c])[0]
        assert got.write() == pio.proof_write(a_pt, b_pt, c_pt)
        pp.release()


def test_parameters_read_errors_match_oracle(worker):
    import bellman_amd
    from bellman_amd import groth16 as pg

    vk, h, l, a, b1, b2 = pointgen.small_parameters()
    good = pio.parameters_write(vk, h, l, a, b1, b2)
    tors1 = pointgen.g1_on_curve_not_in_subgroup(21)
    tors2 = pointgen.g2_on_curve_not_in_subgroup(21)
    off_a = list(a)
    off_a[2] = (a[2][0], (a[2][1] + 1) % bls.P)
    l_inf = list(l)
    l_inf[3] = None
    h_flag = bytearray(good)
    h_flag[864 + 4 + 2 * 96 + 4 + 96] |= 0x20   # second h point: sort flag
    cases = {
        "good": good,
        "cut by one byte": good[:-1],
        "cut inside vk": good[:500],
        "cut inside ic count": good[:866],
        "cut inside a count": good[:864 + 4 + 192 + 4 + 288 + 4 + 384 + 2],
        "empty": b"",
        "torsion point in b_g1": pio.parameters_write(vk, h, l, a, [b1[0], tors1, b1[2]], b2),
        "torsion point in b_g2": pio.parameters_write(vk, h, l, a, b1, [b2[0], b2[1], tors2]),
        "off-curve point in a": pio.parameters_write(vk, h, l, off_a, b1, b2),
        "identity in l": pio.parameters_write(vk, h, l_inf, a, b1, b2),
        "identity in ic": pio.parameters_write(dict(vk, ic=[None, vk["ic"][1]]), h, l, a, b1, b2),
        "identity delta_g1 (allowed by read)": pio.parameters_write(dict(vk, delta_g1=None), h, l, a, b1, b2),
        "torsion alpha_g1 (vk is always checked)": pio.parameters_write(dict(vk, alpha_g1=tors1), h, l, a, b1, b2),
        "torsion gamma_g2": pio.parameters_write(dict(vk, gamma_g2=tors2), h, l, a, b1, b2),
        "sort flag in h + cut": bytes(h_flag[:-7]),
        "identity in l + torsion later": pio.parameters_write(vk, h, l_inf, a, [tors1], b2),
        "huge count": good[:864 + 4 + 192] + b"\xff\xff\xff\xff" + good[864 + 4 + 192 + 4:],
    }
    kinds = {pio.UnexpectedEof: bellman_amd.UnexpectedEof, pio.InvalidPoint: bellman_amd.InvalidPoint,
             pio.PointAtInfinity: bellman_amd.PointAtInfinity}
    outcomes = set()
    for name, blob in cases.items():
        for checked in (True, False):
            try:
                pio.parameters_read(blob, checked)
                want = None
            except pio.IoError as e:
                want = kinds[type(e)]
            try:
                pp = pg.Parameters.read(worker, blob, checked)
                pp.release()
                got = None
            except (bellman_amd.UnexpectedEof, bellman_amd.InvalidData) as e:
                got = type(e)
            assert got == want, (name, checked, got, want)
            outcomes.add(want)
    assert outcomes == {None, bellman_amd.UnexpectedEof, bellman_amd.InvalidPoint, bellman_amd.PointAtInfinity}
