"""Byte formats on either side of the proving path, restated on Python integers (TEST INFRASTRUCTURE).

Follows /root/reference/groth16/src/lib.rs:
  :38-46    Proof::write        compressed A (48) | B (96) | C (48)
  :143-156  VerifyingKey::write uncompressed alpha_g1, beta_g1, beta_g2, gamma_g2, delta_g1, delta_g2,
                                u32 BE ic count, ic points
  :159-215  VerifyingKey::read  every point through from_uncompressed (always checked); ic points must
                                not be the identity
  :258-287  Parameters::write   vk, then u32 BE count + uncompressed points for h, l, a, b_g1, b_g2
  :289-398  Parameters::read    `checked` selects from_uncompressed / from_uncompressed_unchecked for the
                                query points; every query point must not be the identity

The point decoding itself lives in the third-party crate `bls12_381 0.8.0` (Cargo.lock:105-108, source
absent from /root/reference): **parity unpinned** beyond the reference's own round-trip test
(groth16/src/lib.rs:486-567).  The published rules of that encoding (Zcash BLS12-381 serialisation)
are restated here:
  byte 0 bit 7 = compressed, bit 6 = infinity, bit 5 = sort (y lexicographically largest);
  uncompressed: compressed and sort flags must be clear; if infinity is set every other bit must be
  zero; coordinates are big-endian canonical (< p), G2 sends c1 before c0;
  from_uncompressed additionally requires the point to be on the curve and in the prime-order subgroup.
Errors are the reference's io::Error kinds; the integer codes are the C ABI's.
"""

from . import bls12_381 as bls


class IoError(Exception):
    code = -1


class UnexpectedEof(IoError):
    """read_exact / read_u32 ran out of bytes (io::ErrorKind::UnexpectedEof)"""

    code = 2


class InvalidPoint(IoError):
    """io::ErrorKind::InvalidData, "invalid G1" / "invalid G2" (lib.rs:300-304, 326-330, 169-173, 181-185)"""

    code = 6


class PointAtInfinity(IoError):
    """io::ErrorKind::InvalidData, "point at infinity" (lib.rs:306-315, 332-341, 199-207)"""

    code = 7


def _fp_from_bytes(b):
    v = int.from_bytes(b, "big")
    return v if v < bls.P else None


def from_uncompressed(group, data, checked):
    """-> affine point (None = identity); raises InvalidPoint.  `data` is 96 (G1) / 192 (G2) bytes."""
    flags = data[0]
    body = bytes([data[0] & 0x1F]) + bytes(data[1:])
    n = 2 if group == 1 else 4
    coords = [_fp_from_bytes(body[48 * i:48 * i + 48]) for i in range(n)]
    if any(c is None for c in coords):
        raise InvalidPoint("non-canonical coordinate")
    if flags & 0x80 or flags & 0x20:
        raise InvalidPoint("compressed / sort flag on an uncompressed point")
    if flags & 0x40:
        if any(coords):
            raise InvalidPoint("infinity flag with non-zero coordinates")
        return None
    if group == 1:
        pt = (coords[0], coords[1])
        curve = bls.G1
    else:
        pt = ((coords[1], coords[0]), (coords[3], coords[2]))   # c1 travels first
        curve = bls.G2
    if checked:
        if not curve.on_curve(pt):
            raise InvalidPoint("not on the curve")
        if curve.mul(pt, bls.Q) is not None:
            raise InvalidPoint("not in the prime-order subgroup")
    return pt


def _enc(group, pt):
    return bls.g1_uncompressed(pt) if group == 1 else bls.g2_uncompressed(pt)


def vk_write(vk):
    """vk: dict alpha_g1, beta_g1, beta_g2, gamma_g2, delta_g1, delta_g2, ic (list)   lib.rs:143-156"""
    out = _enc(1, vk["alpha_g1"]) + _enc(1, vk["beta_g1"]) + _enc(2, vk["beta_g2"]) + _enc(2, vk["gamma_g2"])
    out += _enc(1, vk["delta_g1"]) + _enc(2, vk["delta_g2"])
    out += len(vk["ic"]).to_bytes(4, "big")
    for p in vk["ic"]:
        out += _enc(1, p)
    return out


def parameters_write(vk, h, l, a, b_g1, b_g2):
    """lib.rs:258-287"""
    out = bytearray(vk_write(vk))
    for group, q in ((1, h), (1, l), (1, a), (1, b_g1), (2, b_g2)):
        out += len(q).to_bytes(4, "big")
        for p in q:
            out += _enc(group, p)
    return bytes(out)


class _Reader:
    def __init__(self, data):
        self.data, self.pos = data, 0

    def read_exact(self, n):
        if self.pos + n > len(self.data):
            raise UnexpectedEof()
        out = self.data[self.pos:self.pos + n]
        self.pos += n
        return out

    def read_u32(self):
        return int.from_bytes(self.read_exact(4), "big")


def parameters_read(data, checked):
    """lib.rs:289-398 (and :159-215 for the verifying key) -> dict; raises the first error in stream order."""
    rd = _Reader(data)

    def point(group, chk, allow_identity):
        p = from_uncompressed(group, rd.read_exact(96 if group == 1 else 192), chk)
        if p is None and not allow_identity:
            raise PointAtInfinity()
        return p

    vk = {}
    for name, group in (("alpha_g1", 1), ("beta_g1", 1), ("beta_g2", 2), ("gamma_g2", 2), ("delta_g1", 1), ("delta_g2", 2)):
        vk[name] = point(group, True, True)
    vk["ic"] = [point(1, True, False) for _ in range(rd.read_u32())]
    out = {"vk": vk}
    for name, group in (("h", 1), ("l", 1), ("a", 1), ("b_g1", 1), ("b_g2", 2)):
        out[name] = [point(group, checked, False) for _ in range(rd.read_u32())]
    return out


def proof_write(a, b, c):
    """lib.rs:38-46"""
    return bls.g1_compress(a) + bls.g2_compress(b) + bls.g1_compress(c)
