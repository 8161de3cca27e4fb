"""SynthesisError mirror (reference: src/lib.rs:303-319) + C-ABI return-code mapping."""


class SynthesisError(Exception):
    pass


class UnexpectedIdentity(SynthesisError):
    """src/multiexp.rs:63-65"""


class UnexpectedEof(SynthesisError, IOError):
    """io::ErrorKind::UnexpectedEof: `SynthesisError::IoError` "expected more bases from source" from multiexp
    (src/multiexp.rs:55-61) and the plain `io::Error` of a truncated `Parameters::read` / `VerifyingKey::read`
    (groth16/src/lib.rs:159-215,289-398) - so it is both a SynthesisError and an IOError."""


class PolynomialDegreeTooLarge(SynthesisError):
    """src/domain.rs:57-59"""


class UnconstrainedVariable(SynthesisError):
    """groth16/src/generator.rs:464-470"""


class InvalidData(IOError):
    """io::ErrorKind::InvalidData from Parameters::read / VerifyingKey::read (groth16/src/lib.rs:159-215,289-398)"""


class InvalidPoint(InvalidData):
    """"invalid G1" / "invalid G2": bad encoding, or (checked) not on the curve / not in the subgroup"""


class PointAtInfinity(InvalidData):
    """"point at infinity": a query or ic point is the identity"""


class BellmanHipError(RuntimeError):
    """HIP runtime failure / missing device: never silently replaced by a CPU path."""


def check(rc, what="bellman_hip call"):
    if rc == 0:
        return
    if rc == 1:
        raise UnexpectedIdentity()
    if rc == 2:
        raise UnexpectedEof("expected more bases from source")
    if rc == 3:
        raise PolynomialDegreeTooLarge()
    if rc == 5:
        raise UnconstrainedVariable()
    if rc == 6:
        raise InvalidPoint("invalid G1/G2")
    if rc == 7:
        raise PointAtInfinity("point at infinity")
    if rc == -2:
        # the reference panics here (assert!), e.g. src/multiexp.rs:324-329, src/domain.rs:155,174
        raise AssertionError("%s: invalid argument (the reference panics)" % what)
    if rc == -3:
        raise BellmanHipError("%s: no gfx950 device available (no CPU fallback)" % what)
    raise BellmanHipError("%s failed with code %d" % (what, rc))
