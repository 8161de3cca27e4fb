"""SynthesisError mirror (reference: src/lib.rs:303-319) + C-ABI return-code mapping."""


class SynthesisError(Exception):
    pass


class UnexpectedIdentity(SynthesisError):
    """src/multiexp.rs:63-65"""


class UnexpectedEof(SynthesisError):
    """io::ErrorKind::UnexpectedEof, "expected more bases from source" (src/multiexp.rs:55-61)"""


class PolynomialDegreeTooLarge(SynthesisError):
    """src/domain.rs:57-59"""


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
    if rc == -2:
        # the reference panics here (assert!), e.g. src/multiexp.rs:324-329, src/domain.rs:155,174
        raise AssertionError("%s: invalid argument (the reference panics)" % what)
    if rc == -3:
        raise BellmanHipError("%s: no gfx950 device available (no CPU fallback)" % what)
    raise BellmanHipError("%s failed with code %d" % (what, rc))
