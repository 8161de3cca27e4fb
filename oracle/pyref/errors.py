"""SynthesisError restated from /root/reference/src/lib.rs:303-319.

Only the variants reachable from the proving hot path are listed.  The integer
codes are the ones the C ABI (include/bellman_hip.h) returns.
"""


class SynthesisError(Exception):
    code = -1


class AssignmentMissing(SynthesisError):
    code = 4


class PolynomialDegreeTooLarge(SynthesisError):
    """src/domain.rs:57-59"""

    code = 3


class UnexpectedIdentity(SynthesisError):
    """src/multiexp.rs:63-65"""

    code = 1


class IoErrorUnexpectedEof(SynthesisError):
    """src/multiexp.rs:55-61,74-80: io::ErrorKind::UnexpectedEof, "expected more bases from source" """

    code = 2


class UnconstrainedVariable(SynthesisError):
    code = 5
