"""bellman_amd - MI355X (gfx950) implementation of bellman's Groth16 proving hot path.

Python host mirror of the reference's interface for the path (names and argument meaning follow
/root/reference/src/{multicore,multiexp,domain}.rs), implemented over the C ABI in
include/bellman_hip.h.  All arithmetic runs in hand-written HIP kernels (bellman_amd/csrc);
there is no CPU fallback - importing works without a GPU, creating a `Worker` does not.
"""

from .errors import (  # noqa: F401
    BellmanHipError,
    InvalidData,
    InvalidPoint,
    PointAtInfinity,
    PolynomialDegreeTooLarge,
    SynthesisError,
    UnconstrainedVariable,
    UnexpectedEof,
    UnexpectedIdentity,
)
from .multicore import Waiter, Worker  # noqa: F401
from .multiexp import Bases, DensityTracker, FullDensity, Scalars, multiexp, multiexp_scalars, multiexp_sharded, point_add  # noqa: F401
from .domain import EvaluationDomain  # noqa: F401
