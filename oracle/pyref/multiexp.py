"""Line-by-line restatement of /root/reference/src/multiexp.rs.

Generic over a group object `G` (identity/add/double) and a scalar field
(NUM_BITS).  Scalars are Python ints in [0, r); `Exponent` classification and
digit extraction follow multiexp.rs:159-208.
"""

import math

from .errors import IoErrorUnexpectedEof, SynthesisError, UnexpectedIdentity
from .multicore import Waiter


class VecSource:
    """`(Arc<Vec<G>>, usize)` as Source - multiexp.rs:45-86."""

    def __init__(self, group, bases, offset):
        self.G = group
        self.bases = bases
        self.pos = offset

    def next(self):
        # multiexp.rs:54-71
        if len(self.bases) <= self.pos:
            raise IoErrorUnexpectedEof("expected more bases from source")
        if self.G.is_identity(self.bases[self.pos]):
            raise UnexpectedIdentity()
        ret = self.bases[self.pos]
        self.pos += 1
        return ret

    def skip(self, amt):
        # multiexp.rs:73-85 : EOF check BEFORE advancing, no identity check
        if len(self.bases) <= self.pos:
            raise IoErrorUnexpectedEof("expected more bases from source")
        self.pos += amt


class FullDensity:
    """multiexp.rs:95-115"""

    def iter(self):
        while True:
            yield True

    def get_query_size(self):
        return None


class DensityTracker:
    """multiexp.rs:117-157"""

    def __init__(self):
        self.bv = []

    def iter(self):
        return iter(self.bv)

    def get_query_size(self):
        return len(self.bv)

    def add_element(self):
        self.bv.append(False)

    def inc(self, idx):
        if not self.bv[idx]:
            self.bv[idx] = True

    def get_total_density(self):
        return sum(1 for b in self.bv if b)


ZERO, ONE, BITS = 0, 1, 2


def exponent_from(s):
    """`impl From<&F> for Exponent<F>` - multiexp.rs:172-182.
    Returns (kind, canonical little-endian integer)."""
    if s == 0:
        return (ZERO, 0)
    if s == 1:
        return (ONE, 1)
    return (BITS, s)


def chunks(exp, c, repr_bits):
    """`Exponent::chunks` - multiexp.rs:190-208.  `repr_bits` is the width of
    FieldBits<ReprBits> (256 for BLS12-381 Fr = [u64;4]; 64 for the toy field),
    split into ceil(repr_bits/c) LSB-first c-bit digits."""
    kind, v = exp
    if kind != BITS:
        return (kind, None)
    n = -(-repr_bits // c)
    mask = (1 << c) - 1
    return (BITS, [(v >> (c * i)) & mask for i in range(n)])


def window_size(n):
    """multiexp.rs:318-322"""
    if n < 32:
        return 3
    return int(math.ceil(math.log(float(n & 0xFFFFFFFF))))


def multiexp_inner(G, num_bits, repr_bits, bases, offset, density_map, exponents, c):
    """multiexp.rs:210-301.  Returns the group element or raises."""

    def this(chunk):
        # multiexp.rs:224-278
        acc = G.identity()
        src = VecSource(G, bases, offset)
        buckets = [G.identity() for _ in range((1 << c) - 1)]
        handle_trivial = chunk == 0
        for (kind, digits), density in zip(chunked, density_map.iter()):
            if density:
                if kind == ZERO:
                    src.skip(1)
                elif kind == ONE:
                    if handle_trivial:
                        acc = G.add(acc, src.next())
                    else:
                        src.skip(1)
                else:
                    e = digits[chunk]
                    if e != 0:
                        buckets[e - 1] = G.add(buckets[e - 1], src.next())
                    else:
                        src.skip(1)
        # summation by parts, multiexp.rs:271-275
        running_sum = G.identity()
        for b in reversed(buckets):
            running_sum = G.add(running_sum, b)
            acc = G.add(acc, running_sum)
        return acc

    chunked = [chunks(e, c, repr_bits) for e in exponents]  # multiexp.rs:281-286

    parts = []
    for chunk, _ in enumerate(range(0, num_bits, c)):  # multiexp.rs:288-293
        try:
            parts.append(this(chunk))
        except SynthesisError as e:  # Result<_, SynthesisError> per window
            parts.append(e)

    # multiexp.rs:295-300: fold high -> low; first Err met (highest window) wins
    acc = G.identity()
    for part in reversed(parts):
        if isinstance(part, Exception):
            raise part
        for _ in range(c):
            acc = G.double(acc)
        acc = G.add(acc, part)
    return acc


def multiexp(pool, G, field, bases, offset, density_map, exponents, repr_bits=None):
    """multiexp.rs:305-332.  `exponents` is a list of Exponent tuples
    (see exponent_from).  Returns a Waiter whose wait() yields the element or
    raises the SynthesisError."""
    if repr_bits is None:
        repr_bits = 256 if field.NUM_BITS > 64 else 64
    c = window_size(len(exponents))
    qs = density_map.get_query_size()
    if qs is not None:
        assert qs == len(exponents)  # multiexp.rs:324-329 (panic)

    def run():
        try:
            return multiexp_inner(
                G, field.NUM_BITS, repr_bits, bases, offset, density_map, exponents, c
            )
        except SynthesisError as e:
            return e

    w = pool.compute(run)

    class _ResultWaiter(Waiter):
        def wait(self_inner):
            v = w.wait()
            if isinstance(v, Exception):
                raise v
            return v

    return _ResultWaiter(None)


def naive_multiexp(G, bases, scalars):
    """multiexp.rs:337-350 (the reference test's own checker)."""
    assert len(bases) == len(scalars)
    acc = G.identity()
    for b, s in zip(bases, scalars):
        acc = G.add(acc, G.mul(b, s))
    return acc
