"""Minimal restatement of the circuit-facing traits in /root/reference/src/lib.rs
(Variable/Index :163-185, LinearCombination :190-300, ConstraintSystem :374-437).
Host-side plumbing only; kept to what the prover/generator restatements need.
"""

INPUT, AUX = 0, 1


class Variable:
    __slots__ = ("kind", "idx")

    def __init__(self, kind, idx):
        self.kind = kind
        self.idx = idx


class LinearCombination:
    """lib.rs:190-300: an ordered list of (variable, coeff); NO merging of
    duplicate variables (e.g. `a + a` keeps two terms)."""

    def __init__(self, r, terms=None):
        self.r = r
        self.terms = list(terms or [])

    def add(self, var, coeff=1):
        return LinearCombination(self.r, self.terms + [(var, coeff % self.r)])

    def sub(self, var, coeff=1):
        return LinearCombination(self.r, self.terms + [(var, (-coeff) % self.r)])

    def __add__(self, other):
        if isinstance(other, Variable):
            return self.add(other)
        coeff, var = other
        return self.add(var, coeff)

    def __sub__(self, other):
        if isinstance(other, Variable):
            return self.sub(other)
        coeff, var = other
        return self.sub(var, coeff)


class ConstraintSystem:
    """lib.rs:374-437 subset: one(), alloc, alloc_input, enforce."""

    @staticmethod
    def one():
        return Variable(INPUT, 0)
