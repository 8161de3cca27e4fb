"""Restatement of /root/reference/groth16/src/generator.rs:159-507
(`generate_parameters`) - a FIXTURE PRODUCER for the parity tests (SURVEY §2
row 7: out of scope for kernels).  wNAF fixed-base multiplication and
batch_normalize are replaced by plain scalar multiplication: they only change
how [k]G is computed, not its value.
"""

from .core import AUX, INPUT, ConstraintSystem, LinearCombination, Variable
from .domain import EvaluationDomain
from .errors import UnconstrainedVariable, UnexpectedIdentity
from .multicore import Worker


class KeypairAssembly(ConstraintSystem):
    """generator.rs:43-156"""

    def __init__(self, r):
        self.r = r
        self.num_inputs = 0
        self.num_aux = 0
        self.num_constraints = 0
        self.at_inputs, self.bt_inputs, self.ct_inputs = [], [], []
        self.at_aux, self.bt_aux, self.ct_aux = [], [], []

    def alloc(self, f):
        idx = self.num_aux
        self.num_aux += 1
        self.at_aux.append([])
        self.bt_aux.append([])
        self.ct_aux.append([])
        return Variable(AUX, idx)

    def alloc_input(self, f):
        idx = self.num_inputs
        self.num_inputs += 1
        self.at_inputs.append([])
        self.bt_inputs.append([])
        self.ct_inputs.append([])
        return Variable(INPUT, idx)

    def enforce(self, a, b, c):
        zero = LinearCombination(self.r)

        def ev(lc, inputs, aux):
            for var, coeff in lc.terms:
                (inputs if var.kind == INPUT else aux)[var.idx].append((coeff, self.num_constraints))

        ev(a(zero), self.at_inputs, self.at_aux)
        ev(b(zero), self.bt_inputs, self.bt_aux)
        ev(c(zero), self.ct_inputs, self.ct_aux)
        self.num_constraints += 1


class VerifyingKey:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class Parameters:
    """groth16/src/lib.rs:219-245"""

    def __init__(self, vk, h, l, a, b_g1, b_g2):
        self.vk, self.h, self.l, self.a, self.b_g1, self.b_g2 = vk, h, l, a, b_g1, b_g2


def generate_parameters(engine, circuit, g1, g2, alpha, beta, gamma, delta, tau, worker=None):
    F = engine.Fr
    q = F.r
    G1, G2 = engine.G1, engine.G2
    worker = worker or Worker()
    asm = KeypairAssembly(q)
    asm.alloc_input(lambda: 1)
    circuit(asm)
    for i in range(asm.num_inputs):
        asm.enforce(lambda lc, i=i: lc + Variable(INPUT, i), lambda lc: lc, lambda lc: lc)

    dom = EvaluationDomain.from_coeffs(F, [0] * asm.num_constraints)
    m = len(dom.coeffs)
    if gamma % q == 0 or delta % q == 0:
        raise UnexpectedIdentity()
    gamma_inverse = F.inv(gamma)
    delta_inverse = F.inv(delta)

    # powers of tau, generator.rs:252-264
    dom.coeffs = [pow(tau, i, q) for i in range(m)]
    coeff = (dom.z(tau) * delta_inverse) % q  # :267-268
    h = [G1.mul(g1, (dom.coeffs[i] * coeff) % q) for i in range(m - 1)]  # :271-296

    dom.ifft(worker)  # :300
    lag = dom.into_coeffs()

    def eval_at_tau(p):
        acc = 0
        for c_, index in p:
            acc = (acc + lag[index] * c_) % q
        return acc

    def evaluate(at, bt, ct, inv):
        a, b1, b2, ext = [], [], [], []
        for at_i, bt_i, ct_i in zip(at, bt, ct):
            at_v = eval_at_tau(at_i)
            bt_v = eval_at_tau(bt_i)
            ct_v = eval_at_tau(ct_i)
            a.append(G1.mul(g1, at_v) if at_v != 0 else G1.identity())
            b1.append(G1.mul(g1, bt_v) if bt_v != 0 else G1.identity())
            b2.append(G2.mul(g2, bt_v) if bt_v != 0 else G2.identity())
            e = ((at_v * beta + bt_v * alpha + ct_v) * inv) % q
            ext.append(G1.mul(g1, e))
        return a, b1, b2, ext

    a_in, b1_in, b2_in, ic = evaluate(asm.at_inputs, asm.bt_inputs, asm.ct_inputs, gamma_inverse)
    a_aux, b1_aux, b2_aux, l = evaluate(asm.at_aux, asm.bt_aux, asm.ct_aux, delta_inverse)

    for e in l:  # :466-470
        if G1.is_identity(e):
            raise UnconstrainedVariable()

    vk = VerifyingKey(
        alpha_g1=G1.mul(g1, alpha),
        beta_g1=G1.mul(g1, beta),
        beta_g2=G2.mul(g2, beta),
        gamma_g2=G2.mul(g2, gamma),
        delta_g1=G1.mul(g1, delta),
        delta_g2=G2.mul(g2, delta),
        ic=ic,
    )
    # :491-505 identities filtered out of the A/B queries
    return Parameters(
        vk,
        h,
        l,
        [e for e in a_in + a_aux if not G1.is_identity(e)],
        [e for e in b1_in + b1_aux if not G1.is_identity(e)],
        [e for e in b2_in + b2_aux if not G2.is_identity(e)],
    )
