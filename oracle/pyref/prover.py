"""Line-by-line restatement of /root/reference/groth16/src/prover.rs."""

from .core import AUX, INPUT, ConstraintSystem, LinearCombination, Variable
from .domain import EvaluationDomain
from .errors import UnexpectedIdentity
from .multicore import Worker
from .multiexp import DensityTracker, FullDensity, exponent_from, multiexp


def eval_lc(r, lc, input_density, aux_density, input_assignment, aux_assignment):
    """prover.rs:19-55 (zero coefficients contribute neither value nor density, :31)."""
    acc = 0
    for var, coeff in lc.terms:
        if coeff != 0:
            if var.kind == INPUT:
                tmp = input_assignment[var.idx]
                if input_density is not None:
                    input_density.inc(var.idx)
            else:
                tmp = aux_assignment[var.idx]
                if aux_density is not None:
                    aux_density.inc(var.idx)
            if coeff != 1:
                tmp = (tmp * coeff) % r
            acc = (acc + tmp) % r
    return acc


class ProvingAssignment(ConstraintSystem):
    """prover.rs:57-162"""

    def __init__(self, r):
        self.r = r
        self.a_aux_density = DensityTracker()
        self.b_input_density = DensityTracker()
        self.b_aux_density = DensityTracker()
        self.a = []
        self.b = []
        self.c = []
        self.input_assignment = []
        self.aux_assignment = []

    def alloc(self, f):
        self.aux_assignment.append(f() % self.r)
        self.a_aux_density.add_element()
        self.b_aux_density.add_element()
        return Variable(AUX, len(self.aux_assignment) - 1)

    def alloc_input(self, f):
        self.input_assignment.append(f() % self.r)
        self.b_input_density.add_element()
        return Variable(INPUT, len(self.input_assignment) - 1)

    def enforce(self, a, b, c):
        zero = LinearCombination(self.r)
        a = a(zero)
        b = b(zero)
        c = c(zero)
        ia, aa = self.input_assignment, self.aux_assignment
        self.a.append(eval_lc(self.r, a, None, self.a_aux_density, ia, aa))
        self.b.append(eval_lc(self.r, b, self.b_input_density, self.b_aux_density, ia, aa))
        self.c.append(eval_lc(self.r, c, None, None, ia, aa))


class Proof:
    def __init__(self, a, b, c):
        self.a, self.b, self.c = a, b, c


def compute_h_coeffs(field, a_ev, b_ev, c_ev, worker):
    """prover.rs:221-240: the quotient polynomial, truncated to m-1 coefficients."""
    a = EvaluationDomain.from_coeffs(field, a_ev)
    b = EvaluationDomain.from_coeffs(field, b_ev)
    c = EvaluationDomain.from_coeffs(field, c_ev)
    a.ifft(worker)
    a.coset_fft(worker)
    b.ifft(worker)
    b.coset_fft(worker)
    c.ifft(worker)
    c.coset_fft(worker)
    a.mul_assign(worker, b)
    a.sub_assign(worker, c)
    a.divide_by_z_on_coset(worker)
    a.icoset_fft(worker)
    co = a.into_coeffs()
    return co[: len(co) - 1]


def create_proof(engine, circuit, params, r, s, worker=None, trace=None):
    """prover.rs:182-361.  `circuit(cs)` synthesises into `cs`.
    `params` is a generator.Parameters (the `&Parameters` ParameterSource,
    groth16/src/lib.rs:435-473).  `trace` (dict) receives intermediates."""
    F = engine.Fr
    G1, G2 = engine.G1, engine.G2
    q = F.r
    prover = ProvingAssignment(q)
    prover.alloc_input(lambda: 1)  # :204
    circuit(prover)  # :206
    for i in range(len(prover.input_assignment)):  # :208-215
        prover.enforce(lambda lc, i=i: lc + Variable(INPUT, i), lambda lc: lc, lambda lc: lc)

    worker = worker or Worker()
    vk = params.vk

    h_coeffs = compute_h_coeffs(F, prover.a, prover.b, prover.c, worker)
    if trace is not None:
        trace["h_coeffs"] = list(h_coeffs)
        trace["a_evals"] = list(prover.a)
        trace["b_evals"] = list(prover.b)
        trace["c_evals"] = list(prover.c)
    h_exp = [exponent_from(x) for x in h_coeffs]
    h = multiexp(worker, G1, F, params.h, 0, FullDensity(), h_exp)

    input_assignment = [exponent_from(x) for x in prover.input_assignment]
    aux_assignment = [exponent_from(x) for x in prover.aux_assignment]
    if trace is not None:
        trace["input_assignment"] = list(prover.input_assignment)
        trace["aux_assignment"] = list(prover.aux_assignment)
        trace["a_aux_density"] = list(prover.a_aux_density.bv)
        trace["b_input_density"] = list(prover.b_input_density.bv)
        trace["b_aux_density"] = list(prover.b_aux_density.bv)

    l = multiexp(worker, G1, F, params.l, 0, FullDensity(), aux_assignment)

    n_in = len(input_assignment)
    # get_a(num_inputs, _): ((a,0),(a,num_inputs))   groth16/src/lib.rs:451-457
    a_inputs = multiexp(worker, G1, F, params.a, 0, FullDensity(), input_assignment)
    a_aux = multiexp(worker, G1, F, params.a, n_in, prover.a_aux_density, aux_assignment)

    b_input_density_total = prover.b_input_density.get_total_density()
    # get_b_g1(b_input_density_total, _): ((b,0),(b,b_input_density_total))   lib.rs:459-473
    b_g1_inputs = multiexp(worker, G1, F, params.b_g1, 0, prover.b_input_density, input_assignment)
    b_g1_aux = multiexp(
        worker, G1, F, params.b_g1, b_input_density_total, prover.b_aux_density, aux_assignment
    )
    b_g2_inputs = multiexp(worker, G2, F, params.b_g2, 0, prover.b_input_density, input_assignment)
    b_g2_aux = multiexp(
        worker, G2, F, params.b_g2, b_input_density_total, prover.b_aux_density, aux_assignment
    )

    if G1.is_identity(vk.delta_g1) or G2.is_identity(vk.delta_g2):  # :320-324
        raise UnexpectedIdentity()

    g_a = G1.add(G1.mul(vk.delta_g1, r), vk.alpha_g1)  # :326-327
    g_b = G2.add(G2.mul(vk.delta_g2, s), vk.beta_g2)  # :328-329
    rs = (r * s) % q
    g_c = G1.mul(vk.delta_g1, rs)  # :335
    g_c = G1.add(g_c, G1.mul(vk.alpha_g1, s))
    g_c = G1.add(g_c, G1.mul(vk.beta_g1, r))

    a_answer = G1.add(a_inputs.wait(), a_aux.wait())  # :339-340
    g_a = G1.add(g_a, a_answer)
    a_answer = G1.mul(a_answer, s)
    g_c = G1.add(g_c, a_answer)

    b1_answer = G1.add(b_g1_inputs.wait(), b_g1_aux.wait())  # :345-346
    b2_answer = G2.add(b_g2_inputs.wait(), b_g2_aux.wait())
    g_b = G2.add(g_b, b2_answer)
    b1_answer = G1.mul(b1_answer, r)
    g_c = G1.add(g_c, b1_answer)
    g_c = G1.add(g_c, h.wait())
    g_c = G1.add(g_c, l.wait())
    return Proof(g_a, g_b, g_c)
