"""Circuits used by the prover tests, written once against the ConstraintSystem interface shared by
the oracle restatement (oracle/pyref/core.py) and the product's Python mirror (bellman_amd/groth16.py).
The C++ versions of the same circuits live in bellman_amd/csrc/groth16_capi.cpp."""

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
MASK64 = (1 << 64) - 1
MIMC_ROUNDS = 322


def mimc_hash(xl, xr, constants):
    """groth16/tests/common/mod.rs:20-35"""
    for c in constants:
        t = (xl + c) % Q
        xl, xr = (t * t % Q * t + xr) % Q, xl
    return xl


def mimc_circuit(xl, xr, constants):
    """MiMCDemo::synthesize, groth16/tests/common/mod.rs:48-129"""

    def synth(cs):
        xl_v, xr_v = xl, xr
        xlv = cs.alloc(lambda: xl_v)
        xrv = cs.alloc(lambda: xr_v)
        n = len(constants)
        for i, ci in enumerate(constants):
            t0 = (xl_v + ci) % Q
            tmp_v = t0 * t0 % Q
            tmp = cs.alloc(lambda: tmp_v)
            cs.enforce(lambda lc: lc + xlv + (ci, cs.one()), lambda lc: lc + xlv + (ci, cs.one()), lambda lc: lc + tmp)
            new_v = (t0 * tmp_v + xr_v) % Q
            new_xl = cs.alloc_input(lambda: new_v) if i == n - 1 else cs.alloc(lambda: new_v)
            cs.enforce(lambda lc: lc + tmp, lambda lc: lc + xlv + (ci, cs.one()), lambda lc: lc + new_xl - xrv)
            xrv, xr_v = xlv, xl_v
            xlv, xl_v = new_xl, new_v

    return synth


def _splitmix(state):
    state = (state + 0x9E3779B97F4A7C15) & MASK64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return state, z ^ (z >> 31)


def chain_constants(rounds, seed):
    st = seed & MASK64
    ks = []
    for _ in range(rounds):
        st, k = _splitmix(st)
        st, k2 = _splitmix(st)
        ks.append((k % Q, (k2 | 1) % Q))
    return ks


def chain_circuit(rounds, seed, x0):
    """ChainCircuit::synthesize of groth16_capi.cpp (synthetic R1CS of SURVEY.md 8d)."""

    def synth(cs):
        x_v = x0 % Q
        x = cs.alloc(lambda: x_v)
        first = x
        for i, (k, k2) in enumerate(chain_constants(rounds, seed)):
            lhs = (x_v + k) % Q
            rhs = k2 if (i & 1) else (x_v + k2) % Q
            nxt_v = lhs * rhs % Q
            nxt = cs.alloc(lambda: nxt_v)
            if i & 1:
                cs.enforce(lambda lc: lc + x + (k, cs.one()) + (0, first), lambda lc: lc + (k2, cs.one()), lambda lc: lc + nxt)
            else:
                cs.enforce(lambda lc: lc + x + (k, cs.one()), lambda lc: lc + x + (k2, cs.one()), lambda lc: lc + nxt)
            x, x_v = nxt, nxt_v
        out = cs.alloc_input(lambda: x_v)
        cs.enforce(lambda lc: lc + x, lambda lc: lc + cs.one(), lambda lc: lc + out)

    return synth


def chain_assignment_fast(rounds, seed, x0):
    """The ProvingAssignment the chain circuit produces (after create_proof's input constraints),
    computed in closed form - for sizes where generic Python synthesis is too slow."""
    xs = [x0 % Q]
    a_ev, b_ev, c_ev = [], [], []
    for i, (k, k2) in enumerate(chain_constants(rounds, seed)):
        x = xs[-1]
        lhs = (x + k) % Q
        rhs = k2 if (i & 1) else (x + k2) % Q
        nxt = lhs * rhs % Q
        a_ev.append(lhs)
        b_ev.append(rhs)
        c_ev.append(nxt)
        xs.append(nxt)
    out = xs[-1]
    a_ev.append(out); b_ev.append(1); c_ev.append(out)  # x_M * 1 = out
    a_ev += [1, out]; b_ev += [0, 0]; c_ev += [0, 0]    # input_i * 0 = 0 (prover.rs:208-215)
    n_aux = rounds + 1
    a_aux_density = [True] * n_aux
    b_aux_density = [(i % 2 == 0) and i < rounds for i in range(n_aux)]
    b_input_density = [True, False]
    return dict(a=a_ev, b=b_ev, c=c_ev, input_assignment=[1, out], aux_assignment=xs,
                a_aux_density=a_aux_density, b_input_density=b_input_density, b_aux_density=b_aux_density)


def forms_circuit(rounds, seed, x0):
    """FormsCircuit::synthesize of csrc/demo_circuits.cpp: every way a linear combination can reach `enforce` through the
    C++ mirror - evaluating terms that need a product, subtraction, zero coefficients, repeated variables, stored
    combinations of five and six terms that ignore the closure's argument, the empty combination.  Written against the
    ConstraintSystem interface of oracle/pyref/core.py (the terms of `LinearCombination(Q)` are what the C++ stored
    combination holds)."""
    from oracle.pyref.core import LinearCombination

    def synth(cs):
        st = seed & MASK64
        v_v = x0 % Q
        w_v = (v_v * v_v + 3) % Q
        v = cs.alloc(lambda: v_v)
        w = cs.alloc(lambda: w_v)
        for i in range(rounds):
            st, k = _splitmix(st)
            st, k2 = _splitmix(st)
            k, k2 = k % Q, (k2 | 1) % Q
            out_v = (v_v * k - w_v) * k2 % Q
            out = cs.alloc_input(lambda: out_v) if i % 3 == 2 else cs.alloc(lambda: out_v)
            one = cs.one()

            def b_side(_lc):
                s = LinearCombination(Q) + v + (k2, w) - (k, one) + one - out
                if i & 1:
                    s = s + (k, out)
                return s

            cs.enforce(lambda lc: lc + (k, v) - w - (k2, one) + (0, out) + v, b_side, (lambda lc: lc) if i % 3 == 0 else (lambda lc: lc + out))
            v, v_v = w, w_v
            w, w_v = out, out_v

    return synth


def random_circuit(rounds, seed, x0):
    """RandomCircuit::synthesize of csrc/demo_circuits.cpp: the structure of every constraint is drawn from
    SplitMix64(seed) - the same draws in the same order as the C++ fixture."""
    from oracle.pyref.core import LinearCombination

    class Draw:
        def __init__(self, st):
            self.st = st & MASK64

        def next(self):
            self.st, z = _splitmix(self.st)
            return z

        def below(self, n):
            return self.next() % n

    def synth(cs):
        d = Draw(seed)
        vs = [cs.one()]
        value = [x0 % Q]

        def fresh_value():
            value[0] = (value[0] * value[0] + d.next()) % Q
            return value[0]

        def coefficient():
            c = d.below(5)
            if c == 0:
                return 0
            if c == 1:
                return 1
            if c == 2:
                return Q - 1
            if c == 3:
                return d.below(16)
            a = d.next()
            return a * d.next() % Q

        def apply(lc, terms):
            for op, k, v in terms:
                if op == 0:
                    lc = lc + v
                elif op == 1:
                    lc = lc - v
                elif op == 2:
                    lc = lc + (k, v)
                else:
                    lc = lc - (k, v)
            return lc

        for _ in range(rounds):
            for _j in range(d.below(3)):
                is_input = d.below(4) == 0
                v = fresh_value()
                vs.append(cs.alloc_input(lambda: v) if is_input else cs.alloc(lambda: v))
            sides = []
            for _s in range(3):
                form = d.below(3)
                n_terms = 0 if form == 0 else (1 + d.below(9) if form == 1 else d.below(10))
                terms = []
                for _t in range(n_terms):
                    op = d.below(4)
                    k = coefficient() if op >= 2 else 1
                    terms.append((op, k, vs[d.below(len(vs))]))
                sides.append((form, terms))

            def closure(side):
                form, terms = side
                if form == 2:
                    return lambda _lc: apply(LinearCombination(Q), terms)
                return lambda lc: apply(lc, terms)

            cs.enforce(closure(sides[0]), closure(sides[1]), closure(sides[2]))

    return synth


def _boolmix_indices(i):
    a, c = (7 * i + 1) % 64, (29 * i + 11) % 64
    b = (13 * i + 5) % 64
    if a == b:
        b = (b + 1) % 64
    return a, b, c


def _boolmix_init(seed, x0):
    _, z = _splitmix(seed & MASK64)
    return ((x0 % Q) & MASK64) ^ z


def boolmix_rounds(log_m):
    """steps of the boolean circuit whose constraint count (64 bit checks + steps + one pack per 64 steps + the final pack +
    create_proof's 2 input constraints) fills a domain of 2^log_m without exceeding it"""
    return ((1 << log_m) - 67) * 64 // 65


def boolmix_circuit(rounds, seed, x0):
    """BoolMixCircuit::synthesize of csrc/demo_circuits.cpp: a boolean-heavy circuit in the shape of the reference's
    bit-level gadgets (src/gadgets/boolean.rs: AllocatedBit::alloc, and, xor), the state packed into a field element every
    64 steps.  Written against the ConstraintSystem interface of oracle/pyref/core.py."""

    def synth(cs):
        init = _boolmix_init(seed, x0)
        one = cs.one()
        s, v = [], []
        for j in range(64):
            bit = (init >> j) & 1
            var = cs.alloc(lambda: bit)
            cs.enforce(lambda lc: lc + one - var, lambda lc: lc + var, lambda lc: lc)
            s.append(var)
            v.append(bit)

        def pack_value():
            return sum(b << j for j, b in enumerate(v))

        def pack_lc(lc):
            for j in range(64):
                lc = lc + (1 << j, s[j])
            return lc

        def xor_into(dst, other, other_v):
            tv = v[dst] ^ other_v
            x = s[dst]
            t = cs.alloc(lambda: tv)
            cs.enforce(lambda lc: lc + x + x, lambda lc: lc + other, lambda lc: lc + x + other - t)
            s[dst], v[dst] = t, tv

        pending, pending_v = s[0], v[0]
        for i in range(rounds):
            a, b, c = _boolmix_indices(i)
            if i % 3 == 0:
                uv = v[a] & v[b]
                xa, xb = s[a], s[b]
                u = cs.alloc(lambda: uv)
                cs.enforce(lambda lc: lc + xa, lambda lc: lc + xb, lambda lc: lc + u)
                pending, pending_v = u, uv
            elif i % 3 == 1:
                xor_into(c, pending, pending_v)
            else:
                xor_into(a, s[b], v[b])
            if i % 64 == 63:
                nv = pack_value()
                num = cs.alloc(lambda: nv)
                cs.enforce(pack_lc, lambda lc: lc + one, lambda lc: lc + num)
        outv = pack_value()
        out = cs.alloc_input(lambda: outv)
        cs.enforce(pack_lc, lambda lc: lc + one, lambda lc: lc + out)

    return synth


def boolmix_assignment_fast(rounds, seed, x0):
    """The ProvingAssignment of boolmix_circuit (after create_proof's input constraints) computed directly - evaluations,
    assignments and density maps as prover.rs:19-55 would leave them - for sizes where the generic synthesis is too slow.
    Checked against the generic synthesis in tests/test_boolean_circuit_cpu.py."""
    init = _boolmix_init(seed, x0)
    aux, a_ev, b_ev, c_ev = [], [], [], []
    a_den, b_den = [], []          # density of the aux variables in the A and B queries
    s, v = [], []

    def alloc(val):
        aux.append(val)
        a_den.append(False)
        b_den.append(False)
        return len(aux) - 1

    for j in range(64):
        bit = (init >> j) & 1
        var = alloc(bit)
        a_ev.append((1 - bit) % Q); b_ev.append(bit); c_ev.append(0)
        a_den[var] = True; b_den[var] = True
        s.append(var); v.append(bit)

    def pack():
        val = 0
        for j in range(64):
            val |= v[j] << j
            a_den[s[j]] = True
        return val

    pending, pending_v = s[0], v[0]
    for i in range(rounds):
        a, b, c = _boolmix_indices(i)
        k = i % 3
        if k == 0:
            uv = v[a] & v[b]
            a_den[s[a]] = True; b_den[s[b]] = True
            u = alloc(uv)
            a_ev.append(v[a]); b_ev.append(v[b]); c_ev.append(uv)
            pending, pending_v = u, uv
        else:
            dst, other, other_v = (c, pending, pending_v) if k == 1 else (a, s[b], v[b])
            xv = v[dst]
            tv = xv ^ other_v
            a_den[s[dst]] = True; b_den[other] = True
            t = alloc(tv)
            a_ev.append(2 * xv); b_ev.append(other_v); c_ev.append((xv + other_v - tv) % Q)
            s[dst], v[dst] = t, tv
        if i % 64 == 63:
            nv = pack()
            alloc(nv)
            a_ev.append(nv); b_ev.append(1); c_ev.append(nv)
    outv = pack()
    a_ev.append(outv); b_ev.append(1); c_ev.append(outv)
    a_ev += [1, outv]; b_ev += [0, 0]; c_ev += [0, 0]    # input_i * 0 = 0 (prover.rs:208-215)
    return dict(a=a_ev, b=b_ev, c=c_ev, input_assignment=[1, outv], aux_assignment=aux,
                a_aux_density=a_den, b_input_density=[True, False], b_aux_density=b_den)
