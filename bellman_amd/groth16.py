"""`groth16::create_proof` mirror (reference: groth16/src/prover.rs:19-361) for Python callers.

Circuits are written against `ConstraintSystem` exactly like bellman user code; synthesis and
linear-combination evaluation run on the host (as in the reference), everything after
`circuit.synthesize` - the h-polynomial FFT pipeline and the eight multiexps - runs on the GPU
through bh_groth16_prove_assignment.  Field elements are Python ints in [0, q).
"""

import ctypes

import numpy as np

from . import _lib
from .errors import check

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
_R = (1 << 256) % Q
_MASK = (1 << 64) - 1
INPUT, AUX = 0, 1


def fr_to_mont_array(vals):
    """ints in [0,q) -> [n,4] uint64 Montgomery limbs (the bytes of bls12_381::Scalar).  An [n,4] uint64 array is
    taken as already converted (callers that prove many times convert their constants once)."""
    if isinstance(vals, np.ndarray) and vals.dtype == np.uint64 and vals.ndim == 2 and vals.shape[1] == 4:
        return np.ascontiguousarray(vals)
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        m = (v % Q) * _R % Q
        out[i] = [(m >> (64 * k)) & _MASK for k in range(4)]
    return out


class Variable:
    """src/lib.rs:163-185"""

    __slots__ = ("kind", "idx")

    def __init__(self, kind, idx):
        self.kind, self.idx = kind, idx


class LinearCombination:
    """src/lib.rs:190-300: ordered (variable, coeff) terms, duplicates not merged."""

    def __init__(self, terms=None):
        self.terms = list(terms or [])

    @staticmethod
    def zero():
        return LinearCombination()

    def __add__(self, other):
        coeff, var = (1, other) if isinstance(other, Variable) else other
        return LinearCombination(self.terms + [(var, coeff % Q)])

    def __sub__(self, other):
        coeff, var = (1, other) if isinstance(other, Variable) else other
        return LinearCombination(self.terms + [(var, (-coeff) % Q)])


class ConstraintSystem:
    """src/lib.rs:374-437 (subset used by circuits: one, alloc, alloc_input, enforce)."""

    @staticmethod
    def one():
        return Variable(INPUT, 0)


class _Density:
    """src/multiexp.rs:117-157"""

    def __init__(self):
        self.bv = []

    def add_element(self):
        self.bv.append(False)

    def inc(self, i):
        self.bv[i] = True

    def get_total_density(self):
        return sum(self.bv)

    def words(self):
        n = len(self.bv)
        padded = np.zeros(((n + 63) // 64 + 1) * 64, dtype=np.uint8)
        padded[:n] = np.asarray(self.bv, dtype=np.uint8)
        return np.packbits(padded, bitorder="little").view(np.uint64).copy()


class ProvingAssignment(ConstraintSystem):
    """prover.rs:57-162"""

    def __init__(self):
        self.a_aux_density, self.b_input_density, self.b_aux_density = _Density(), _Density(), _Density()
        self.a, self.b, self.c = [], [], []
        self.input_assignment, self.aux_assignment = [], []

    def alloc(self, f):
        self.aux_assignment.append(f() % Q)
        self.a_aux_density.add_element()
        self.b_aux_density.add_element()
        return Variable(AUX, len(self.aux_assignment) - 1)

    def alloc_input(self, f):
        self.input_assignment.append(f() % Q)
        self.b_input_density.add_element()
        return Variable(INPUT, len(self.input_assignment) - 1)

    def _eval(self, lc, input_density, aux_density):
        """prover.rs:19-55"""
        acc = 0
        for var, coeff in lc.terms:
            if coeff == 0:
                continue
            if var.kind == INPUT:
                tmp = self.input_assignment[var.idx]
                if input_density is not None:
                    input_density.inc(var.idx)
            else:
                tmp = self.aux_assignment[var.idx]
                if aux_density is not None:
                    aux_density.inc(var.idx)
            if coeff != 1:
                tmp = tmp * coeff % Q
            acc = (acc + tmp) % Q
        return acc

    def enforce(self, a, b, c):
        z = LinearCombination.zero()
        self.a.append(self._eval(a(z), None, self.a_aux_density))
        self.b.append(self._eval(b(z), self.b_input_density, self.b_aux_density))
        self.c.append(self._eval(c(z), None, None))


class Proof:
    """groth16/src/lib.rs:25-30: affine records (numpy uint64): a [12], b [24], c [12]."""

    def __init__(self, raw):
        self.a, self.b, self.c = raw[:12].copy(), raw[12:36].copy(), raw[36:48].copy()

    def write(self):
        """Proof::write (groth16/src/lib.rs:38-46): compressed A (48) | B (96) | C (48)"""
        raw = np.concatenate([self.a, self.b, self.c]).astype(np.uint64)
        out = np.zeros(192, dtype=np.uint8)
        _lib.load().bh_proof_write(raw.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
        return out.tobytes()


class Parameters:
    """`&Parameters` as ParameterSource (groth16/src/lib.rs:435-473); query vectors live in HBM."""

    def __init__(self, worker, alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2, h, l, a, b_g1, b_g2):
        lib = _lib.load()
        self.worker = worker
        arrs = [np.ascontiguousarray(x, dtype=np.uint64) for x in (alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2)]
        qs = [np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, w) for x, w in ((h, 12), (l, 12), (a, 12), (b_g1, 12), (b_g2, 24))]
        p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        h_ = ctypes.c_void_p()
        check(lib.bh_groth16_params_create(worker.ctx, *[p(x) for x in arrs], p(qs[0]), qs[0].shape[0], p(qs[1]),
                                           qs[1].shape[0], p(qs[2]), qs[2].shape[0], p(qs[3]), qs[3].shape[0],
                                           p(qs[4]), qs[4].shape[0], ctypes.byref(h_)), "Parameters")
        self._h = h_

    @classmethod
    def read(cls, worker, data, checked):
        """Parameters::read(reader, checked) (groth16/src/lib.rs:289-398): serialized CRS bytes ->
        device-resident parameters; decoding and validation run on the GPU.  Raises UnexpectedEof,
        InvalidPoint or PointAtInfinity - whichever the reference's sequential reader hits first."""
        lib = _lib.load()
        buf = np.frombuffer(data, dtype=np.uint8)   # bytes, bytearray or any buffer: no copy
        self = cls.__new__(cls)
        self.worker = worker
        h_ = ctypes.c_void_p()
        self._h = None
        check(lib.bh_groth16_params_read(worker.ctx, buf.ctypes.data_as(ctypes.c_void_p), buf.size, 1 if checked else 0,
                                         ctypes.byref(h_)), "Parameters.read")
        self._h = h_
        return self

    @classmethod
    def generate(cls, worker, r1cs, g1, g2, alpha, beta, gamma, delta, tau):
        """generate_parameters (groth16/src/generator.rs:163-510) on the device for the circuit whose
        matrices are `r1cs` (R1CS.from_circuit / from_demo).  g1, g2: affine Montgomery records (numpy
        uint64 [12] / [24]); alpha..tau: ints.  Raises UnexpectedIdentity (gamma or delta = 0),
        UnconstrainedVariable, PolynomialDegreeTooLarge."""
        lib = _lib.load()
        g1 = np.ascontiguousarray(g1, dtype=np.uint64)
        g2 = np.ascontiguousarray(g2, dtype=np.uint64)
        sc = fr_to_mont_array([alpha, beta, gamma, delta, tau])
        p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        self = cls.__new__(cls)
        self.worker, self._h = worker, None
        h_ = ctypes.c_void_p()
        check(lib.bh_groth16_generate(worker.ctx, r1cs._h, p(g1), p(g2), *[p(sc[i:i + 1]) for i in range(5)], ctypes.byref(h_)),
              "generate_parameters")
        self._h = h_
        return self

    def write(self):
        """Parameters::write (groth16/src/lib.rs:258-287) -> bytes"""
        lib = _lib.load()
        n = ctypes.c_size_t()
        check(lib.bh_groth16_params_write(self._h, None, 0, ctypes.byref(n)), "Parameters.write")
        out = ctypes.create_string_buffer(n.value)
        check(lib.bh_groth16_params_write(self._h, ctypes.cast(out, ctypes.c_void_p), n.value, ctypes.byref(n)), "Parameters.write")
        return out.raw   # immutable, hashable bytes like the reference's Vec<u8> handed to a writer; write_into avoids the copy

    def serialized_len(self):
        """byte length of Parameters::write's output"""
        n = ctypes.c_size_t()
        check(_lib.load().bh_groth16_params_write(self._h, None, 0, ctypes.byref(n)), "Parameters.write")
        return n.value

    def write_into(self, buffer):
        """Parameters::write into a caller-provided writable buffer (bytearray, numpy uint8, mmap ...): no second copy of a
        CRS of half a gigabyte.  Returns the number of bytes written; the buffer must hold at least that many."""
        lib = _lib.load()
        n = ctypes.c_size_t()
        check(lib.bh_groth16_params_write(self._h, None, 0, ctypes.byref(n)), "Parameters.write")
        mv = memoryview(buffer).cast("B")
        if mv.readonly or len(mv) < n.value:
            raise ValueError("write_into needs a writable buffer of at least %d bytes" % n.value)
        view = (ctypes.c_ubyte * len(mv)).from_buffer(mv)
        check(lib.bh_groth16_params_write(self._h, ctypes.cast(view, ctypes.c_void_p), n.value, ctypes.byref(n)), "Parameters.write")
        del view
        return n.value

    def vk_ext(self):
        """gamma_g2 ([24] uint64) and ic ([n,12] uint64): the verifier-side key elements"""
        lib = _lib.load()
        n = ctypes.c_size_t()
        gamma = np.zeros(24, dtype=np.uint64)
        check(lib.bh_groth16_params_vk_ext(self._h, gamma.ctypes.data_as(ctypes.c_void_p), None, 0, ctypes.byref(n)), "vk_ext")
        ic = np.zeros((n.value, 12), dtype=np.uint64)
        check(lib.bh_groth16_params_vk_ext(self._h, None, ic.ctypes.data_as(ctypes.c_void_p), n.value, None), "vk_ext")
        return gamma, ic

    def query(self, which):
        """which in h, l, a, b_g1, b_g2 -> affine Montgomery records of that query, read back from HBM"""
        lib = _lib.load()
        idx = ("h", "l", "a", "b_g1", "b_g2").index(which)
        b, n = ctypes.c_void_p(), ctypes.c_size_t()
        check(lib.bh_groth16_params_query(self._h, idx, ctypes.byref(b), ctypes.byref(n)), "Parameters.query")
        out = np.zeros((n.value, 24 if idx == 4 else 12), dtype=np.uint64)
        check(lib.bh_bases_download(self.worker.ctx, b, 0, n.value, out.ctypes.data_as(ctypes.c_void_p)), "Parameters.query")
        return out

    def bases(self, which):
        """the device-resident query itself as a (borrowed) Bases handle: get_h / get_l / get_a / get_b_g1 / get_b_g2 of
        ParameterSource (groth16/src/lib.rs:443-473) without the offset - for multiexps issued by the caller"""
        from .multiexp import Bases

        idx = ("h", "l", "a", "b_g1", "b_g2").index(which)
        b, n = ctypes.c_void_p(), ctypes.c_size_t()
        check(_lib.load().bh_groth16_params_query(self._h, idx, ctypes.byref(b), ctypes.byref(n)), "Parameters.bases")
        hb = Bases.__new__(Bases)
        hb.worker, hb.group, hb.n = self.worker, 2 if idx == 4 else 1, n.value
        hb._h = b
        hb.release = lambda: None   # owned by the parameters
        return hb

    def vk(self):
        """alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2 as affine Montgomery records"""
        outs = [np.zeros(w, dtype=np.uint64) for w in (12, 12, 24, 12, 24)]
        check(_lib.load().bh_groth16_params_vk(self._h, *[o.ctypes.data_as(ctypes.c_void_p) for o in outs]), "Parameters.vk")
        return outs

    def release(self):
        if self._h:
            _lib.load().bh_groth16_params_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def prove_assignment(prover, params, r, s, timings=None):
    """prover.rs:217-360 on a synthesised ProvingAssignment."""
    lib = _lib.load()
    a, b, c = (fr_to_mont_array(v) for v in (prover.a, prover.b, prover.c))
    ia, aa = fr_to_mont_array(prover.input_assignment), fr_to_mont_array(prover.aux_assignment)
    d1, d2, d3 = prover.a_aux_density.words(), prover.b_input_density.words(), prover.b_aux_density.words()
    rs = fr_to_mont_array([r, s])
    out = np.zeros(48, dtype=np.uint64)
    tm = (ctypes.c_float * 4)()
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    check(lib.bh_groth16_prove_assignment(params._h, p(a), p(b), p(c), a.shape[0], p(ia), ia.shape[0], p(aa), aa.shape[0],
                                          p(d1), p(d2), p(d3), p(rs[0:1]), p(rs[1:2]), p(out), tm), "create_proof")
    if timings is not None:
        timings[:] = list(tm)
    return Proof(out)


def create_proof(circuit, params, r, s, timings=None):
    """prover.rs:182-215: `circuit` is a callable circuit(cs) (the Circuit::synthesize body)."""
    prover = ProvingAssignment()
    prover.alloc_input(lambda: 1)
    circuit(prover)
    for i in range(len(prover.input_assignment)):
        prover.enforce(lambda lc, i=i: lc + Variable(INPUT, i), lambda lc: lc, lambda lc: lc)
    return prove_assignment(prover, params, r, s, timings)


def create_random_proof(circuit, params, rng=None, r1cs=None):
    """prover.rs:164-180: r and s drawn uniformly from Fr, then create_proof.  `rng` is any object with
    randrange (random.Random, random.SystemRandom); default: the operating system's CSPRNG.  With
    `r1cs` the constraint evaluation runs on the device (create_proof_r1cs)."""
    import random

    rng = rng or random.SystemRandom()
    r, s = rng.randrange(Q), rng.randrange(Q)
    if r1cs is not None:
        return create_proof_r1cs(circuit, r1cs, params, r, s)
    return create_proof(circuit, params, r, s)


def create_proof_demo(params, kind, size, seed, witness, constants, r, s, timings=None):
    """create_proof on one of the C++ demo circuits (groth16_capi.cpp): 0 = MiMCDemo, 1 = chain."""
    lib = _lib.load()
    wit = fr_to_mont_array(witness if isinstance(witness, np.ndarray) else list(witness))
    con = fr_to_mont_array(constants if isinstance(constants, np.ndarray) else list(constants)) if constants is not None \
        else np.zeros((1, 4), dtype=np.uint64)
    rs = fr_to_mont_array([r, s])
    out = np.zeros(48, dtype=np.uint64)
    tm = (ctypes.c_float * 4)()
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    check(lib.bh_groth16_prove_demo(params._h, kind, size, seed, p(wit), p(con), p(rs[0:1]), p(rs[1:2]), p(out), tm),
          "create_proof")
    if timings is not None:
        timings[:] = list(tm)
    return Proof(out)


# ---- constraint matrices resident in HBM -------------------------------------------


class ShapeAssembly(ConstraintSystem):
    """Structure-only ConstraintSystem: records the three matrices, never calls the value closures
    (what generator.rs:43-131 KeypairAssembly does when the CRS for the circuit is built)."""

    def __init__(self):
        self.num_inputs = self.num_aux = 0
        self.rows = ([], [], [])  # per matrix: list of rows, each a list of (kind, idx, coeff)

    def alloc(self, f):
        self.num_aux += 1
        return Variable(AUX, self.num_aux - 1)

    def alloc_input(self, f):
        self.num_inputs += 1
        return Variable(INPUT, self.num_inputs - 1)

    def enforce(self, a, b, c):
        z = LinearCombination.zero()
        for m, lc in enumerate((a(z), b(z), c(z))):
            self.rows[m].append([(v.kind, v.idx, k) for v, k in lc.terms if k != 0])  # prover.rs:31

    @staticmethod
    def capture(circuit):
        cs = ShapeAssembly()
        cs.alloc_input(lambda: 1)
        circuit(cs)
        for i in range(cs.num_inputs):  # prover.rs:208-215
            cs.enforce(lambda lc, i=i: lc + Variable(INPUT, i), lambda lc: lc, lambda lc: lc)
        return cs

    def csr(self):
        """-> (row_ptr, var, coeff_index) x3 as uint32 arrays, coefficient table (ints, [0] == 1)."""
        table, index = [1], {1: 0}
        out = []
        for rows in self.rows:
            row_ptr, var, coeff = [0], [], []
            for row in rows:
                for kind, idx, k in row:
                    if k not in index:
                        index[k] = len(table)
                        table.append(k)
                    var.append(idx if kind == INPUT else self.num_inputs + idx)
                    coeff.append(index[k])
                row_ptr.append(len(var))
            out.append(tuple(np.asarray(x, dtype=np.uint32) for x in (row_ptr, var, coeff)))
        return out, table


class _Csr(ctypes.Structure):
    _fields_ = [("row_ptr", ctypes.c_void_p), ("var", ctypes.c_void_p), ("coeff", ctypes.c_void_p)]


class R1CS:
    """The circuit's A/B/C matrices registered on the device (bh_r1cs)."""

    def __init__(self, worker, handle):
        self.worker, self._h = worker, handle
        lib = _lib.load()
        n = [ctypes.c_size_t() for _ in range(3)]
        check(lib.bh_r1cs_shape(self._h, *[ctypes.byref(x) for x in n]), "R1CS")
        self.num_inputs, self.num_aux, self.num_constraints = (x.value for x in n)

    @staticmethod
    def from_csr(worker, num_inputs, num_aux, matrices, coeff_table):
        lib = _lib.load()
        n_cons = len(matrices[0][0]) - 1
        keep = [tuple(np.ascontiguousarray(x, dtype=np.uint32) for x in m) for m in matrices]
        abc = (_Csr * 3)(*[_Csr(*[x.ctypes.data for x in m]) for m in keep])
        coeffs = fr_to_mont_array(list(coeff_table))
        h = ctypes.c_void_p()
        check(lib.bh_r1cs_create(worker.ctx, num_inputs, num_aux, n_cons, ctypes.byref(abc),
                                 coeffs.ctypes.data_as(ctypes.c_void_p), coeffs.shape[0], ctypes.byref(h)), "R1CS")
        return R1CS(worker, h)

    @staticmethod
    def from_circuit(worker, circuit):
        cs = ShapeAssembly.capture(circuit)
        matrices, table = cs.csr()
        return R1CS.from_csr(worker, cs.num_inputs, cs.num_aux, matrices, table)

    @staticmethod
    def from_demo(worker, kind, size, seed=0, constants=None):
        lib = _lib.load()
        con = fr_to_mont_array(list(constants)) if constants is not None else np.zeros((1, 4), dtype=np.uint64)
        h = ctypes.c_void_p()
        check(lib.bh_groth16_demo_r1cs(worker.ctx, kind, size, seed, con.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h)), "R1CS")
        return R1CS(worker, h)

    def density(self, which):
        """which: 0 a_aux, 1 b_input, 2 b_aux -> (bool array, total)"""
        lib = _lib.load()
        host, total = ctypes.c_void_p(), ctypes.c_size_t()
        check(lib.bh_r1cs_density(self._h, which, None, ctypes.byref(host), ctypes.byref(total)), "R1CS")
        n = self.num_inputs if which == 1 else self.num_aux
        nw = (n + 63) // 64
        words = np.ctypeslib.as_array(ctypes.cast(host, ctypes.POINTER(ctypes.c_uint64)), shape=(max(nw, 1),)).copy()
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:n].astype(bool)
        return bits, total.value

    def eval(self, input_assignment, aux_assignment):
        """a, b, c evaluations ([m,4] uint64 Montgomery, m = next power of two) - parity-test hook."""
        lib = _lib.load()
        ctx = self.worker.ctx
        log_m = 0
        while (1 << log_m) < self.num_constraints:
            log_m += 1
        m = 1 << log_m
        ia, aa = fr_to_mont_array(list(input_assignment)), fr_to_mont_array(list(aux_assignment))
        bufs = []
        for nbytes in (ia.nbytes + 32, aa.nbytes + 32, m * 32, m * 32, m * 32):
            d = ctypes.c_void_p()
            check(lib.bh_dev_alloc(ctx, nbytes, ctypes.byref(d)), "R1CS.eval")
            bufs.append(d)
        try:
            check(lib.bh_dev_upload(ctx, bufs[0], ia.ctypes.data_as(ctypes.c_void_p), ia.nbytes), "R1CS.eval")
            if aa.nbytes:
                check(lib.bh_dev_upload(ctx, bufs[1], aa.ctypes.data_as(ctypes.c_void_p), aa.nbytes), "R1CS.eval")
            check(lib.bh_r1cs_eval_dev(ctx, self._h, bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], log_m, None), "R1CS.eval")
            outs = []
            for d in bufs[2:]:
                o = np.zeros((m, 4), dtype=np.uint64)
                check(lib.bh_dev_download(ctx, o.ctypes.data_as(ctypes.c_void_p), d, o.nbytes), "R1CS.eval")
                outs.append(o)
        finally:
            for d in bufs:
                lib.bh_dev_free(ctx, d)
        return outs

    def release(self):
        if self._h:
            _lib.load().bh_r1cs_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class WitnessAssignment(ConstraintSystem):
    """prover.rs:57-162 reduced to witness generation (enforce is a no-op)."""

    def __init__(self):
        self.input_assignment, self.aux_assignment = [], []

    def alloc(self, f):
        self.aux_assignment.append(f() % Q)
        return Variable(AUX, len(self.aux_assignment) - 1)

    def alloc_input(self, f):
        self.input_assignment.append(f() % Q)
        return Variable(INPUT, len(self.input_assignment) - 1)

    def enforce(self, a, b, c):
        pass


def prove_witness(r1cs, params, input_assignment, aux_assignment, r, s, timings=None):
    lib = _lib.load()
    ia, aa = fr_to_mont_array(list(input_assignment)), fr_to_mont_array(list(aux_assignment))
    rs = fr_to_mont_array([r, s])
    out = np.zeros(48, dtype=np.uint64)
    tm = (ctypes.c_float * 4)()
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    check(lib.bh_groth16_prove_witness(params._h, r1cs._h, p(ia), ia.shape[0], p(aa), aa.shape[0], p(rs[0:1]), p(rs[1:2]),
                                       p(out), tm), "create_proof")
    if timings is not None:
        timings[:] = list(tm)
    return Proof(out)


def create_proof_r1cs(circuit, r1cs, params, r, s, timings=None):
    """create_proof with the constraint evaluation on the device: only `circuit`'s value closures run here."""
    w = WitnessAssignment()
    w.alloc_input(lambda: 1)
    circuit(w)
    return prove_witness(r1cs, params, w.input_assignment, w.aux_assignment, r, s, timings)


def create_proof_demo_r1cs(params, r1cs, kind, size, seed, witness, constants, r, s, timings=None):
    lib = _lib.load()
    wit = fr_to_mont_array(witness if isinstance(witness, np.ndarray) else list(witness))
    con = fr_to_mont_array(constants if isinstance(constants, np.ndarray) else list(constants)) if constants is not None \
        else np.zeros((1, 4), dtype=np.uint64)
    rs = fr_to_mont_array([r, s])
    out = np.zeros(48, dtype=np.uint64)
    tm = (ctypes.c_float * 4)()
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    check(lib.bh_groth16_prove_demo_r1cs(params._h, r1cs._h, kind, size, seed, p(wit), p(con), p(rs[0:1]), p(rs[1:2]),
                                         p(out), tm), "create_proof")
    if timings is not None:
        timings[:] = list(tm)
    return Proof(out)


# ---- one proof over several GPUs ------------------------------------------------------
SUMS_WORDS = 120  # BH_MSM_SUMS_BYTES / 8


def prove_witness_part(r1cs, params, input_assignment, aux_assignment, part, parts, timings=None):
    """The eight multiexp results of create_proof over part `part` of `parts` of the scalar indices
    -> uint64[120] (a_inputs, a_aux, b_g1_inputs, b_g1_aux | b_g2_inputs, b_g2_aux | h, l)."""
    lib = _lib.load()
    ia, aa = fr_to_mont_array(list(input_assignment)), fr_to_mont_array(list(aux_assignment))
    out = np.zeros(SUMS_WORDS, dtype=np.uint64)
    tm = (ctypes.c_float * 4)()
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    check(lib.bh_groth16_prove_witness_part(params._h, r1cs._h, p(ia), ia.shape[0], p(aa), aa.shape[0], part, parts, p(out), tm),
          "create_proof")
    if timings is not None:
        timings[:] = list(tm)
    return out


def prove_demo_part(params, r1cs, kind, size, seed, witness, constants, part, parts, timings=None):
    lib = _lib.load()
    wit = fr_to_mont_array(list(witness))
    con = fr_to_mont_array(list(constants)) if constants is not None else np.zeros((1, 4), dtype=np.uint64)
    out = np.zeros(SUMS_WORDS, dtype=np.uint64)
    tm = (ctypes.c_float * 4)()
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    check(lib.bh_groth16_prove_demo_r1cs_part(params._h, r1cs._h, kind, size, seed, p(wit), p(con), part, parts, p(out), tm),
          "create_proof")
    if timings is not None:
        timings[:] = list(tm)
    return out


def sums_add(acc, other):
    """slot-wise group addition of two multiexp-result records (host code)"""
    acc = np.ascontiguousarray(acc, dtype=np.uint64).copy()
    other = np.ascontiguousarray(other, dtype=np.uint64)
    _lib.load().bh_groth16_sums_add(acc.ctypes.data_as(ctypes.c_void_p), other.ctypes.data_as(ctypes.c_void_p))
    return acc


def assemble(params, sums, r, s):
    """prover.rs:326-360 from the (summed) multiexp results"""
    rs = fr_to_mont_array([r, s])
    sums = np.ascontiguousarray(sums, dtype=np.uint64)
    out = np.zeros(48, dtype=np.uint64)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    check(_lib.load().bh_groth16_assemble(params._h, p(sums), p(rs[0:1]), p(rs[1:2]), p(out)), "create_proof")
    return Proof(out)


# ---- the reference's call sites as the Rust shim issues them (shim/patches, csrc/groth16_callsites.cpp) ----------


def demo_assignment(kind, size, seed, witness, constants=None):
    """The ProvingAssignment create_proof synthesises for a C++ demo circuit (prover.rs:182-215), as numpy arrays:
    dict(a, b, c, input_assignment, aux_assignment: [n,4] uint64 Montgomery; a_aux_density, b_input_density,
    b_aux_density: LSB0 uint64 words).  Host only (test hook)."""
    lib = _lib.load()
    wit = fr_to_mont_array(witness if isinstance(witness, np.ndarray) else list(witness))
    con = fr_to_mont_array(constants if isinstance(constants, np.ndarray) else list(constants)) if constants is not None \
        else np.zeros((1, 4), dtype=np.uint64)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    counts = (ctypes.c_size_t * 3)()
    check(lib.bh_test_demo_assignment(kind, size, seed, p(wit), p(con), counts, None, None, None, None, None, None, None, None))
    nc, ni, na = counts[0], counts[1], counts[2]
    out = dict(a=np.zeros((nc, 4), np.uint64), b=np.zeros((nc, 4), np.uint64), c=np.zeros((nc, 4), np.uint64),
               input_assignment=np.zeros((ni, 4), np.uint64), aux_assignment=np.zeros((na, 4), np.uint64),
               a_aux_density=np.zeros((na + 63) // 64 + 1, np.uint64), b_input_density=np.zeros((ni + 63) // 64 + 1, np.uint64),
               b_aux_density=np.zeros((na + 63) // 64 + 1, np.uint64))
    check(lib.bh_test_demo_assignment(kind, size, seed, p(wit), p(con), counts, p(out["a"]), p(out["b"]), p(out["c"]),
                                      p(out["input_assignment"]), p(out["aux_assignment"]), p(out["a_aux_density"]),
                                      p(out["b_input_density"]), p(out["b_aux_density"])))
    return out


def prove_assignment_arrays(params, asg, r, s, timings=None):
    """bh_groth16_prove_assignment on the arrays of demo_assignment (already Montgomery)."""
    lib = _lib.load()
    rs = fr_to_mont_array([r, s])
    out = np.zeros(48, dtype=np.uint64)
    tm = (ctypes.c_float * 4)()
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    check(lib.bh_groth16_prove_assignment(params._h, p(asg["a"]), p(asg["b"]), p(asg["c"]), asg["a"].shape[0],
                                          p(asg["input_assignment"]), asg["input_assignment"].shape[0], p(asg["aux_assignment"]),
                                          asg["aux_assignment"].shape[0], p(asg["a_aux_density"]), p(asg["b_input_density"]),
                                          p(asg["b_aux_density"]), p(rs[0:1]), p(rs[1:2]), p(out), tm), "create_proof")
    if timings is not None:
        timings[:] = list(tm)
    return Proof(out)


def prove_via_call_sites(params, asg, r, s, patched_prover, timings=None):
    """The h block + eight multiexps issued as bellman's create_proof issues them through the Rust shim:
    patched_prover=True  - groth16/src/prover.rs patched (scalars registered once, h block in one device call);
    patched_prover=False - only multiexp.rs / domain.rs patched (7 host FFT round trips, host pointwise passes, serial
                           Fr -> Exponent, 8 multiexps each uploading its scalars);
    patched_prover="resident" - the same patch level with the EvaluationDomain's vector kept in HBM between its calls,
                           `Exponent::from` deferred (no conversion on the host) and each shared exponent vector
                           uploaded once (csrc/groth16_callsites.cpp mode 2).
    timings: [issue + waits, total] host ms."""
    lib = _lib.load()
    rs = fr_to_mont_array([r, s])
    out = np.zeros(48, dtype=np.uint64)
    tm = (ctypes.c_float * 2)()
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    mode = 2 if patched_prover == "resident" else 1 if patched_prover else 0
    check(lib.bh_test_groth16_prove_via_call_sites(params._h, mode, p(asg["a"]), p(asg["b"]), p(asg["c"]),
                                                   asg["a"].shape[0], p(asg["input_assignment"]), asg["input_assignment"].shape[0],
                                                   p(asg["aux_assignment"]), asg["aux_assignment"].shape[0], p(asg["a_aux_density"]),
                                                   p(asg["b_input_density"]), p(asg["b_aux_density"]), p(rs[0:1]), p(rs[1:2]), p(out), tm),
          "create_proof (call sites)")
    if timings is not None:
        timings[:] = list(tm)
    return Proof(out)


def create_proof_demo_async(params, r1cs, kind, size, seed, witness, constants, r, s):
    """bh_groth16_prove_demo_async: synthesis of the C++ demo circuit on this thread, the device part on a helper thread.
    Returns wait(timings=None) -> Proof.  r1cs=None: host synthesis as in the reference."""
    lib = _lib.load()
    wit = fr_to_mont_array(witness if isinstance(witness, np.ndarray) else list(witness))
    con = fr_to_mont_array(constants if isinstance(constants, np.ndarray) else list(constants)) if constants is not None \
        else np.zeros((1, 4), dtype=np.uint64)
    rs = fr_to_mont_array([r, s])
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    job = ctypes.c_void_p()
    check(lib.bh_groth16_prove_demo_async(params._h, None if r1cs is None else r1cs._h, kind, size, seed, p(wit), p(con),
                                          p(rs[0:1]), p(rs[1:2]), ctypes.byref(job)), "create_proof (async)")

    def wait(timings=None):
        out = np.zeros(48, dtype=np.uint64)
        tm = (ctypes.c_float * 4)()
        check(lib.bh_groth16_proof_wait(job, p(out), tm), "create_proof (async wait)")
        if timings is not None:
            timings[:] = list(tm)
        return Proof(out)

    return wait


def _proof_waiter(lib, job, keep):
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731

    def wait(timings=None):
        _alive = keep  # noqa: F841  (the arrays the device part reads in place stay referenced until the wait)
        out = np.zeros(48, dtype=np.uint64)
        tm = (ctypes.c_float * 4)()
        check(lib.bh_groth16_proof_wait(job, p(out), tm), "create_proof (async wait)")
        if timings is not None:
            timings[:] = list(tm)
        return Proof(out)

    return wait


def prove_assignment_arrays_async(params, asg, r, s):
    """bh_groth16_prove_assignment_async on the arrays of demo_assignment: the device part of create_proof
    (prover.rs:217-360) on a helper thread; the arrays are read in place until the returned wait() has been called."""
    lib = _lib.load()
    rs = fr_to_mont_array([r, s])
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    job = ctypes.c_void_p()
    check(lib.bh_groth16_prove_assignment_async(params._h, p(asg["a"]), p(asg["b"]), p(asg["c"]), asg["a"].shape[0],
                                                p(asg["input_assignment"]), asg["input_assignment"].shape[0],
                                                p(asg["aux_assignment"]), asg["aux_assignment"].shape[0], p(asg["a_aux_density"]),
                                                p(asg["b_input_density"]), p(asg["b_aux_density"]), p(rs[0:1]), p(rs[1:2]),
                                                ctypes.byref(job)), "create_proof (async)")
    return _proof_waiter(lib, job, (asg, rs))


def prove_witness_async(r1cs, params, input_assignment, aux_assignment, r, s):
    """bh_groth16_prove_witness_async: the witness vectors (Montgomery [n,4] uint64 arrays, or lists of ints) are copied
    before this returns; the device part runs on a helper thread.  Returns wait(timings=None) -> Proof."""
    lib = _lib.load()
    ia = input_assignment if isinstance(input_assignment, np.ndarray) else fr_to_mont_array(list(input_assignment))
    aa = aux_assignment if isinstance(aux_assignment, np.ndarray) else fr_to_mont_array(list(aux_assignment))
    rs = fr_to_mont_array([r, s])
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    job = ctypes.c_void_p()
    check(lib.bh_groth16_prove_witness_async(params._h, r1cs._h, p(ia), ia.shape[0], p(aa), aa.shape[0], p(rs[0:1]), p(rs[1:2]),
                                             ctypes.byref(job)), "create_proof (async)")
    return _proof_waiter(lib, job, (r1cs,))
