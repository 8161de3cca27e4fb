"""Line-by-line restatement of /root/reference/src/domain.rs for the
`Scalar<S>` instantiation (elements are ints mod r; `Point<G>` is unused by
every caller in the reference and is out of scope).
"""

from .errors import PolynomialDegreeTooLarge


def bitreverse(n, l):
    """domain.rs:273-280"""
    r = 0
    for _ in range(l):
        r = (r << 1) | (n & 1)
        n >>= 1
    return r


def serial_fft(a, r, omega, log_n):
    """domain.rs:272-314 (in place on list `a`)."""
    n = len(a)
    assert n == 1 << log_n
    for k in range(n):
        rk = bitreverse(k, log_n)
        if k < rk:
            a[rk], a[k] = a[k], a[rk]
    m = 1
    for _ in range(log_n):
        w_m = pow(omega, n // (2 * m), r)
        k = 0
        while k < n:
            w = 1
            for j in range(m):
                t = (a[k + j + m] * w) % r
                tmp = (a[k + j] - t) % r
                a[k + j + m] = tmp
                a[k + j] = (a[k + j] + t) % r
                w = (w * w_m) % r
            k += 2 * m
        m *= 2


def parallel_fft(a, r, omega, log_n, log_cpus):
    """domain.rs:316-372 (executed serially; the split is what matters)."""
    assert log_n >= log_cpus
    num_cpus = 1 << log_cpus
    log_new_n = log_n - log_cpus
    tmp = [[0] * (1 << log_new_n) for _ in range(num_cpus)]
    new_omega = pow(omega, num_cpus, r)
    for j in range(num_cpus):
        omega_j = pow(omega, j, r)
        omega_step = pow(omega, j << log_new_n, r)
        elt = 1
        t_j = tmp[j]
        for i in range(1 << log_new_n):
            for s in range(num_cpus):
                idx = (i + (s << log_new_n)) % (1 << log_n)
                t_j[i] = (t_j[i] + a[idx] * elt) % r
                elt = (elt * omega_step) % r
            elt = (elt * omega_j) % r
        serial_fft(t_j, r, new_omega, log_new_n)
    mask = (1 << log_cpus) - 1
    for idx in range(len(a)):
        a[idx] = tmp[idx & mask][idx >> log_cpus]


def best_fft(a, r, worker, omega, log_n):
    """domain.rs:261-269"""
    log_cpus = worker.log_num_threads()
    if log_n <= log_cpus:
        serial_fft(a, r, omega, log_n)
    else:
        parallel_fft(a, r, omega, log_n, log_cpus)


class EvaluationDomain:
    def __init__(self, field, coeffs, exp, omega):
        self.F = field
        self.r = field.r
        self.coeffs = coeffs
        self.exp = exp
        self.omega = omega
        self.omegainv = field.inv(omega)
        self.geninv = field.inv(field.MULTIPLICATIVE_GENERATOR)
        self.minv = field.inv(len(coeffs) % field.r)

    @classmethod
    def from_coeffs(cls, field, coeffs):
        """domain.rs:47-79"""
        coeffs = list(coeffs)
        m = 1
        exp = 0
        while m < len(coeffs):
            m *= 2
            exp += 1
            if exp >= field.S:
                raise PolynomialDegreeTooLarge()
        omega = field.ROOT_OF_UNITY
        for _ in range(exp, field.S):
            omega = (omega * omega) % field.r
        coeffs.extend([0] * (m - len(coeffs)))
        return cls(field, coeffs, exp, omega)

    def into_coeffs(self):
        return self.coeffs

    def fft(self, worker):
        """domain.rs:81-83"""
        best_fft(self.coeffs, self.r, worker, self.omega, self.exp)

    def ifft(self, worker):
        """domain.rs:85-99"""
        best_fft(self.coeffs, self.r, worker, self.omegainv, self.exp)
        minv = self.minv
        self.coeffs = [(v * minv) % self.r for v in self.coeffs]

    def distribute_powers(self, worker, g):
        """domain.rs:101-113 (chunked exactly as Worker::scope chunks)."""
        n = len(self.coeffs)
        chunk = worker.chunk_size(n)
        for i, start in enumerate(range(0, n, chunk)):
            u = pow(g, i * chunk, self.r)
            for k in range(start, min(start + chunk, n)):
                self.coeffs[k] = (self.coeffs[k] * u) % self.r
                u = (u * g) % self.r

    def coset_fft(self, worker):
        """domain.rs:115-118"""
        self.distribute_powers(worker, self.F.MULTIPLICATIVE_GENERATOR)
        self.fft(worker)

    def icoset_fft(self, worker):
        """domain.rs:120-125"""
        geninv = self.geninv
        self.ifft(worker)
        self.distribute_powers(worker, geninv)

    def z(self, tau):
        """domain.rs:129-134"""
        return (pow(tau, len(self.coeffs), self.r) - 1) % self.r

    def divide_by_z_on_coset(self, worker):
        """domain.rs:139-151"""
        i = self.F.inv(self.z(self.F.MULTIPLICATIVE_GENERATOR))
        self.coeffs = [(v * i) % self.r for v in self.coeffs]

    def mul_assign(self, worker, other):
        """domain.rs:154-170"""
        assert len(self.coeffs) == len(other.coeffs)
        self.coeffs = [(a * b) % self.r for a, b in zip(self.coeffs, other.coeffs)]

    def sub_assign(self, worker, other):
        """domain.rs:173-189"""
        assert len(self.coeffs) == len(other.coeffs)
        self.coeffs = [(a - b) % self.r for a, b in zip(self.coeffs, other.coeffs)]
