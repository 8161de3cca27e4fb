"""Restatement of /root/reference/src/multicore.rs (Worker / Waiter).

The oracle executes everything inline (like the reference's non-`multicore`
Worker, multicore.rs:145-213) but keeps the quantities that are OBSERVABLE in
the algorithms: `log_num_threads` (drives best_fft's serial/parallel split,
domain.rs:261-269) and `scope`'s chunk size (multicore.rs:78-91, drives
distribute_powers' per-chunk pow, domain.rs:101-113).
"""


def log2_floor(num):
    """multicore.rs:120-130"""
    assert num > 0
    pow_ = 0
    while (1 << (pow_ + 1)) <= num:
        pow_ += 1
    return pow_


class Waiter:
    """multicore.rs:94-118 - holds an already-computed value."""

    def __init__(self, val):
        self._val = val

    def wait(self):
        return self._val

    @staticmethod
    def done(val):
        return Waiter(val)


class Worker:
    def __init__(self, num_threads=8):
        self.num_threads = num_threads

    def log_num_threads(self):
        """multicore.rs:29-31"""
        return log2_floor(self.num_threads)

    def compute(self, f):
        """multicore.rs:33-76 (inline execution; result order is unobservable)."""
        return Waiter(f())

    def chunk_size(self, elements):
        """multicore.rs:83-88"""
        if elements < self.num_threads:
            return 1
        return elements // self.num_threads
