"""`Worker` / `Waiter` mirror (reference: src/multicore.rs:21-118).

On the MI355X a Worker is a device context (one per GPU per process): `compute` becomes
"enqueue on a HIP stream", the Waiter's `wait()` blocks on that stream.
"""

import ctypes

from . import _lib
from .errors import check


class Worker:
    def __init__(self, device=0):
        self._lib = _lib.load()
        ctx = ctypes.c_void_p()
        check(self._lib.bh_ctx_create(device, ctypes.byref(ctx)), "Worker::new")
        self._ctx = ctx
        self.device = device

    @property
    def ctx(self):
        return self._ctx

    def log_num_threads(self):
        """multicore.rs:29-31 analogue: floor(log2(#CUs))."""
        return self._lib.bh_ctx_log_num_cus(self._ctx)

    def synchronize(self):
        check(self._lib.bh_ctx_synchronize(self._ctx))

    def set_limits(self, max_jobs_in_flight=0, pool_cap_bytes=None, table_budget_bytes=None, fft_table_budget_bytes=None):
        """bh_ctx_set_limits: jobs in flight before an issuing thread completes the oldest one itself (the
        Worker::compute back-pressure of src/multicore.rs:47-73), cap of the workspace pool, budget of automatic
        window tables, budget of the cached per-size FFT tables.  None = leave unchanged."""
        keep = ctypes.c_size_t(-1).value
        check(self._lib.bh_ctx_set_limits(self._ctx, max_jobs_in_flight, keep if pool_cap_bytes is None else pool_cap_bytes,
                                          keep if table_budget_bytes is None else table_budget_bytes,
                                          keep if fft_table_budget_bytes is None else fft_table_budget_bytes))

    def info(self):
        """bh_ctx_info as a dict (device, CUs, HBM, hardware-queue request, jobs in flight, pool and table bytes)."""
        class _Info(ctypes.Structure):
            _fields_ = [("device", ctypes.c_int32), ("num_cus", ctypes.c_uint32), ("hbm_bytes", ctypes.c_uint64),
                        ("hw_queues_requested", ctypes.c_uint32), ("hw_queues_set_before_hip_init", ctypes.c_uint32),
                        ("max_jobs_in_flight", ctypes.c_uint32), ("jobs_in_flight", ctypes.c_uint32),
                        ("pool_bytes_held", ctypes.c_uint64), ("pool_bytes_idle", ctypes.c_uint64),
                        ("table_bytes", ctypes.c_uint64), ("table_budget", ctypes.c_uint64),
                        ("fft_table_bytes", ctypes.c_uint64), ("fft_table_budget", ctypes.c_uint64)]
        i = _Info()
        check(self._lib.bh_ctx_info(self._ctx, ctypes.byref(i)))
        return {k: getattr(i, k) for k, _ in _Info._fields_}

    # ---- host-side task helpers (multicore.rs:33-91): device work goes through multiexp /
    # EvaluationDomain; these exist for caller code that used the pool for its own host tasks ----
    _pool = None

    @classmethod
    def _host_pool(cls):
        if cls._pool is None:
            import os
            from concurrent.futures import ThreadPoolExecutor

            cls._host_threads = os.cpu_count() or 1
            cls._pool = ThreadPoolExecutor(max_workers=cls._host_threads)
        return cls._pool

    def compute(self, f):
        """multicore.rs:33-76: run `f` on the host pool, return a Waiter for its result."""
        fut = self._host_pool().submit(f)
        return Waiter(fn=fut.result)

    def scope(self, elements, f):
        """multicore.rs:78-91: f(scope, chunk_size) with chunk_size = 1 if elements < threads else
        elements // threads; `scope.spawn(g)` schedules g(scope) and every spawned task has finished
        when this call returns."""
        self._host_pool()
        n = self._host_threads
        chunk = 1 if elements < n else elements // n
        sc = _Scope(self._host_pool())
        try:
            out = f(sc, chunk)
        finally:
            sc.join()
        return out

    def trim(self):
        """give idle cached device memory (job workspaces, FFT tables) back to the driver"""
        check(self._lib.bh_ctx_trim(self._ctx))

    def close(self):
        if self._ctx:
            self._lib.bh_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- raw device buffers (inputs resident in HBM) ----
    def alloc(self, nbytes):
        p = ctypes.c_void_p()
        check(self._lib.bh_dev_alloc(self._ctx, nbytes, ctypes.byref(p)))
        return p

    def free(self, p):
        check(self._lib.bh_dev_free(self._ctx, p))

    def upload(self, dev, arr):
        check(self._lib.bh_dev_upload(self._ctx, dev, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes))

    def download(self, arr, dev):
        check(self._lib.bh_dev_download(self._ctx, arr.ctypes.data_as(ctypes.c_void_p), dev, arr.nbytes))


class _Scope:
    """rayon::Scope stand-in for Worker.scope"""

    def __init__(self, pool):
        self._pool, self._futs = pool, []

    def spawn(self, g):
        self._futs.append(self._pool.submit(g, self))

    def join(self):
        i = 0
        while i < len(self._futs):   # tasks may spawn further tasks
            self._futs[i].result()
            i += 1


class Waiter:
    """multicore.rs:94-118.  `wait()` consumes the waiter."""

    def __init__(self, fn=None, value=None):
        self._fn = fn
        self._value = value

    def wait(self):
        if self._fn is not None:
            fn, self._fn = self._fn, None
            self._value = fn()
        return self._value

    @staticmethod
    def done(val):
        return Waiter(value=val)
