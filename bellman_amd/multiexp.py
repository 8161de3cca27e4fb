"""`multiexp` mirror (reference: src/multiexp.rs:305-332) over bh_msm_async / bh_msm_wait.

Points are numpy uint64 records in the library's format (96 B G1 / 192 B G2 affine, Montgomery,
all-zero = identity); scalars are [n,4] uint64 little-endian, canonical unless `mont=True`.
"""

import ctypes
import os

import numpy as np

from . import _lib
from .errors import check
from .multicore import Waiter

G1, G2 = 1, 2
_WORDS = {G1: 12, G2: 24}


class FullDensity:
    """src/multiexp.rs:95-115"""

    def get_query_size(self):
        return None


class DensityTracker:
    """src/multiexp.rs:117-157 (bit i set <=> scalar i has a base in the compacted vector)."""

    def __init__(self, bits=None):
        self.bv = [] if bits is None else [bool(b) for b in bits]

    def add_element(self):
        self.bv.append(False)

    def inc(self, idx):
        self.bv[idx] = True

    def get_total_density(self):
        return sum(self.bv)

    def get_query_size(self):
        return len(self.bv)

    def words(self):
        n = len(self.bv)
        nwords = (n + 63) // 64
        padded = np.zeros(nwords * 64, dtype=np.uint8)
        padded[:n] = np.asarray(self.bv, dtype=np.uint8)
        return np.packbits(padded, bitorder="little").view(np.uint64).copy()


class Bases:
    """Device-resident `Arc<Vec<G::Affine>>` (src/multiexp.rs:45-52, groth16/src/lib.rs:443-473).
    `with_skip(k)` is the `(bases, k)` SourceBuilder."""

    def __init__(self, worker, group, points, stride=None, inf_offset=-1):
        self.worker = worker
        self.group = group
        lib = _lib.load()
        pts = np.ascontiguousarray(points)
        rec = _WORDS[group] * 8
        if stride is None:
            pts = pts.view(np.uint64).reshape(-1, _WORDS[group])
            n, stride = pts.shape[0], rec
        else:
            n = pts.nbytes // stride
        h = ctypes.c_void_p()
        check(
            lib.bh_bases_register(worker.ctx, group, pts.ctypes.data_as(ctypes.c_void_p), n, stride, inf_offset,
                                  ctypes.byref(h)),
            "bases_register",
        )
        self._h = h
        self.n = n
        self._auto_precompute()

    @classmethod
    def from_uncompressed(cls, worker, group, data):
        """Bases from bellman's serialized CRS bytes (`Parameters::write`, groth16/src/lib.rs:258-287):
        concatenated `to_uncompressed()` points; decoded on the device."""
        rec = _WORDS[group] * 8
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        assert buf.size % rec == 0
        self = cls.__new__(cls)
        self.worker, self.group, self.n = worker, group, buf.size // rec
        h = ctypes.c_void_p()
        check(_lib.load().bh_bases_register_uncompressed(worker.ctx, group, buf.ctypes.data_as(ctypes.c_void_p), self.n,
                                                         ctypes.byref(h)), "bases_register_uncompressed")
        self._h = h
        self._auto_precompute()
        return self

    @classmethod
    def read_uncompressed(cls, worker, group, data, checked=True, forbid_identity=True):
        """The per-point part of `Parameters::read(reader, checked)` (groth16/src/lib.rs:289-341):
        `from_uncompressed` (checked: on the curve and in the prime-order subgroup) or
        `from_uncompressed_unchecked`, plus the "point at infinity" rule; all on the device.
        Raises InvalidPoint / PointAtInfinity carrying `.index` = first offending point."""
        rec = _WORDS[group] * 8
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        assert buf.size % rec == 0
        self = cls.__new__(cls)
        self.worker, self.group, self.n = worker, group, buf.size // rec
        h, bad = ctypes.c_void_p(), ctypes.c_size_t(0)
        flags = (1 if checked else 0) | (2 if forbid_identity else 0)
        try:
            check(_lib.load().bh_bases_read_uncompressed(worker.ctx, group, buf.ctypes.data_as(ctypes.c_void_p), self.n, flags,
                                                         ctypes.byref(h), ctypes.byref(bad)), "bases_read_uncompressed")
        except IOError as e:
            e.index = bad.value
            raise
        self._h = h
        self._auto_precompute()
        return self

    def _auto_precompute(self):
        # BELLMAN_HIP_PRECOMPUTE=1 (or =<window bits>): build the window table at registration
        v = os.environ.get("BELLMAN_HIP_PRECOMPUTE", "")
        if v and v != "0" and self.n:
            self.precompute(0 if v == "1" else int(v))

    def precompute(self, window_bits=0):
        """Build the window table 2^(c*j) P_i next to the bases (W x the memory): multiexps over this
        vector then use one bucket set for all windows.  Same results, faster reduction."""
        check(_lib.load().bh_bases_precompute(self.worker.ctx, self._h, window_bits), "bases_precompute")
        return self

    def table_info(self):
        """(window bits, rows, bytes) of the window table; zeros without one"""
        c_, w_, b_ = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_size_t()
        check(_lib.load().bh_bases_table_info(self._h, ctypes.byref(c_), ctypes.byref(w_), ctypes.byref(b_)))
        return c_.value, w_.value, b_.value

    def download(self, first=0, count=None):
        """affine Montgomery records [count, 12|24] uint64 back from HBM"""
        count = self.n - first if count is None else count
        out = np.zeros((count, _WORDS[self.group]), dtype=np.uint64)
        check(_lib.load().bh_bases_download(self.worker.ctx, self._h, first, count, out.ctypes.data_as(ctypes.c_void_p)),
              "bases_download")
        return out

    @classmethod
    def wrap_device(cls, worker, group, dev_ptr, n):
        """A LIVE view of a device array of packed records (not owned, nothing snapshotted: no window table unless
        precompute() is called explicitly)."""
        self = cls.__new__(cls)
        self.worker, self.group, self.n = worker, group, n
        h = ctypes.c_void_p()
        check(_lib.load().bh_bases_wrap_dev(worker.ctx, group, dev_ptr, n, ctypes.byref(h)))
        self._h = h
        self._auto_precompute()
        return self

    @classmethod
    def copy_device(cls, worker, group, dev_ptr, n):
        """An owned handle holding a device-to-device copy of n packed records (registered like a CRS query: the
        automatic window table applies)."""
        self = cls.__new__(cls)
        self.worker, self.group, self.n = worker, group, n
        h = ctypes.c_void_p()
        check(_lib.load().bh_bases_copy_dev(worker.ctx, group, dev_ptr, n, ctypes.byref(h)))
        self._h = h
        self._auto_precompute()
        return self

    def __len__(self):
        return self.n

    def release(self):
        if self._h:
            _lib.load().bh_bases_release(self.worker.ctx, self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class MsmOpts(ctypes.Structure):
    """bh_msm_opts: per-job plan overrides (window bits c, chunk K, BH_MSM_* flags); zero = tuned default"""
    _fields_ = [("window_bits", ctypes.c_uint32), ("chunk", ctypes.c_uint32), ("flags", ctypes.c_uint32)]


ACC_REGISTERS, ACC_LDS, NO_TABLE, NO_SMALL_PATH, G2_SINGLE_LANE, G2_LANE_TRIPLES, STAGE_TIMES, HOLD, G2_LANE_PAIRS = 1, 2, 4, 8, 16, 32, 64, 128, 256


STATS_FIELDS = ("sorted_entries", "zero_digits", "mixed_additions", "chunk_lanes", "window_bits", "chunk", "bucket_sets", "digit_columns")


def multiexp(pool, bases, density_map, exponents, skip=0, mont=False, timed=False, scalars_dev=None, n=None,
             density_dev=None, window_bits=0, chunk=0, flags=0, stats=False):
    """multiexp(pool, (bases, skip), density_map, exponents) -> Waiter (src/multiexp.rs:305-332).

    exponents: [n,4] uint64 scalars on the host; or pass scalars_dev (device pointer) + n.
    The Waiter's wait() returns the affine result record (numpy uint64[12|24]) or raises the
    SynthesisError the reference would return.  With timed=True it returns (record, [total, sort, accumulate, reduce] device ms);
    with stats=True (implies timed) a third element: what the job executed, counted on the device (bh_msm_wait_stats,
    a dict over STATS_FIELDS).  window_bits / chunk / flags override the plan of THIS job only (tests, tuning sweeps)."""
    timed = timed or stats
    lib = _lib.load()
    words = None
    dlen = 0
    if isinstance(density_map, DensityTracker):
        dlen = density_map.get_query_size()
        words = density_map.words()
    job = ctypes.c_void_p()
    fmt = 1 if mont else 0
    opts = MsmOpts(window_bits, chunk, flags | (STAGE_TIMES if timed else 0))
    po = ctypes.cast(ctypes.pointer(opts), ctypes.c_void_p)
    if scalars_dev is None:
        sc = np.ascontiguousarray(exponents, dtype=np.uint64).reshape(-1, 4)
        n = sc.shape[0]
        rc = lib.bh_msm_async_opts(pool.ctx, bases._h, skip, sc.ctypes.data_as(ctypes.c_void_p), n, fmt,
                                   None if words is None else words.ctypes.data_as(ctypes.c_void_p), dlen, po,
                                   ctypes.byref(job))
    else:
        rc = lib.bh_msm_async_dev_opts(pool.ctx, bases._h, skip, scalars_dev, n, fmt, density_dev,
                                       dlen if density_dev is not None else 0, po, ctypes.byref(job))
    check(rc, "multiexp")
    w = _WORDS[bases.group]

    def finish():
        out = np.zeros(w, dtype=np.uint64)
        ms = (ctypes.c_float * 4)()
        if stats:
            st = (ctypes.c_uint64 * 8)()
            check(lib.bh_msm_wait_stats(job, out.ctypes.data_as(ctypes.c_void_p), ms, st), "multiexp.wait")
            return out, list(ms), dict(zip(STATS_FIELDS, (int(x) for x in st)))
        check(lib.bh_msm_wait_profile(job, out.ctypes.data_as(ctypes.c_void_p), ms), "multiexp.wait")
        return (out, list(ms)) if timed else out

    w_ = Waiter(fn=finish)
    w_.start = lambda: check(lib.bh_msm_start(job), "multiexp.start")   # for jobs issued with flags=HOLD
    return w_


class Scalars:
    """A scalar vector resident in HBM (bh_scalars_*): the `Arc<Vec<Exponent>>` that create_proof shares between
    several multiexps (groth16/src/prover.rs:267-318), uploaded once."""

    def __init__(self, worker, scalars, mont=False):
        sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        self.worker, self.n = worker, sc.shape[0]
        h = ctypes.c_void_p()
        check(_lib.load().bh_scalars_register(worker.ctx, sc.ctypes.data_as(ctypes.c_void_p), self.n, 1 if mont else 0,
                                              ctypes.byref(h)), "scalars_register")
        self._h = h

    def __len__(self):
        return self.n

    def release(self):
        if self._h:
            _lib.load().bh_scalars_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def multiexp_scalars(pool, bases, density_map, scalars, skip=0, first=0, n=None, window_bits=0, chunk=0, flags=0):
    """multiexp over scalars [first, first + n) of a registered vector (bh_msm_async_scalars) -> Waiter."""
    lib = _lib.load()
    n = len(scalars) - first if n is None else n
    words, dlen = None, 0
    if isinstance(density_map, DensityTracker):
        dlen = density_map.get_query_size()
        words = density_map.words()
    opts = MsmOpts(window_bits, chunk, flags)
    job = ctypes.c_void_p()
    check(lib.bh_msm_async_scalars(pool.ctx, bases._h, skip, scalars._h, first, n,
                                   None if words is None else words.ctypes.data_as(ctypes.c_void_p), dlen,
                                   ctypes.cast(ctypes.pointer(opts), ctypes.c_void_p), ctypes.byref(job)), "multiexp")
    w = _WORDS[bases.group]

    def finish():
        _keep = (words, scalars)  # noqa: F841  (alive until the job has been waited on)
        out = np.zeros(w, dtype=np.uint64)
        check(lib.bh_msm_wait(job, out.ctypes.data_as(ctypes.c_void_p)), "multiexp.wait")
        return out

    return Waiter(fn=finish)


def multiexp_sharded(workers, shards, density_map, exponents, skip=0, mont=False):
    """ONE multiexp over several contexts of this process (bh_msm_sharded_async): workers[k] / shards[k] = the
    context on GPU k and the k-th contiguous piece of the base vector registered on it -> Waiter."""
    lib = _lib.load()
    k = len(workers)
    assert k == len(shards) and k > 0
    sc = np.ascontiguousarray(exponents, dtype=np.uint64).reshape(-1, 4)
    words, dlen = None, 0
    if isinstance(density_map, DensityTracker):
        dlen = density_map.get_query_size()
        words = density_map.words()
    ctxs = (ctypes.c_void_p * k)(*[w.ctx for w in workers])
    hs = (ctypes.c_void_p * k)(*[b._h for b in shards])
    job = ctypes.c_void_p()
    check(lib.bh_msm_sharded_async(ctxs, hs, k, skip, sc.ctypes.data_as(ctypes.c_void_p), sc.shape[0], 1 if mont else 0,
                                   None if words is None else words.ctypes.data_as(ctypes.c_void_p), dlen, ctypes.byref(job)),
          "multiexp (sharded)")
    w = _WORDS[shards[0].group]

    def finish():
        _keep = (sc, words)  # noqa: F841
        out = np.zeros(w, dtype=np.uint64)
        check(lib.bh_msm_sharded_wait(job, out.ctypes.data_as(ctypes.c_void_p)), "multiexp.wait (sharded)")
        return out

    return Waiter(fn=finish)


def point_add(group, a, b):
    """Host-side affine addition of result records (folding per-GPU partial sums)."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.zeros_like(a)
    _lib.load().bh_point_add(group, out.ctypes.data_as(ctypes.c_void_p), a.ctypes.data_as(ctypes.c_void_p),
                             b.ctypes.data_as(ctypes.c_void_p), 1)
    return out
