"""G1 multiexp over a window table stored at a 128-byte record stride (api.hip bh_bases::table_padded: 2^19 points and
more) against the classic plan over the same vector: identical results with full density, with a density map, with a skip,
with a forced chunk and with the forced LDS-accumulator variant of the kernel.   python tools/table_pad_check.py [log_n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bellman_amd  # noqa: E402
from bellman_amd import _lib  # noqa: E402
import importlib  # noqa: E402
mx = importlib.import_module("bellman_amd.multiexp")  # (the package attribute of that name is the function)
from profile_suite import make_bases  # noqa: E402
from bench import splitmix_scalars  # noqa: E402


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 19
    n = 1 << log_n
    lib = _lib.load()
    w = bellman_amd.Worker(0)
    dout = make_bases(w, lib, 1, n)
    bases = bellman_amd.Bases.copy_device(w, 1, dout, n)
    s = splitmix_scalars(n, 2)
    rng = np.random.default_rng(7)
    m = n - 4321
    dens = mx.DensityTracker(rng.integers(0, 2, m).astype(bool))
    cases = [("full", dict(density_map=mx.FullDensity(), exponents=s)),
             ("short+skip", dict(density_map=mx.FullDensity(), exponents=s[: n - 12345], skip=777)),
             ("density", dict(density_map=dens, exponents=s[:m], skip=5))]
    ref = {}
    for name, kw in cases:
        ref[name] = bytes(mx.multiexp(w, bases, flags=mx.NO_TABLE, **kw).wait())
    for c in (16, 20):
        bases.precompute(c)
        info = bases.table_info()
        assert info[2] == info[1] * n * (128 if os.environ.get("BELLMAN_HIP_TABLE_PAD", "1") != "0" and log_n >= 19 else 96), info
        for name, kw in cases:
            for flags, chunk in ((0, 0), (mx.ACC_LDS, 0), (0, 64)):
                r, ms = mx.multiexp(w, bases, timed=True, flags=flags, chunk=chunk, **kw).wait()
                ok = bytes(r) == ref[name]
                print("2^%d table c=%d (%d rows, %.2f GB) %-10s flags=%d K=%d: %s  (device %.3f ms)" %
                      (log_n, c, info[1], info[2] / 1e9, name, flags, chunk, "identical to the classic plan" if ok else "MISMATCH", ms[0]),
                      flush=True)
                assert ok
    print("table_pad_check ok")


main()
