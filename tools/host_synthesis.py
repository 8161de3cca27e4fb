#!/usr/bin/env python3
"""Host-only timing of circuit synthesis through the C++ mirror (no GPU needed): the chain circuit of BASELINE config C4
into a recycled ProvingAssignment (mode 2: create_proof's host part as the reference shapes it) and into a recycled
WitnessAssignment (mode 3: constraint matrices resident in HBM).

    python tools/host_synthesis.py [log2 rounds] [repeats] [library dir ...]

With several library directories (A/B builds: `make -C bellman_amd/csrc OUT=../lib_expA`) the measurements alternate
between them in one process, so that a noisy host hits all builds alike; minimum and median are reported."""
import ctypes, os, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
dirs = sys.argv[3:] or [os.path.join(root, "bellman_amd", "lib")]
libs = []
for d in dirs:
    lib = ctypes.CDLL(os.path.join(d, "libbellman_hip_test.so"))
    lib.bh_test_synthesis_ms.restype = ctypes.c_double
    lib.bh_test_synthesis_ms.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_int]
    libs.append(lib)
n = (1 << log_n) - 1
times = {(i, m): [] for i in range(len(libs)) for m in (2, 3)}
for r in range(reps + 1):
    for mode in (2, 3):
        for i, lib in enumerate(libs):
            t = lib.bh_test_synthesis_ms(1, n, 99, mode)
            if r:   # the first round grows the recycled vectors
                times[i, mode].append(t)
for i, d in enumerate(dirs):
    for mode, name in ((2, "ProvingAssignment"), (3, "WitnessAssignment")):
        t = sorted(times[i, mode])
        print(f"{os.path.relpath(d, root):24s} chain 2^{log_n} {name} (recycled): min {t[0]:6.1f} ms, median {t[len(t) // 2]:6.1f} ms"
              f" ({t[0] * 1e6 / n:.1f} ns per constraint)")
