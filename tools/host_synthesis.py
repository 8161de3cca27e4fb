#!/usr/bin/env python3
"""Host-only timing of circuit synthesis through the C++ mirror (no GPU needed): the chain circuit of BASELINE config C4
into a recycled ProvingAssignment (mode 2: create_proof's host part as the reference shapes it) and into a recycled
WitnessAssignment (mode 3: constraint matrices resident in HBM).  `python tools/host_synthesis.py [log2 rounds] [repeats]`"""
import ctypes, os, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(root, "bellman_amd", "lib", "libbellman_hip_test.so"))
lib.bh_test_synthesis_ms.restype = ctypes.c_double
lib.bh_test_synthesis_ms.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_int]
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
n = (1 << log_n) - 1
for mode, name in ((2, "ProvingAssignment (recycled)"), (3, "WitnessAssignment (recycled)")):
    t = sorted(lib.bh_test_synthesis_ms(1, n, 99, mode) for _ in range(reps + 1))[:-1]
    print(f"chain 2^{log_n} {name}: min {t[0]:.1f} ms, median {t[len(t) // 2]:.1f} ms ({t[0] * 1e6 / n:.1f} ns per constraint)")
