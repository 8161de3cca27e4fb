"""TEST INFRASTRUCTURE: prover.rs:217-360 restated on top of the C oracle's multiexp / FFT
(oracle/c), for sizes where the generic Python restatement (oracle/pyref/prover.py) is too slow.
Inputs are the fields of a synthesised ProvingAssignment (ints) and the CRS as numpy records."""

import numpy as np

from . import cref

Q = cref.Q


def prove_assignment(a_ev, b_ev, c_ev, input_assignment, aux_assignment, a_aux_density, b_input_density,
                     b_aux_density, vk, h, l, a, b_g1, b_g2, r, s, threads=0, concurrent=False, timing=None):
    """vk: dict of numpy records alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2.
    Returns (a, b, c) affine records, or raises RuntimeError(code).
    concurrent: issue the eight multiexps together, as prover.rs:244-318 does on the rayon pool
    (their window tasks then share all host cores).  timing (dict, optional) receives the seconds
    spent in the C restatement only (h block + multiexps + assembly), excluding the Python-side
    conversion of the inputs."""
    import time
    from concurrent.futures import ThreadPoolExecutor

    n = len(a_ev)
    m = 1
    while m < n:
        m *= 2
    pad = lambda v: cref.fr_to_mont(cref.ints_to_arr(list(v) + [0] * (m - n), 4))  # noqa: E731
    pa, pb, pc = pad(a_ev), pad(b_ev), pad(c_ev)
    ia = cref.ints_to_arr(list(input_assignment), 4)
    aa = cref.ints_to_arr(list(aux_assignment), 4) if len(aux_assignment) else np.zeros((0, 4), dtype=np.uint64)
    n_in = len(input_assignment)
    b_in_total = int(sum(b_input_density))
    dens = {id(d): cref.density_bitmap(d) for d in (a_aux_density, b_input_density, b_aux_density)}
    t0 = time.perf_counter()
    hco = cref.fr_from_mont(cref.h_coeffs(pa, pb, pc, threads=threads or 8))  # canonical

    def me(group, bases, offset, density, scalars):
        rc, pt = cref.multiexp(group, bases, offset, None if density is None else dens[id(density)], scalars,
                               threads=threads)
        return rc, pt

    # wait order of prover.rs:339-354
    calls = [
        (1, a, 0, None, ia),
        (1, a, n_in, a_aux_density, aa),
        (1, b_g1, 0, b_input_density, ia),
        (1, b_g1, b_in_total, b_aux_density, aa),
        (2, b_g2, 0, b_input_density, ia),
        (2, b_g2, b_in_total, b_aux_density, aa),
        (1, h, 0, None, hco),
        (1, l, 0, None, aa),
    ]
    if concurrent:
        with ThreadPoolExecutor(max_workers=len(calls)) as ex:
            jobs = list(ex.map(lambda c: me(*c), calls))
    else:
        jobs = [me(*c) for c in calls]
    if timing is not None:
        timing["multiexp_and_h_s"] = time.perf_counter() - t0
    if not vk["delta_g1"].any() or not vk["delta_g2"].any():
        raise RuntimeError(1)
    for rc, _ in jobs:
        if rc:
            raise RuntimeError(rc)
    (a_in, a_aux, b1_in, b1_aux, b2_in, b2_aux, h_res, l_res) = [j[1] for j in jobs]
    add, mul = cref.point_add, cref.point_mul
    g_a = add(1, mul(1, vk["delta_g1"], r), vk["alpha_g1"])
    g_b = add(2, mul(2, vk["delta_g2"], s), vk["beta_g2"])
    g_c = mul(1, vk["delta_g1"], r * s % Q)
    g_c = add(1, g_c, mul(1, vk["alpha_g1"], s))
    g_c = add(1, g_c, mul(1, vk["beta_g1"], r))
    a_answer = add(1, a_in, a_aux)
    g_a = add(1, g_a, a_answer)
    g_c = add(1, g_c, mul(1, a_answer, s))
    b1_answer = add(1, b1_in, b1_aux)
    b2_answer = add(2, b2_in, b2_aux)
    g_b = add(2, g_b, b2_answer)
    g_c = add(1, g_c, mul(1, b1_answer, r))
    g_c = add(1, g_c, h_res)
    g_c = add(1, g_c, l_res)
    if timing is not None:
        timing["total_s"] = time.perf_counter() - t0
    return g_a, g_b, g_c
