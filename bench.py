#!/usr/bin/env python3
"""Benchmark of the MI355X Groth16 hot path - driver contract in the task statement.

Workload (BASELINE.json configs[1], the config the headline MSM metric is quoted on):
  one "step" = one G1 multi-scalar multiplication over 2^20 (base, scalar) terms per GPU,
  BLS12-381, FullDensity, inputs already resident in HBM (bases registered once as the CRS
  would be; scalars in a device buffer) -> bellman's multiexp::multiexp (src/multiexp.rs:305).
  metric = G1 MSM throughput in M scalar-mul/s = terms processed by all ranks / wall time.

N > 1 (one process per GPU, torch.distributed/RCCL): the MSM shards by bases;
every rank runs the full single-GPU pipeline on its own 2^20-term shard ("weak" scaling) and
the per-rank partial results (one 96-byte affine point each) are exchanged with ONE tiny
all-gather per step and folded locally - RCCL has no user-defined reduction for EC addition.

Extra objects on the JSON line:
  roofline     dominant kernel = msm_accumulate_kernel (bucket accumulation).  achieved =
               algorithmic bytes per launch (128 B per term: 96 B affine base + 32 B scalar,
               SURVEY.md 8d) / mean launch duration measured with HIP events on the job's
               stream inside the library (bh_msm_wait_profile).  peak = 8000 GB/s HBM3E.
  cpu_baseline the C restatement of bellman's rayon path (oracle/c, one task per window,
               src/multiexp.rs:288-293) timed on this box's host cores on the same inputs.
"""

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

# before anything initialises the HIP runtime (torch does, below): the library's job streams want more than the
# runtime's default 4 hardware queues (bellman_amd/csrc/api.hip)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LOG_N = 20
HBM_PEAK_GBPS = 8000.0
BYTES_PER_TERM_G1 = 128  # 96 B affine base + 32 B scalar, each read once


# XYZZ madd: 8M + 2S; 13-limb radix-2^30 product (169 mads) + reduction (169 + 13 quotient products); symmetric squaring
# (91 + 182).  Since round 4 the last line Y3 = R*(Q - X3) - Y1*PPP is two products under ONE reduction (ff.cuh fe_mul2):
# one reduction (169 + 13) fewer per addition - the numerator of roofline.alu shrinks with it.
MADS_PER_MIXED_ADD = 8 * 351 + 2 * 273 - 182
# Algorithmic figures of the other kernels, for the roofline objects of the fft and create_proof blocks:
MADS_PER_FR_BUTTERFLY = 153          # one Fr product with a pre-sliced twiddle (ff.cuh: 81 + 72 mads), one per butterfly
MADS_PER_MIXED_ADD_G2 = 3 * MADS_PER_MIXED_ADD   # an Fp2 product is three Fp products by Karatsuba: the algorithmic count
MAD_PEAK_T = 26.2                    # v_mad_u64_u32 ceiling of the chip, T/s (profiles/r1_microbench_int.txt)
BYTES_PER_TERM_G2 = 224


def splitmix_scalars(n, seed):
    """Uniform-ish scalars < q: SplitMix64 limbs, top limb clamped below q's top limb."""
    idx = np.arange(n * 4, dtype=np.uint64) + np.uint64(seed)
    z = idx * np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    a = z.reshape(n, 4)
    a[:, 3] %= np.uint64(0x73EDA753299D7D48)
    return np.ascontiguousarray(a)


G1_GEN_MONT = np.array(  # BLS12-381 G1 generator, Montgomery limbs (x | y)
    [0x5CB38790FD530C16, 0x7817FC679976FFF5, 0x154F95C7143BA1C1, 0xF0AE6ACDF3D0E747,
     0xEDCE6ECC21DBF440, 0x120177419E0BFB75, 0xBAAC93D50CE72271, 0x8C22631A7918FD8E,
     0xDD595F13570725CE, 0x51AC582950405194, 0x0E1C8C3FAD0059C0, 0x0BBC3EFC5008A26A],
    dtype=np.uint64,
)


G2_GEN_MONT = np.array(  # BLS12-381 G2 generator, Montgomery limbs (x.c0 | x.c1 | y.c0 | y.c1)
    [0xF5F28FA202940A10, 0xB3F5FB2687B4961A, 0xA1A893B53E2AE580, 0x9894999D1A3CAEE9, 0x6F67B7631863366B, 0x058191924350BCD7,
     0xA5A9C0759E23F606, 0xAAA0C59DBCCD60C3, 0x3BB17E18E2867806, 0x1B1AB6CC8541B367, 0xC2B6ED0EF2158547, 0x11922A097360EDF3,
     0x4C730AF860494C4A, 0x597CFA1F5E369C5A, 0xE7E6856CAA0A635A, 0xBBEFB5E96E0D495F, 0x07D3A975F0EF25A2, 0x0083FD8E7E80DAE5,
     0xADC0FC92DF64B05D, 0x18AA270A2B1461DC, 0x86ADAC6A3BE4EBA0, 0x79495C4EC93DA33A, 0xE7175850A43CCAED, 0x0B2BC2A163DE1BF2],
    dtype=np.uint64,
)


def proof_roofline(log_n, n_aux, a_dense, b_dense, device_ms, g1_rows=16):
    """Work of ONE create_proof (prover.rs:217-318) against the span of its device part: the five large multiexps (h: m - 1
    terms, l: n_aux, a_aux / b_g1 / b_g2: the dense entries of their queries) as digit columns x terms mixed additions -
    g1_rows columns for the G1 queries (13 since round 6: their 20-bit window tables; 16 before), 16 for the G2 query - the 7
    FFTs as (m / 2) log m butterflies, bytes per SURVEY.md 8(d)."""
    m = 1 << log_n
    g1_terms = (m - 1) + n_aux + a_dense + b_dense
    mads = g1_rows * g1_terms * MADS_PER_MIXED_ADD + 16 * b_dense * MADS_PER_MIXED_ADD_G2 + 7 * (m // 2) * log_n * MADS_PER_FR_BUTTERFLY
    nbytes = g1_terms * BYTES_PER_TERM_G1 + b_dense * BYTES_PER_TERM_G2 + 7 * 64 * m + 128 * m
    t = device_ms * 1e-3
    return {"bound": "alu", "unit": "Tmad/s", "peak": MAD_PEAK_T, "achieved": round(mads / t / 1e12, 2), "frac": round(mads / t / 1e12 / MAD_PEAK_T, 4),
            "device_ms": round(device_ms, 2),
            "work": "%d digit columns x (h %d + l %d + a_aux %d + b_g1 %d) G1 mixed additions x %d mads + 16 x %d G2 mixed additions x %d "
                    "(three Fp products per Fp2 product) + 7 FFTs x (m/2) log m butterflies x %d; the input multiexps, the fused "
                    "quotient and the constraint evaluation not counted" % (g1_rows, m - 1, n_aux, a_dense, b_dense, MADS_PER_MIXED_ADD, b_dense,
                                                                            MADS_PER_MIXED_ADD_G2, MADS_PER_FR_BUTTERFLY),
            "hbm": {"algorithmic_bytes": nbytes, "achieved_GBps": round(nbytes / t / 1e9, 1), "frac_of_8TBps": round(nbytes / t / 8e12, 5)}}


def bench_create_proof(worker, lib, log_n, proofs=10, cpu_baseline=True):
    """BASELINE config C4: groth16::create_proof on a synthetic 2^log_n-constraint R1CS (the C++
    chain circuit of groth16_capi.cpp).  The CRS is a real one for this circuit, made by the product's
    device generator (generate_parameters, generator.rs:163-510) from fixed toxic waste - as the
    reference's own tests do (groth16/src/tests/mod.rs:93-99); not a secure setup."""
    from bellman_amd import groth16 as pg

    lib.use_unchecked_demo_circuits()   # the timed legs run the demo circuits built WITHOUT closure checks (csrc/Makefile)
    rounds = (1 << log_n) - 3
    CIRCUIT_SEED = 2020
    t0 = time.perf_counter()
    r1cs = pg.R1CS.from_demo(worker, 1, rounds, CIRCUIT_SEED)
    capture_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    params = pg.Parameters.generate(worker, r1cs, G1_GEN_MONT, G2_GEN_MONT, alpha=48577, beta=22580, gamma=53332, delta=5481,
                                    tau=3673)
    generate_ms = (time.perf_counter() - t0) * 1e3
    # the serialized form and back (Parameters::write / Parameters::read, groth16/src/lib.rs:258-398): the proofs
    # below use the parameters that came through the reader with `checked = true`
    t0 = time.perf_counter()
    blob = bytearray(params.serialized_len())   # (the writer's buffer; Parameters.write() would add a copy into immutable bytes)
    params.write_into(blob)
    write_ms = (time.perf_counter() - t0) * 1e3
    params.release()
    t0 = time.perf_counter()
    params = pg.Parameters.read(worker, blob, True)
    read_checked_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    pg.Parameters.read(worker, blob, False).release()
    read_unchecked_ms = (time.perf_counter() - t0) * 1e3
    crs_bytes = len(blob)
    del blob
    tms = []
    for i in range(proofs + 1):
        tm = [0, 0, 0, 0]
        t0 = time.perf_counter()
        last = pg.create_proof_demo(params, 1, rounds, CIRCUIT_SEED, [987654321 + i], None, 0xABCDEF0123 + i, 0x123456789AB, tm)
        wall = (time.perf_counter() - t0) * 1e3
        if i:
            tms.append(tm + [wall])
    # concurrent throughput: independent proofs from several host threads through ONE context (the
    # library is thread-safe; ctypes releases the GIL), so one proof's single-threaded synthesis
    # overlaps the others' GPU work - how a proving service would drive it
    from concurrent.futures import ThreadPoolExecutor

    def one(i):
        pg.create_proof_demo(params, 1, rounds, CIRCUIT_SEED, [1234567 + i], None, 0x55AA + i, 0x77, None)

    threads, per_thread = int(os.environ.get("BENCH_PROOF_THREADS", "12")), 2
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(one, range(threads * per_thread)))
    conc = threads * per_thread / (time.perf_counter() - t0)
    # the same proofs with the constraint matrices resident in HBM: the matrices are
    # captured ONCE per circuit (outside the timed region, like the CRS); a proof then runs only the
    # circuit's witness closures on the host and evaluates a = A.w, b = B.w, c = C.w on the device
    tms_r = []
    for i in range(proofs + 1):
        tm = [0, 0, 0, 0]
        t0 = time.perf_counter()
        last_r = pg.create_proof_demo_r1cs(params, r1cs, 1, rounds, CIRCUIT_SEED, [987654321 + i], None, 0xABCDEF0123 + i, 0x123456789AB, tm)
        wall = (time.perf_counter() - t0) * 1e3
        if i:
            tms_r.append(tm + [wall])

    def one_r(i):
        pg.create_proof_demo_r1cs(params, r1cs, 1, rounds, CIRCUIT_SEED, [1234567 + i], None, 0x55AA + i, 0x77, None)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(one_r, range(threads * per_thread)))
    conc_r = threads * per_thread / (time.perf_counter() - t0)
    assert last_r.a.tobytes() == last.a.tobytes() and last_r.b.tobytes() == last.b.tobytes() and \
        last_r.c.tobytes() == last.c.tobytes(), "device-evaluated constraints gave a different proof"

    # ONE caller thread, two proofs deep (bh_groth16_prove_demo_async / groth16::ProofPipeline): the synthesis of proof
    # k+1 runs on this thread while a helper thread drives the device part of proof k.  Every proof uses the inputs of
    # `last`, which it must equal.
    def pipelined(r1cs_or_none, count):
        waits, got = [], []
        t0 = time.perf_counter()
        for _ in range(count):
            waits.append(pg.create_proof_demo_async(params, r1cs_or_none, 1, rounds, CIRCUIT_SEED, [987654321 + proofs], None,
                                                    0xABCDEF0123 + proofs, 0x123456789AB))
            if len(waits) == 2:
                got.append(waits.pop(0)())
        while waits:
            got.append(waits.pop(0)())
        rate = count / (time.perf_counter() - t0)
        for g in got:
            assert g.a.tobytes() == last.a.tobytes() and g.b.tobytes() == last.b.tobytes() and g.c.tobytes() == last.c.tobytes(), \
                "pipelined proof differs from the synchronous one"
        return rate

    pipe_r = pipelined(r1cs, 10)
    pipe_h = pipelined(None, 6)
    # The drop-in as a patched bellman drives it (shim/patches/bellman-hip.patch; the C calls are issued by the C++
    # transcription csrc/groth16_callsites.cpp - no Rust toolchain here): the h block + eight multiexps of
    # prover.rs:217-318 on an assignment synthesised beforehand (synthesis is the same host work in every variant),
    # (a) with groth16/src/prover.rs patched, (b) with only multiexp.rs / domain.rs patched (the shipped form with a
    # device-resident EvaluationDomain, and the round-4 form with a round trip per call); (c) the mirror's own
    # bh_groth16_prove_assignment.  All must give the same proof.
    asg = pg.demo_assignment(1, rounds, CIRCUIT_SEED, [987654321 + proofs])
    call_sites = {}
    ref_tm = []
    for i in range(proofs + 1):
        tm = [0, 0, 0, 0]
        t0 = time.perf_counter()
        ref_proof = pg.prove_assignment_arrays(params, asg, 0xABCDEF0123 + proofs, 0x123456789AB, tm)
        if i:
            ref_tm.append((time.perf_counter() - t0) * 1e3)
    assert ref_proof.a.tobytes() == last.a.tobytes() and ref_proof.b.tobytes() == last.b.tobytes() and \
        ref_proof.c.tobytes() == last.c.tobytes()
    for name, patched in (("create_proof_via_patched_call_sites", True), ("create_proof_via_multiexp_and_fft_call_sites_only", "resident"),
                          ("create_proof_via_multiexp_and_fft_call_sites_only_round4_patch", False)):
        ws = []
        for i in range((proofs if patched else min(proofs, 3)) + 1):
            t0 = time.perf_counter()
            got = pg.prove_via_call_sites(params, asg, 0xABCDEF0123 + proofs, 0x123456789AB, patched)
            if i:
                ws.append((time.perf_counter() - t0) * 1e3)
        assert got.a.tobytes() == last.a.tobytes() and got.b.tobytes() == last.b.tobytes() and got.c.tobytes() == last.c.tobytes(), \
            name + ": proof differs from bh_groth16_prove_assignment"
        call_sites[name] = {"ms_after_synthesis": round(float(np.median(ws)), 2), "proofs_per_s_after_synthesis": round(1e3 / float(np.median(ws)), 3),
                            "samples": len(ws)}
    call_sites["create_proof_via_patched_call_sites"]["calls"] = (
        "groth16/src/prover.rs patched: bh_scalars_register x2 (Montgomery, shared by the multiexps that use them), "
        "bh_msm_async_scalars x8, bh_h_poly_fr_scalars x1 (h coefficients stay in HBM), bh_msm_wait x8; host tail prover.rs:320-360")
    call_sites["create_proof_via_multiexp_and_fft_call_sites_only"]["calls"] = (
        "only src/multiexp.rs + src/domain.rs (+ src/hip.rs) patched, prover.rs untouched: an EvaluationDomain keeps its vector in HBM "
        "between its calls (3 uploads, 7 x bh_fft_fr_dev, bh_fr_mul_assign_dev / sub_assign / divide_by_z_on_coset on the device, 1 "
        "download at into_coeffs); Exponent::from(&Scalar) defers the conversion (Montgomery words handed to the device); every "
        "Arc<Vec<Exponent>> gathered on the worker's threads and registered once (bh_scalars_register x3), bh_msm_async_scalars x8")
    call_sites["create_proof_via_multiexp_and_fft_call_sites_only_round4_patch"]["calls"] = (
        "the round-4 form of that patch level: 7 x bh_fft_fr on host vectors (upload + download each), mul/sub/divide_by_z "
        "on the host (one chunk per host thread), serial Fr -> Exponent passes (prover.rs:241-261), 8 x bh_msm_async with "
        "canonical host scalars (each uploads its vector again)")
    call_sites["bh_groth16_prove_assignment_same_inputs"] = {"ms_after_synthesis": round(float(np.median(ref_tm)), 2)}
    call_sites["note"] = ("all four proofs asserted bit-identical; host synthesis (ms_host_synthesis above) precedes each of them "
                          "in a real create_proof")
    cpu = None
    if cpu_baseline:
        h, l, a, b1, b2 = (params.query(q) for q in ("h", "l", "a", "b_g1", "b_g2"))
        vkr = params.vk()
        # CPU baseline for this metric: the C restatement of prover.rs:217-360 (oracle/cprover.py: bellman's
        # parallel_fft split, its window rule, the eight multiexps issued together so that their window
        # tasks share all host cores), one proof, same inputs as the last GPU proof - which it must equal.
        from oracle import cprover, cref
        from tests import circuits

        i = proofs
        f = circuits.chain_assignment_fast(rounds, CIRCUIT_SEED, 987654321 + i)
        vk = dict(alpha_g1=vkr[0], beta_g1=vkr[1], beta_g2=vkr[2], delta_g1=vkr[3], delta_g2=vkr[4])
        cpu_threads = cref.lib().orc_max_threads()
        tcpu = {}
        want = cprover.prove_assignment(f["a"], f["b"], f["c"], f["input_assignment"], f["aux_assignment"], f["a_aux_density"],
                                        f["b_input_density"], f["b_aux_density"], vk, h, l, a, b1, b2, 0xABCDEF0123 + i,
                                        0x123456789AB, threads=cpu_threads, concurrent=True, timing=tcpu)
        assert last.a.tobytes() == want[0].tobytes() and last.b.tobytes() == want[1].tobytes() and \
            last.c.tobytes() == want[2].tobytes(), "GPU proof differs from the CPU oracle's"
        cpu = {
            "value": round(1.0 / tcpu["total_s"], 4), "unit": "proofs/s", "cores": cpu_threads, "kind": "port",
            "sample": "1 proof, same 2^%d-constraint instance (proof bit-identical to the GPU's): C restatement of "
                      "prover.rs:217-360 on evaluations synthesised beforehand, the 8 multiexps issued together; host "
                      "synthesis excluded (it would add to the CPU side only)" % log_n,
            "seconds": round(tcpu["total_s"], 3),
        }
    g1_rows = params.bases("h").table_info()[1] or 16   # digit columns of the G1 multiexps: the rows of their window tables
    r1cs.release()
    params.release()
    m = np.median(np.array(tms), axis=0)
    mr = np.median(np.array(tms_r), axis=0)
    pop = lambda words: int(np.unpackbits(words.view(np.uint8)).sum())  # noqa: E731
    n_aux = asg["aux_assignment"].shape[0]
    roof = proof_roofline(log_n, n_aux, pop(asg["a_aux_density"]), pop(asg["b_aux_density"]), float(mr[4]) - float(mr[0]), g1_rows)
    return {
        "workload": "groth16::create_proof, synthetic multiplicative-chain R1CS, 2^%d constraints, 1 public input "
                    "(BASELINE.json configs[3]): 7 FFTs + fused quotient, 4 large G1 + 1 large G2 multiexp (+3 small)" % log_n,
        "proofs_per_s": round(1e3 / float(m[4]), 3),
        "ms_total": round(float(m[4]), 2),
        "ms_host_synthesis": round(float(m[0]), 2),
        "ms_issue_7_multiexps_then_h_block_incl_uploads": round(float(m[1]), 2),
        "ms_h_multiexp_and_waits": round(float(m[2]), 2),
        "proofs_per_s_excluding_host_synthesis": round(1e3 / (float(m[4]) - float(m[0])), 3),
        "proofs_per_s_one_caller_pipelined": round(pipe_h, 3),
        "proofs_per_s_concurrent": round(conc, 3),
        "concurrent_host_threads": threads,
        "samples": proofs,
        "statistic": "median of %d proofs after one warm-up (round 5: mean of 3); demo circuits from libbellman_hip_demo.so, "
                     "built without BELLMAN_HIP_CHECK_CLOSURES" % proofs,
        "roofline": roof,
        "cpu_baseline": cpu,
        "drop_in_call_sites": call_sites,
        "crs": "generate_parameters on the device from fixed toxic waste: %.0f ms (h, l, a, b_g1, b_g2 = %d G1 + %d G2 "
               "fixed-base multiplications, 1 iFFT, transposed sparse product); Parameters::write %.0f ms (%.0f MB, host "
               "encoding); Parameters::read(checked) %.0f ms / (unchecked) %.0f ms incl. the host-to-device copy - decoding, "
               "on-curve and subgroup tests on the device; untimed set-up"
               % (generate_ms, 4 * (1 << log_n), 1 << log_n, write_ms, crs_bytes / 1e6, read_checked_ms, read_unchecked_ms),
        # what a user waits for before the first proof (none of it is in any per-proof figure); structured so that
        # regressions there are visible (groth16/src/lib.rs:258-398, generator.rs:163-510)
        "setup": {"constraints_log2": log_n, "r1cs_capture_ms": round(capture_ms, 1), "generate_parameters_ms": round(generate_ms, 1),
                  "params_write_ms": round(write_ms, 1), "params_bytes": crs_bytes, "params_read_checked_ms": round(read_checked_ms, 1),
                  "params_read_unchecked_ms": round(read_unchecked_ms, 1), "window_table_bytes": int(worker.info().get("table_bytes", 0))},
        "with_r1cs_resident_in_hbm": {
            "note": "constraint matrices captured once per circuit (%.0f ms, untimed, like the CRS upload); per proof: "
                    "witness closures on the host, A.w/B.w/C.w + everything else on the device; identical proofs" % capture_ms,
            "proofs_per_s": round(1e3 / float(mr[4]), 3),
            "ms_total": round(float(mr[4]), 2),
            "ms_host_witness": round(float(mr[0]), 2),
            "ms_issue_7_multiexps_then_h_block_incl_uploads": round(float(mr[1]), 2),
            "ms_h_multiexp_and_waits": round(float(mr[2]), 2),
            "proofs_per_s_one_caller_pipelined": round(pipe_r, 3),
            "pipeline_note": "one caller thread, two proofs deep (bh_groth16_prove_demo_async): witness closures of proof k+1 on "
                             "the caller thread while a helper thread runs the device part of proof k; proofs asserted identical",
            "proofs_per_s_concurrent": round(conc_r, 3),
        },
    }


def bench_create_proof_c5(worker, log_n=24):
    """BASELINE config C5's proof leg on ONE GPU: create_proof at 2^24 constraints (R1CS resident, CRS from the device
    generator).  One sample after one warm-up; the parts == single and pairing checks live in tests/test_gpu_scale.py."""
    from bellman_amd import groth16 as pg

    rounds = (1 << log_n) - 3
    seed = 2024
    t0 = time.perf_counter()
    r1cs = pg.R1CS.from_demo(worker, 1, rounds, seed)
    capture_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    params = pg.Parameters.generate(worker, r1cs, G1_GEN_MONT, G2_GEN_MONT, alpha=48577, beta=22580, gamma=53332, delta=5481, tau=3673)
    generate_s = time.perf_counter() - t0
    out = []
    for i in range(2):
        tm = [0, 0, 0, 0]
        t0 = time.perf_counter()
        proof = pg.create_proof_demo_r1cs(params, r1cs, 1, rounds, seed, [55555 + i], None, 0xC5C5 + i, 0x5C5C, tm)
        out.append((tm, (time.perf_counter() - t0) * 1e3))
    tm, wall = out[1]
    # sliced over 2 "ranks" on this one GPU: the fold must be the same proof (what N = 2 would assemble)
    total = None
    for part in range(2):
        sums = pg.prove_demo_part(params, r1cs, 1, rounds, seed, [55555 + 1], None, part, 2)
        total = sums if total is None else pg.sums_add(total, sums)
    folded = pg.assemble(params, total, 0xC5C5 + 1, 0x5C5C)
    assert folded.a.tobytes() == proof.a.tobytes() and folded.b.tobytes() == proof.b.tobytes() and folded.c.tobytes() == proof.c.tobytes(), \
        "2-part proof differs from the single-GPU proof"
    r1cs.release()
    params.release()
    worker.trim()
    return {"workload": "groth16::create_proof, 2^%d constraints (BASELINE.json configs[4], proof leg), one GPU, R1CS resident; "
                        "h query 2^%d G1 points, b_g2 query 2^%d G2 points (classic 16-window plan: no window table above 2^22)"
                        % (log_n, log_n, log_n - 1),
            "ms_total": round(wall, 1), "ms_host_witness": round(float(tm[0]), 1), "ms_gpu_part": round(wall - float(tm[0]), 1),
            "proofs_per_s": round(1e3 / wall, 3), "samples": 1,
            "check": "fold of the 2-part proof (prove_witness_part x2 + sums_add + assemble) == this proof",
            "setup_s": {"r1cs_capture": round(capture_s, 1), "generate_parameters": round(generate_s, 1)}}


def bench_create_proof_sharded(worker, log_n, world, rank, coll_dev, proofs=3):
    """N > 1: ONE proof at a time over all ranks (BASELINE config C5's proof leg): every
    rank holds the CRS and the constraint matrices, builds the witness and runs the h block; each of the
    eight multiexps is computed over the rank's slice of the scalar indices; one all-gather of 960 bytes
    (RCCL), slot-wise fold, identical proof on every rank.  Strong scaling of a fixed-size proof."""
    import torch
    import torch.distributed as dist

    from bellman_amd import groth16 as pg
    from bellman_amd import sharding

    rounds = (1 << log_n) - 3
    seed = 2020
    r1cs = pg.R1CS.from_demo(worker, 1, rounds, seed)
    params = pg.Parameters.generate(worker, r1cs, G1_GEN_MONT, G2_GEN_MONT, alpha=48577, beta=22580, gamma=53332, delta=5481, tau=3673)
    dev = "cuda" if coll_dev == "cuda" else None

    def one(i):
        return sharding.create_proof_sharded(
            lambda rk, wd: pg.prove_demo_part(params, r1cs, 1, rounds, seed, [987654321 + i], None, rk, wd),
            params, 0xABCDEF0123 + i, 0x123456789AB, device=dev)

    one(0)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(1, proofs + 1):
        last = one(i)
    dist.barrier()
    torch.cuda.synchronize()
    tt = torch.tensor([time.perf_counter() - t0], device=coll_dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    # every rank must hold the same proof, and it must be the single-GPU one
    whole = pg.create_proof_demo_r1cs(params, r1cs, 1, rounds, seed, [987654321 + proofs], None, 0xABCDEF0123 + proofs, 0x123456789AB)
    same = last.a.tobytes() == whole.a.tobytes() and last.b.tobytes() == whole.b.tobytes() and last.c.tobytes() == whole.c.tobytes()
    flag = torch.tensor([1 if same else 0], device=coll_dev, dtype=torch.int64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    assert int(flag.item()) == 1, "sharded proof differs from the single-GPU proof"
    r1cs.release()
    params.release()
    return {
        "workload": "one groth16 proof at a time over %d ranks, 2^%d constraints (multiexps sliced by scalar index, witness + "
                    "h block replicated, one 960-byte all-gather per proof); identical to the single-GPU proof" % (world, log_n),
        "proofs_per_s": round(proofs / elapsed, 3),
        "ms_per_proof": round(elapsed * 1e3 / proofs, 2),
        "scaling": "strong",
        "samples": proofs,
    }


def bench_fft(worker, lib, log_n=22, iters=10):
    """BASELINE config C3: 2^22-point radix-2 FFT / iFFT / coset variants, vector resident in HBM."""
    n = 1 << log_n
    data = splitmix_scalars(n, 3)
    d = worker.alloc(n * 32)
    worker.upload(d, data)
    out = {"workload": "EvaluationDomain fft/ifft/coset_fft/icoset_fft, 2^%d Fr elements resident in HBM "
                       "(BASELINE.json configs[2])" % log_n, "algorithmic_bytes_per_element": 64}
    # The clock is still ramping up when the first transforms of an idle device run: the same pass executes the same wave
    # cycles in 287 us as the process' first mode and in 262 us as its fourth (profiles/archive/r5_call2_fft_wave_local.txt) -
    # 40 untimed transforms (each table built once, ~25 ms of load) come first, so that all four modes are timed alike.
    for i in range(40):
        assert lib.bh_fft_fr_dev(worker.ctx, d, log_n, i & 3, None) == 0
    worker.synchronize()
    for mode, name in [(0, "fft"), (1, "ifft"), (2, "coset_fft"), (3, "icoset_fft")]:
        for _ in range(2):
            assert lib.bh_fft_fr_dev(worker.ctx, d, log_n, mode, None) == 0
        worker.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            lib.bh_fft_fr_dev(worker.ctx, d, log_n, mode, None)
        worker.synchronize()
        dt = (time.perf_counter() - t0) / iters
        out[name] = {"ms": round(dt * 1e3, 4), "algorithmic_GBps": round(64.0 * n / dt / 1e9, 1),
                     "frac_of_8TBps": round(64.0 * n / dt / 8e12, 4), "Gbutterflies_per_s": round(n / 2 * log_n / dt / 1e9, 1),
                     "Tmad_per_s": round(n / 2 * log_n * MADS_PER_FR_BUTTERFLY / dt / 1e12, 2),
                     "frac_of_mad_ceiling": round(n / 2 * log_n * MADS_PER_FR_BUTTERFLY / dt / 1e12 / MAD_PEAK_T, 4)}
    worker.free(d)
    worst = max(out[k]["ms"] for k in ("fft", "ifft", "coset_fft", "icoset_fft"))
    out["roofline"] = {
        "kernel": "ntt_pass_kernel (two launches per transform at 2^22)", "statistic": "the slowest of the four modes",
        "hbm": {"bound": "hbm", "achieved": round(64.0 * n / (worst * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(64.0 * n / (worst * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "bytes": "64 B per element: one read + one write (SURVEY.md 8d)"},
        "alu": {"bound": "alu", "achieved": round(n / 2 * log_n * MADS_PER_FR_BUTTERFLY / (worst * 1e-3) / 1e12, 2), "peak": MAD_PEAK_T, "unit": "Tmad/s",
                "frac": round(n / 2 * log_n * MADS_PER_FR_BUTTERFLY / (worst * 1e-3) / 1e12 / MAD_PEAK_T, 4),
                "work": "(n / 2) log2 n butterflies x one Fr product of %d v_mad_u64_u32 (pre-sliced twiddle); the inter-pass twiddle, "
                        "coset and 1/n products the passes also execute are not counted: the algorithmic minimum" % MADS_PER_FR_BUTTERFLY}}
    return out


def bench_msm_shape(worker, lib, group, log_n, iters=10):
    """One more multiexp shape (device-resident inputs, bases registered once): G2 at the size it has inside a
    2^20-constraint proof (2^19) and at 2^20; G1 at 2^16, the shape of the reference's own bench
    (benches/slow.rs:14-44).  Median wall time of `iters` calls after 2 warm-ups."""
    import bellman_amd

    n = 1 << log_n
    words = 12 if group == 1 else 24
    gen = G1_GEN_MONT if group == 1 else G2_GEN_MONT
    t = splitmix_scalars(n, 0xB45E5 + group)
    dt, dout = worker.alloc(n * 32), worker.alloc(n * 8 * words)
    worker.upload(dt, t)
    assert lib.bh_fixed_base_mul_dev(worker.ctx, group, gen.ctypes.data_as(ctypes.c_void_p), dt, n, 0, dout, None) == 0
    worker.synchronize()
    bases = bellman_amd.Bases.copy_device(worker, group, dout, n)   # registered like a CRS query (automatic window table)
    sc = splitmix_scalars(n, 0x5CA1A + group)
    worker.upload(dt, sc)
    walls, stages = [], []
    for it in range(iters + 2):
        t0 = time.perf_counter()
        _, ms = bellman_amd.multiexp(worker, bases, bellman_amd.FullDensity(), None, scalars_dev=dt, n=n, timed=True).wait()
        if it >= 2:
            walls.append((time.perf_counter() - t0) * 1e3)
            stages.append(ms)
    bases.release()
    worker.free(dt)
    worker.free(dout)
    med = float(np.median(walls))
    st = np.median(np.array(stages), axis=0)
    return {"group": "G%d" % group, "log_n": log_n, "ms_median": round(med, 4), "Mscalar_mul_per_s": round(n / med / 1e3, 3),
            "device_ms": {"pipeline": round(float(st[0]), 4), "digits_sort": round(float(st[1]), 4),
                          "bucket_accumulate": round(float(st[2]), 4), "merge_reduce": round(float(st[3]), 4)},
            "samples": iters}


def bench_mimc(worker, proofs=30, cpu_baseline=True):
    """BASELINE config C1: groth16::create_proof on MiMC-322 (groth16/tests/mimc.rs: 646 constraints, m = 2^10),
    through the C++ mirror of the circuit; CPU baseline = the C restatement of prover.rs:217-360 on the same
    assignment (all host threads), whose proof the GPU proof must equal."""
    import random

    from bellman_amd import groth16 as pg
    from oracle.pyref import bls12_381 as bls
    from tests import circuits

    rnd = random.Random(322)
    cons = [rnd.randrange(bls.Q) for _ in range(circuits.MIMC_ROUNDS)]
    xl, xr = rnd.randrange(bls.Q), rnd.randrange(bls.Q)
    r, s = rnd.randrange(bls.Q), rnd.randrange(bls.Q)
    r1cs = pg.R1CS.from_demo(worker, 0, circuits.MIMC_ROUNDS, 0, cons)
    cons_mont = pg.fr_to_mont_array(cons)   # the round constants are fixed: converted once, not per proof
    params = pg.Parameters.generate(worker, r1cs, G1_GEN_MONT, G2_GEN_MONT, alpha=48577, beta=22580, gamma=53332, delta=5481, tau=3673)
    walls = []
    for i in range(proofs + 3):
        t0 = time.perf_counter()
        last = pg.create_proof_demo(params, 0, circuits.MIMC_ROUNDS, 0, [xl, xr], cons_mont, r, s)
        if i >= 3:
            walls.append((time.perf_counter() - t0) * 1e3)
    med = float(np.median(walls))
    out = {"workload": "groth16::create_proof, MiMC-322 (646 constraints, domain 2^10; BASELINE.json configs[0] on the GPU)",
           "ms_median": round(med, 4), "proofs_per_s": round(1e3 / med, 2), "samples": proofs}
    if cpu_baseline:
        from oracle import cprover, cref
        from oracle.pyref.core import INPUT, Variable
        from oracle.pyref.prover import ProvingAssignment

        pa = ProvingAssignment(bls.Q)
        pa.alloc_input(lambda: 1)
        circuits.mimc_circuit(xl, xr, cons)(pa)
        for i in range(len(pa.input_assignment)):
            pa.enforce(lambda lc, i=i: lc + Variable(INPUT, i), lambda lc: lc, lambda lc: lc)
        h, l, a, b1, b2 = (params.query(q) for q in ("h", "l", "a", "b_g1", "b_g2"))
        vkr = params.vk()
        vk = dict(alpha_g1=vkr[0], beta_g1=vkr[1], beta_g2=vkr[2], delta_g1=vkr[3], delta_g2=vkr[4])
        threads = cref.lib().orc_max_threads()
        best = None
        for _ in range(3):
            tc = {}
            want = cprover.prove_assignment(pa.a, pa.b, pa.c, pa.input_assignment, pa.aux_assignment, list(pa.a_aux_density.bv),
                                            list(pa.b_input_density.bv), list(pa.b_aux_density.bv), vk, h, l, a, b1, b2, r, s,
                                            threads=threads, concurrent=True, timing=tc)
            best = tc["total_s"] if best is None else min(best, tc["total_s"])
        assert last.a.tobytes() == want[0].tobytes() and last.b.tobytes() == want[1].tobytes() and \
            last.c.tobytes() == want[2].tobytes(), "GPU MiMC proof differs from the CPU oracle's"
        out["cpu_baseline"] = {"value": round(1.0 / best, 2), "unit": "proofs/s", "cores": threads, "kind": "port",
                               "sample": "best of 3 runs of the C restatement of prover.rs:217-360 on the same MiMC-322 assignment "
                                         "(synthesis excluded); proof bit-identical to the GPU's", "seconds": round(best, 5)}
    r1cs.release()
    params.release()
    return out


def check_sharded_fold(worker, lib, world, rank, coll_dev, log_n_check=12):
    """N > 1: the fold of the per-rank partial results of a base-sharded multiexp equals ONE multiexp over the
    concatenation of all shards (computed on every rank from the same seeds), at a size where that is cheap.
    The shards are made exactly like the timed ones (seed offset per rank)."""
    import bellman_amd
    from bellman_amd import sharding

    n = 1 << log_n_check

    def shard(rk):
        return splitmix_scalars(n, 0x62656C6C6D616E + rk * 4 * n), splitmix_scalars(n, 0x5CA1A25 + rk * 4 * n)

    def msm(t, sc):
        m = t.shape[0]
        dt, dout, ds = worker.alloc(m * 32), worker.alloc(m * 96), worker.alloc(m * 32)
        worker.upload(dt, t)
        worker.upload(ds, sc)
        assert lib.bh_fixed_base_mul_dev(worker.ctx, 1, G1_GEN_MONT.ctypes.data_as(ctypes.c_void_p), dt, m, 0, dout, None) == 0
        worker.synchronize()
        b = bellman_amd.Bases.wrap_device(worker, 1, dout, m)
        out = bellman_amd.multiexp(worker, b, bellman_amd.FullDensity(), None, scalars_dev=ds, n=m).wait()
        b.release()
        for d in (dt, dout, ds):
            worker.free(d)
        return out

    t, sc = shard(rank)
    folded = sharding.fold_partials(msm(t, sc), 1, device=coll_dev if coll_dev == "cuda" else None)
    parts = [shard(rk) for rk in range(world)]
    whole = msm(np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]))
    assert np.array_equal(folded, whole), "fold of the sharded multiexp differs from the multiexp of the concatenated shards"
    return "fold of %d shards of 2^%d terms == one multiexp of 2^%d x %d terms (checked on every rank)" % (world, log_n_check, log_n_check, world)


def bench_one_process_sharded(lib, n_ctx, n_devices, log_n):
    """ONE multiexp over n_ctx contexts of THIS process, context k on device k % n_devices (bh_msm_sharded_async): host
    scalars in, one result out; checked against the fold of per-shard multiexps."""
    import bellman_amd

    n = 1 << log_n
    workers = [bellman_amd.Worker(k % n_devices) for k in range(n_ctx)]
    per = n // n_ctx
    shards, t_all = [], splitmix_scalars(n, 0x0E9C)
    for k, w in enumerate(workers):
        lo, hi = k * per, (n if k == n_ctx - 1 else (k + 1) * per)
        dt, dout = w.alloc((hi - lo) * 32), w.alloc((hi - lo) * 96)
        w.upload(dt, t_all[lo:hi])
        assert lib.bh_fixed_base_mul_dev(w.ctx, 1, G1_GEN_MONT.ctypes.data_as(ctypes.c_void_p), dt, hi - lo, 0, dout, None) == 0
        w.synchronize()
        w.free(dt)
        shards.append(bellman_amd.Bases.copy_device(w, 1, dout, hi - lo))
        w.free(dout)
    sc = splitmix_scalars(n, 0x5CA1A)
    got = bellman_amd.multiexp_sharded(workers, shards, bellman_amd.FullDensity(), sc).wait()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        again = bellman_amd.multiexp_sharded(workers, shards, bellman_amd.FullDensity(), sc).wait()
    dt_s = (time.perf_counter() - t0) / reps
    assert np.array_equal(got, again)
    # the same sum from one multiexp per shard, folded with bh_point_add
    acc = np.zeros(12, dtype=np.uint64)
    for k, w in enumerate(workers):
        lo, hi = k * per, (n if k == n_ctx - 1 else (k + 1) * per)
        part = bellman_amd.multiexp(w, shards[k], bellman_amd.FullDensity(), sc[lo:hi]).wait()
        r = np.zeros(12, dtype=np.uint64)
        lib.bh_point_add(1, r.ctypes.data_as(ctypes.c_void_p), acc.ctypes.data_as(ctypes.c_void_p), part.ctypes.data_as(ctypes.c_void_p), 1)
        acc = r
    assert np.array_equal(got, acc), "one-process sharded multiexp != fold of the per-shard multiexps"
    for h in shards:
        h.release()
    for w in workers:
        w.close()
    return {"workload": "ONE G1 multiexp of 2^%d terms (host scalars) over %d contexts of one process on %d device(s), "
                        "== fold of the per-shard multiexps" % (log_n, n_ctx, min(n_ctx, n_devices)),
            "value": round(n / dt_s / 1e6, 3), "unit": "Mscalar-mul/s", "ms": round(dt_s * 1e3, 2),
            "contexts": n_ctx, "devices": min(n_ctx, n_devices), "scaling": "strong"}


def bench_msm_boolean(worker, lib, bases, t_host, n, uniform_ms, iters=10, check=True):
    """The C2 multiexp on the scalar vectors REAL witnesses produce ("boolean-heavy vectors (~50 % zeros/ones)";
    the reason Exponent::Zero / One exist, src/multiexp.rs:172-182,245-252): same registered bases P_i = [t_i]G, scalars
    resident in HBM, median wall of `iters` calls per mix (tests/scalar_mixes.py), device stages, and what the job executed.
    Outside the timed calls every result is checked against [sum_i s_i t_i]G (the dot product by the oracle's field
    arithmetic); bit-exactness against the restated multiexp is tests/test_gpu_boolean.py."""
    import bellman_amd
    from tests import scalar_mixes

    ds = worker.alloc(n * 32)
    out = {"workload": "G1 multiexp, 2^%d terms, the bases of the headline workload, scalar mixes of tests/scalar_mixes.py "
                       "(bool50: 25 %% zeros + 25 %% ones; bool90: 45 %% + 45 %%; ones; small90: 90 %% below 2^8)" % int(np.log2(n)),
           "uniform_ms_per_step": round(uniform_ms, 4), "mixes": {}}
    for mix in ("bool50", "bool90", "ones", "small90"):
        sc = scalar_mixes.scalars(mix, n, 0xB001EA)
        worker.upload(ds, sc)
        walls, stages, st = [], [], None
        for it in range(iters + 2):
            t0 = time.perf_counter()
            got, ms, st = bellman_amd.multiexp(worker, bases, bellman_amd.FullDensity(), None, scalars_dev=ds, n=n, stats=True).wait()
            if it >= 2:
                walls.append((time.perf_counter() - t0) * 1e3)
                stages.append(ms)
        if check:
            from oracle import cref

            k = np.array(cref.int_to_limbs(cref.fr_dot(sc, t_host), 4), dtype=np.uint64)
            want = np.zeros(12, dtype=np.uint64)
            lib.bh_point_mul(1, want.ctypes.data_as(ctypes.c_void_p), G1_GEN_MONT.ctypes.data_as(ctypes.c_void_p), k.ctypes.data_as(ctypes.c_void_p))
            assert np.array_equal(got, want), "boolean-heavy multiexp (%s) != [sum s_i t_i]G" % mix
        med = float(np.median(walls))
        sg = np.median(np.array(stages), axis=0)
        live = st["sorted_entries"] - st["zero_digits"]
        out["mixes"][mix] = {
            "ms_median": round(med, 4), "Mscalar_mul_per_s": round(n / med / 1e3, 2), "x_uniform_rate": round(uniform_ms / med, 3),
            "device_ms": {"pipeline": round(float(sg[0]), 4), "digits_sort": round(float(sg[1]), 4),
                          "bucket_accumulate": round(float(sg[2]), 4), "merge_reduce": round(float(sg[3]), 4)},
            "merge_reduce_share_of_step": round(float(sg[3]) / med, 3),
            "live_entries": live, "mixed_additions": st["mixed_additions"],
            "accumulate_Tmad_per_s": round(st["mixed_additions"] * MADS_PER_MIXED_ADD / (float(sg[2]) * 1e-3) / 1e12, 2) if sg[2] > 0 else None}
    worker.free(ds)
    return out


def bench_create_proof_boolean(worker, lib, log_n=20, proofs=10):
    """create_proof on the boolean-heavy demo circuit (csrc/demo_circuits.cpp BoolMixCircuit: the AND / XOR / packing
    constraints of src/gadgets/boolean.rs; > 98 % of the aux assignment is 0 or 1) filling a 2^log_n domain: CRS from the
    device generator, host synthesis and R1CS-resident, medians.  The R1CS-resident proof must equal the host-synthesis one;
    equality with the restated prover is tests/test_gpu_boolean.py::test_boolean_circuit_proof_2_20_matches_oracle."""
    from bellman_amd import groth16 as pg
    from tests import circuits

    lib.use_unchecked_demo_circuits()
    rounds = circuits.boolmix_rounds(log_n)
    seed = 777
    r1cs = pg.R1CS.from_demo(worker, 5, rounds, seed)
    params = pg.Parameters.generate(worker, r1cs, G1_GEN_MONT, G2_GEN_MONT, alpha=48577, beta=22580, gamma=53332, delta=5481, tau=3673)
    tms, tms_r = [], []
    for i in range(proofs + 1):
        tm = [0, 0, 0, 0]
        t0 = time.perf_counter()
        last = pg.create_proof_demo(params, 5, rounds, seed, [0x0123456789ABCDEF + i], None, 0xB001 + i, 0xEA, tm)
        if i:
            tms.append(tm + [(time.perf_counter() - t0) * 1e3])
    for i in range(proofs + 1):
        tm = [0, 0, 0, 0]
        t0 = time.perf_counter()
        last_r = pg.create_proof_demo_r1cs(params, r1cs, 5, rounds, seed, [0x0123456789ABCDEF + i], None, 0xB001 + i, 0xEA, tm)
        if i:
            tms_r.append(tm + [(time.perf_counter() - t0) * 1e3])
    assert last_r.a.tobytes() == last.a.tobytes() and last_r.b.tobytes() == last.b.tobytes() and last_r.c.tobytes() == last.c.tobytes(), \
        "boolean circuit: device-evaluated constraints gave a different proof"
    asg = pg.demo_assignment(5, rounds, seed, [0x0123456789ABCDEF + proofs])
    aux = asg["aux_assignment"]
    one = pg.fr_to_mont_array([1])[0]
    zeros = int((aux == 0).all(axis=1).sum())
    ones = int((aux == one).all(axis=1).sum())
    pop = lambda words: int(np.unpackbits(words.view(np.uint8)).sum())  # noqa: E731
    m, mr = np.median(np.array(tms), axis=0), np.median(np.array(tms_r), axis=0)
    r1cs.release()
    params.release()
    worker.trim()
    return {"workload": "groth16::create_proof, boolean-heavy bit-mixing circuit (src/gadgets/boolean.rs shapes), %d constraints "
                        "in a 2^%d domain, 1 public input" % (asg["a"].shape[0], log_n),
            "aux_assignment": {"variables": int(aux.shape[0]), "zeros": zeros, "ones": ones,
                               "boolean_fraction": round((zeros + ones) / aux.shape[0], 4),
                               "a_query_dense": pop(asg["a_aux_density"]), "b_query_dense": pop(asg["b_aux_density"])},
            "proofs_per_s": round(1e3 / float(m[4]), 3), "ms_total": round(float(m[4]), 2), "ms_host_synthesis": round(float(m[0]), 2),
            "with_r1cs_resident_in_hbm": {"proofs_per_s": round(1e3 / float(mr[4]), 3), "ms_total": round(float(mr[4]), 2),
                                          "ms_host_witness": round(float(mr[0]), 2), "ms_device_part": round(float(mr[4]) - float(mr[0]), 2)},
            "samples": proofs, "statistic": "median"}


def bench_scaling_model(worker, lib, log_n_total=26, proof_log_n=24, reps=3):
    """What ONE rank of an N-rank run does, measured here on one GPU, so that the first real SCALE record judges itself:
      (i) MSM leg of BASELINE configs[4]: the 2^26-term G1 multiexp sharded by bases - a rank runs the whole single-GPU
          pipeline on 2^26 / N terms, then one all-gather of N x 96 bytes and N - 1 point additions (assumed 0.1 ms: RCCL's
          small-message latency, never measured here at N > 1).  T(N) = t_msm(2^26 / N) + 0.1 ms.
      (ii) proof leg: every rank builds the whole witness and runs the h block (replicated), and computes 1 / N of every
          multiexp (prove_witness_part).  T(N) = t_part(N) measured as part 0 of N + 0.1 ms: Amdahl-bound by the replicated
          host witness - stated, not hidden."""
    import bellman_amd
    from bellman_amd import groth16 as pg

    out = {"assumed_all_gather_and_fold_ms": 0.1, "msm_2p%d_strong" % log_n_total: {}, "note":
           "predictions for N ranks from single-GPU measurements of one rank's share; N > 1 has never run on hardware here"}
    nmax = 1 << log_n_total
    t = splitmix_scalars(nmax, 0xC5)
    dt, dout = worker.alloc(nmax * 32), worker.alloc(nmax * 96)
    worker.upload(dt, t)
    assert lib.bh_fixed_base_mul_dev(worker.ctx, 1, G1_GEN_MONT.ctypes.data_as(ctypes.c_void_p), dt, nmax, 0, dout, None) == 0
    worker.synchronize()
    worker.upload(dt, splitmix_scalars(nmax, 0x5CA1A25))
    base_ms = None
    for ranks in (1, 2, 4, 8):
        n = nmax // ranks
        # registered like a rank registers its shard of the CRS query: up to 2^24 points with the 20-bit window table
        bases = bellman_amd.Bases.copy_device(worker, 1, dout, n)
        worker.synchronize()
        walls = []
        for it in range(reps + 1):
            t0 = time.perf_counter()
            bellman_amd.multiexp(worker, bases, bellman_amd.FullDensity(), None, scalars_dev=dt, n=n).wait()
            if it:
                walls.append((time.perf_counter() - t0) * 1e3)
        bases_rows = bases.table_info()[1]
        bases.release()
        ms = float(np.median(walls)) + (0.1 if ranks > 1 else 0.0)
        base_ms = ms if ranks == 1 else base_ms
        out["msm_2p%d_strong" % log_n_total][str(ranks)] = {"terms_per_rank": n, "predicted_ms": round(ms, 2),
                                                               "predicted_speedup": round(base_ms / ms, 2),
                                                               "predicted_Mscalar_mul_per_s": round(nmax / ms / 1e3, 1),
                                                               "window_table_rows": int(bases_rows)}
    worker.free(dt)
    worker.free(dout)
    worker.trim()
    if proof_log_n:
        rounds = (1 << proof_log_n) - 3
        seed = 2024
        r1cs = pg.R1CS.from_demo(worker, 1, rounds, seed)
        params = pg.Parameters.generate(worker, r1cs, G1_GEN_MONT, G2_GEN_MONT, alpha=48577, beta=22580, gamma=53332, delta=5481, tau=3673)
        leg = {}
        base = None
        for ranks in (1, 2, 4, 8):
            best = None
            for it in range(2):
                tm = [0, 0, 0, 0]
                t0 = time.perf_counter()
                pg.prove_demo_part(params, r1cs, 1, rounds, seed, [55555], None, 0, ranks, tm)
                wall = (time.perf_counter() - t0) * 1e3
                if it and (best is None or wall < best[0]):
                    best = (wall, tm)
            ms = best[0] + (0.1 if ranks > 1 else 0.0)
            base = ms if ranks == 1 else base
            leg[str(ranks)] = {"predicted_ms": round(ms, 1), "predicted_speedup": round(base / ms, 2),
                               "replicated_host_witness_ms": round(float(best[1][0]), 1)}
        out["proof_2p%d_strong" % proof_log_n] = leg
        r1cs.release()
        params.release()
        worker.trim()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-proof", action="store_true", help="skip the create_proof (config C4) measurement")
    ap.add_argument("--proof-log-n", type=int, default=20)
    ap.add_argument("--c5-proof-log-n", type=int, default=24, help="N = 1: size of the C5 proof leg (0 = skip)")
    ap.add_argument("--c5-log-n", type=int, default=26,
                    help="N > 1 only: total size of the extra sharded MSM of BASELINE configs[4] (0 = skip)")
    ap.add_argument("--check-log-n", type=int, default=12,
                    help="N > 1 only: per-rank size of the fold == concatenated-multiexp check")
    ap.add_argument("--timed-steps-only", action="store_true",
                    help="run only warm-up + the K timed steps (no overlapped / PCIe extras): the command profiled for "
                         "profiles/*kernel_stats.csv, so that rocprof's per-kernel average matches the live HIP-event figure")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    distributed = world > 1
    # BENCH_FORCE_COLLECTIVE=1 at N = 1 (under torch.distributed.run): init the process group and run the per-step
    # all-gather + fold anyway - executes the RCCL code path on a single-GPU box (tests/test_gpu_round3.py)
    collective = distributed or (os.environ.get("BENCH_FORCE_COLLECTIVE") == "1" and "RANK" in os.environ)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU fallback in bellman_amd)")
    # one rank per GPU; BENCH_BACKEND=gloo lets several ranks share one GPU to smoke-test the N>1
    # path on a single-GPU box (RCCL itself needs one device per rank)
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    if collective:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)

    import bellman_amd
    from bellman_amd import _lib, sharding

    lib = _lib.load()
    worker = bellman_amd.Worker(device_index)
    n = 1 << args.log_n

    # ---- who is running: every rank reports its device, rank 0 prints the census (the N > 1 line proves by itself
    # that N ranks ran on N distinct devices; src/multicore.rs:21-92 is "all the cores of one box") ----------------
    props = torch.cuda.get_device_properties(device_index)
    me = {"rank": rank, "local_rank": local_rank, "device_index": device_index, "name": props.name,
          "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None),
          "pci_device_id": getattr(props, "pci_device_id", None), "host": os.uname().nodename, "pid": os.getpid()}
    census = [me]
    if collective:
        census = [None] * world
        dist.all_gather_object(census, me)
    distinct_devices = len({(c["host"], c["uuid"] or c["device_index"], c["pci_bus_id"]) for c in census})
    if collective and backend == "nccl" and world > 1:
        assert distinct_devices == world, "RCCL run with %d ranks on %d distinct devices: %r" % (world, distinct_devices, census)

    # ---- synthetic inputs, generated by the PRODUCT on the device (no oracle involved) --------
    # bases P_i = [t_i] G for SplitMix-derived t_i (distinct, prime-order, never the identity);
    # each rank owns a different shard of the (virtual) world*n-term problem.
    t_host = splitmix_scalars(n, 0x62656C6C6D616E + rank * 4 * n)
    t_dev = torch.from_numpy(t_host.view(np.int64)).cuda()
    bases_dev = torch.empty((n, 12), dtype=torch.int64, device="cuda")
    rc = lib.bh_fixed_base_mul_dev(worker.ctx, 1, G1_GEN_MONT.ctypes.data_as(ctypes.c_void_p),
                                   ctypes.c_void_p(t_dev.data_ptr()), n, 0, ctypes.c_void_p(bases_dev.data_ptr()), None)
    assert rc == 0
    worker.synchronize()
    # registered like a CRS query (bh_bases_copy_dev: the handle owns its device copy and - a G1 vector of 2^19 ... 2^22
    # points, since round 6 - its 13-row 20-bit window table at a 128-byte record stride: 1.7 GB for 2^20 points, built once
    # at registration in ~0.13 s; like the base upload it is outside the timed region, SURVEY.md 8d "excl. one-time base upload")
    treg0 = time.perf_counter()
    bases = bellman_amd.Bases.copy_device(worker, 1, ctypes.c_void_p(bases_dev.data_ptr()), n)
    worker.synchronize()
    register_ms = (time.perf_counter() - treg0) * 1e3
    table_bits, table_rows, table_bytes = bases.table_info()
    s_host = splitmix_scalars(n, 0x5CA1A25 + rank * 4 * n)
    s_dev = torch.from_numpy(s_host.view(np.int64)).cuda()
    torch.cuda.synchronize()

    ab_flags = int(os.environ.get("BH_BENCH_FLAGS", "0"), 0)   # bh_msm_opts.flags for A/B runs of a kernel variant (profiles/archive/tools_r5/*.sh, profiles/archive/tools_history/)

    job_stats = {}

    def step():
        w = bellman_amd.multiexp(worker, bases, bellman_amd.FullDensity(), None, scalars_dev=ctypes.c_void_p(s_dev.data_ptr()),
                                 n=n, stats=True, flags=ab_flags)
        part, ms, st = w.wait()
        job_stats.update(st)   # what the job executed, counted on the device (bh_msm_wait_stats)
        if collective:   # one 96-byte all-gather over RCCL + local fold (bellman_amd/sharding.py)
            return sharding.fold_partials(part, 1, device=coll_dev if coll_dev == "cuda" else None), ms
        return part, ms

    for _ in range(args.warmup):
        result, _ = step()

    def barrier():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    stage = np.zeros(4)
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        result, ms = step()
        step_ms.append((time.perf_counter() - ts) * 1e3)
        stage += np.array(ms)
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [elapsed * 1e3 / max(args.steps, 1)]
    if collective:
        per_rank_ms = [None] * world
        dist.all_gather_object(per_rank_ms, elapsed * 1e3 / max(args.steps, 1))
        tt = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    stage /= max(args.steps, 1)
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)
    value = world * n * args.steps / elapsed / 1e6

    # extra information (not `value`): the same K steps issued with two jobs in flight, the way
    # create_proof drives multiexp (prover.rs:244-318 issues all eight before the first wait): the
    # latency-bound reduction of one MSM overlaps the bucket accumulation of the next
    def issue():
        return bellman_amd.multiexp(worker, bases, bellman_amd.FullDensity(), None,
                                    scalars_dev=ctypes.c_void_p(s_dev.data_ptr()), n=n, timed=True, flags=ab_flags)

    pipelined_value = pcie_value = None
    extras = not args.timed_steps_only
    barrier()
    tp0 = time.perf_counter()
    if extras:
        prev = issue()
        for _ in range(args.steps - 1):
            nxt = issue()
            prev.wait()
            prev = nxt
        prev.wait()
        barrier()
        pipelined_value = world * n * args.steps / (time.perf_counter() - tp0) / 1e6
    # PCIe-inclusive figure (never `value`): scalars handed over as a HOST buffer on every call
    # (bh_msm_async), as the Rust shim would do; bases stay registered in HBM
    barrier()
    th0 = time.perf_counter()
    if extras:
        for _ in range(max(1, min(args.steps, 5))):
            bellman_amd.multiexp(worker, bases, bellman_amd.FullDensity(), s_host).wait()
        pcie_value = n * max(1, min(args.steps, 5)) / (time.perf_counter() - th0) / 1e6

    # the classic plan on the same handle (16 windows over the plain vector, no table; BH_MSM_NO_TABLE): what `value` was
    # measured on until round 6.  (A vector registered WITH a table has no 128-byte-stride copy of its points, which the
    # classic plan's gathers gain 1-3 % from: BELLMAN_HIP_TABLE_MAX_LOG2_G1=18 reproduces the earlier default exactly.)
    classic_value = classic_ms = None
    if extras and table_rows:
        from bellman_amd.multiexp import NO_TABLE
        cl = []
        for _ in range(max(3, min(args.steps, 10))):
            tc0 = time.perf_counter()
            bellman_amd.multiexp(worker, bases, bellman_amd.FullDensity(), None, scalars_dev=ctypes.c_void_p(s_dev.data_ptr()),
                                 n=n, flags=ab_flags | NO_TABLE).wait()
            cl.append((time.perf_counter() - tc0) * 1e3)
        classic_ms = float(np.median(cl[1:]))
        classic_value = n / classic_ms / 1e3

    sharded_check = None
    if distributed:
        sharded_check = check_sharded_fold(worker, lib, world, rank, coll_dev, args.check_log_n)
    sharded_proof = None
    if distributed and not args.no_proof:
        sharded_proof = bench_create_proof_sharded(worker, args.proof_log_n, world, rank, coll_dev)
    # BASELINE.json configs[4]: ONE 2^26-term G1 MSM sharded by bases over all ranks (strong scaling of a
    # fixed problem; extra information, not `value`)
    c5 = None
    if distributed and args.c5_log_n:
        n5 = (1 << args.c5_log_n) // world
        t5 = torch.from_numpy(splitmix_scalars(n5, 0xC5 + rank * 8 * n5).view(np.int64)).cuda()
        b5 = torch.empty((n5, 12), dtype=torch.int64, device="cuda")
        assert lib.bh_fixed_base_mul_dev(worker.ctx, 1, G1_GEN_MONT.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(t5.data_ptr()),
                                         n5, 0, ctypes.c_void_p(b5.data_ptr()), None) == 0
        worker.synchronize()
        # registered like a CRS query: a shard of up to 2^24 points gets its 20-bit window table (round 6) - 8 ranks: 2^23 each
        bases5 = bellman_amd.Bases.copy_device(worker, 1, ctypes.c_void_p(b5.data_ptr()), n5)
        s5 = torch.from_numpy(splitmix_scalars(n5, 0x5CA1A25 + rank * 8 * n5).view(np.int64)).cuda()

        def step5():
            part = bellman_amd.multiexp(worker, bases5, bellman_amd.FullDensity(), None,
                                        scalars_dev=ctypes.c_void_p(s5.data_ptr()), n=n5).wait()
            return sharding.fold_partials(part, 1, device=coll_dev if coll_dev == "cuda" else None)

        step5()
        barrier()
        t0 = time.perf_counter()
        steps5 = 3
        for _ in range(steps5):
            step5()
        barrier()
        tt = torch.tensor([time.perf_counter() - t0], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        e5 = float(tt.item())
        # the model of DESIGN.md 7, on this run's own numbers: T(N) = the slowest rank's shard alone + 0.1 ms for the all-gather and fold
        barrier()
        alone = []
        for _ in range(steps5):
            ta = time.perf_counter()
            bellman_amd.multiexp(worker, bases5, bellman_amd.FullDensity(), None, scalars_dev=ctypes.c_void_p(s5.data_ptr()), n=n5).wait()
            alone.append((time.perf_counter() - ta) * 1e3)
        ta = torch.tensor([float(np.median(alone))], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(ta, op=dist.ReduceOp.MAX)
        c5 = {"workload": "one G1 MSM of 2^%d terms sharded by bases over %d ranks (%d terms per rank) + all-gather + fold"
                          % (args.c5_log_n, world, n5),
              "value": round(world * n5 * steps5 / e5 / 1e6, 3), "unit": "Mscalar-mul/s", "ms_per_msm": round(e5 * 1e3 / steps5, 2),
              "scaling": "strong", "steps": steps5,
              "model": {"predicted_ms": round(float(ta.item()) + 0.1, 2), "measured_ms": round(e5 * 1e3 / steps5, 2),
                        "terms": "slowest rank's shard alone (median of %d, ranks running side by side) + 0.1 ms assumed for the "
                                 "all-gather of %d x 96 B and the fold; the N = 1 line's scaling_model holds the single-GPU "
                                 "prediction for every N" % (steps5, world)}}
        del bases5, b5, t5, s5
        worker.trim()
    # one PROCESS driving every GPU of the node (bh_msm_sharded_async: one context per device, the reference's single
    # process with all cores of one box, src/multicore.rs:21-92): rank 0 runs it while the other ranks wait at the barrier
    one_process = None
    if distributed and args.c5_log_n:
        barrier()
        if rank == 0:
            one_process = bench_one_process_sharded(lib, min(world, torch.cuda.device_count()) if backend == "nccl" else world,
                                                    torch.cuda.device_count(), min(args.c5_log_n, 22))
        barrier()
    out = None
    if rank == 0:
        acc_ms = float(stage[2])
        achieved = n * BYTES_PER_TERM_G1 / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        out = {
            "metric": "G1 MSM Mscalar-mul/s (BLS12-381, 2^%d bases per GPU)" % args.log_n,
            "value": round(value, 3),
            "unit": "Mscalar-mul/s",
            "n_gpus": world,
            "ranks_seen": len(census),
            "distinct_devices": distinct_devices,
            "ms_per_step_per_rank": [round(float(x), 4) for x in per_rank_ms],
            "ranks": [{k: c[k] for k in ("rank", "device_index", "uuid", "pci_bus_id", "host")} for c in census],
            "backend": (backend + (" (RCCL %s)" % ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else "")) if collective else None,
            "library": _lib.library_identity(),
            # SURVEY.md 8(d) words the metric "incl. scalar upload": the same multiexp with the 32 MiB of scalars handed over
            # as a HOST buffer on every call (what an unpatched prover.rs issues); `value` is the resident-scalar figure
            "value_incl_scalar_upload": round(world * pcie_value, 3) if pcie_value else None,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32x12 (381-bit Montgomery integers)",
            "data": "synthetic",
            "config": {
                "workload": "G1 Pippenger MSM, 2^%d (base,scalar) terms per GPU, FullDensity, inputs resident in HBM "
                            "(BASELINE.json configs[1])" % args.log_n,
                "plan": ("bases registered once with their %d-row %d-bit window table (%.2f GB in HBM at a 128-byte record stride; "
                         "registration incl. the table build %.0f ms, outside the timed region like the base upload): every digit of a "
                         "scalar goes to ONE set of 2^%d buckets" % (table_rows, table_bits, table_bytes / 1e9, register_ms, table_bits - 1))
                        if table_rows else "classic plan: 16 windows of 2^15 buckets over the plain base vector",
                "value_classic_plan_no_table_per_gpu": round(classic_value, 3) if classic_value else None,
                "ms_per_step_classic_plan_no_table": round(classic_ms, 4) if classic_ms else None,
                "sharding": "bases split across ranks, one 96-B all-gather per step" if distributed else "single GPU",
                "collective": ("%s%s world_size=%d" % (backend, " (RCCL)" if backend == "nccl" else "", world)) if collective else None,
                "device_ms": {"pipeline": round(float(stage[0]), 4), "digits_sort": round(float(stage[1]), 4),
                              "bucket_accumulate": round(acc_ms, 4), "merge_reduce": round(float(stage[3]), 4)},
                "headline": "`value` = terms of all ranks / wall time of the K timed steps (mean, the driver's contract), scalars and "
                            "bases RESIDENT in HBM - the state multiexp is called in inside create_proof, where the scalars are "
                            "produced on the device; the PCIe-inclusive rate (32 MiB of host scalars per call) is the last field",
                "ms_per_step_median": round(float(np.median(step_ms)), 4),
                "value_at_median_step": round(world * n / float(np.median(step_ms)) / 1e3, 3),
                "value_with_2_jobs_in_flight": round(pipelined_value, 3) if pipelined_value else None,
                "value_per_gpu_with_host_scalars_pcie_inclusive": round(pcie_value, 3) if pcie_value else None,
                # SURVEY.md 8(d) defines the metric "excl. one-time base upload; incl. scalar upload": THIS is that figure; `value` is
                # the resident-scalar rate (the state multiexp is called in inside create_proof)
                "value_incl_scalar_upload_survey_8d": round(world * pcie_value, 3) if pcie_value else None,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "msm_accumulate_kernel<FpOps>",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "traffic": None,
                "note": "integer-ALU bound (v_mad_u64_u32), see DESIGN.md; bytes = 128 B/term x 2^%d" % args.log_n,
                # the roofline that actually bounds this kernel: the v_mad_u64_u32 pipe.  Work per launch =
                # W*n mixed additions x 10 field products x 351 mads (ff.cuh); peak = 26.2 T mad/s measured
                # on this chip (profiles/r1_microbench_int.txt).
                "alu": {"unit": "Tmad/s", "peak": MAD_PEAK_T,
                        "achieved": round(job_stats["mixed_additions"] * MADS_PER_MIXED_ADD / (acc_ms * 1e-3) / 1e12, 2) if acc_ms > 0 else 0.0,
                        "frac": round(job_stats["mixed_additions"] * MADS_PER_MIXED_ADD / (acc_ms * 1e-3) / 1e12 / MAD_PEAK_T, 4) if acc_ms > 0 else 0.0,
                        "mixed_additions_per_launch": job_stats["mixed_additions"],
                        "sorted_entries": job_stats["sorted_entries"], "zero_digits": job_stats["zero_digits"],
                        "bucket_and_chunk_openers": job_stats["sorted_entries"] - job_stats["zero_digits"] - job_stats["mixed_additions"],
                        "plan": {"window_bits": job_stats["window_bits"], "chunk": job_stats["chunk"], "bucket_sets": job_stats["bucket_sets"]},
                        "work": "mixed additions the launch EXECUTED into a non-empty accumulator, counted on the device "
                                "(bh_msm_wait_stats; the entries that open a bucket or a chunk partial are copies) x (8 Fp products x 351 "
                                "+ 2 squarings x 273 - 182: the last two products share one reduction) = %d v_mad_u64_u32 / "
                                "v_mul_lo_u32 per addition" % MADS_PER_MIXED_ADD,
                        "peak_provenance": "v_mad_u64_u32 microbenchmark on this chip (tools/microbench_int.hip, "
                                           "profiles/r1_microbench_int.txt) at its sustained clock; under the accumulate kernel the "
                                           "SQ counters put the clock near 2.0 GHz (profiles/archive/r1_pmc_valu.json), so the fraction is "
                                           "against a peak measured at a higher clock (conservative)"},
            },
        }
        if not args.no_cpu_baseline and not distributed:   # rank 0 at N=1 only
            # CPU baseline: the oracle restatement of bellman's multicore path, timed here as the
            # reported baseline (never the thing shipped).  Sample = the same 2^log_n-term MSM, once.
            from oracle import cref

            bases_host = bases_dev.cpu().numpy().view(np.uint64)
            threads = cref.lib().orc_max_threads()
            c_ref = cref.window_size(n)
            windows = (255 + c_ref - 1) // c_ref
            t1 = time.perf_counter()
            rc, want = cref.multiexp(1, bases_host, 0, None, s_host, threads=threads)
            cpu_s = time.perf_counter() - t1
            assert rc == 0
            if not distributed:
                assert np.array_equal(result, want), "GPU MSM result differs from the CPU oracle"
            out["cpu_baseline"] = {
                "value": round(n / cpu_s / 1e6, 4),
                "unit": "Mscalar-mul/s",
                "cores": min(threads, windows),
                "kind": "port",
                "sample": "same 2^%d-term G1 MSM, 1 run, C restatement of bellman's rayon path "
                          "(c=%d, %d window tasks, %d host threads available)" % (args.log_n, c_ref, windows, threads),
            }
        if sharded_check is not None:
            out["sharded_fold_check"] = sharded_check
        if sharded_proof is not None:
            out["create_proof_sharded"] = sharded_proof
        if c5 is not None:
            out["msm_c5_sharded"] = c5
        if one_process is not None:
            out["msm_one_process_sharded"] = one_process
        if not args.no_proof and not distributed:
            out["msm_boolean_heavy"] = bench_msm_boolean(worker, lib, bases, t_host, n, ms_per_step, check=not args.no_cpu_baseline)
            out["msm_other_shapes"] = [bench_msm_shape(worker, lib, 2, 19), bench_msm_shape(worker, lib, 2, 20),
                                       bench_msm_shape(worker, lib, 1, 16), bench_msm_shape(worker, lib, 1, 18)]
            out["create_proof_mimc"] = bench_mimc(worker, cpu_baseline=not args.no_cpu_baseline)
            out["fft"] = bench_fft(worker, lib)
            out["create_proof"] = bench_create_proof(worker, lib, args.proof_log_n, cpu_baseline=not args.no_cpu_baseline)
            out["setup"] = out["create_proof"]["setup"]   # (also at the top level: the set-up path of the config C4 proof)
            out["create_proof_boolean"] = bench_create_proof_boolean(worker, lib, args.proof_log_n)
            if args.c5_proof_log_n:
                out["create_proof_c5"] = bench_create_proof_c5(worker, args.c5_proof_log_n)
            # what one rank of an N-rank run would do, measured on this GPU: the model the first SCALE record is read against
            out["scaling_model"] = bench_scaling_model(worker, lib, 26, args.c5_proof_log_n)
        # measured HBM traffic of the dominant kernel, recorded from the rocprofv3 --pmc passes of this same command
        # (tools/r6/gpu_final.sh -> profiles/r6_final_pmc_accumulate.json).  Used ONLY if the file was taken at the plan this
        # run executed (log_n, window bits, chunk length - bh_msm_wait_stats): a stale file leaves `traffic` null.  The guide's
        # x2 read correction is calibrated for wide coalesced reads; this kernel gathers 96-byte records in 16-byte pieces, so
        # `traffic` carries the read-corrected (doubled FETCH_SIZE) figure and the note the raw one.
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r6_final_pmc_accumulate.json")))
        except Exception:
            pmc = None
        if pmc is not None:
            same = (pmc.get("log_n") == args.log_n and pmc.get("window_bits") == job_stats.get("window_bits") and
                    pmc.get("chunk") == job_stats.get("chunk"))
            if same:
                fetch, write = pmc["FETCH_SIZE"]["per_launch_kb_mean"] * 1024, pmc["WRITE_SIZE"]["per_launch_kb_mean"] * 1024
                out["roofline"]["traffic"] = int(2 * fetch + write)
                out["roofline"]["traffic_note"] = (
                    "bytes per launch from profiles/r6_final_pmc_accumulate.json (same plan: c = %d, K = %d): 2 x FETCH_SIZE (gfx950 read "
                    "correction of MI355X_MICROARCH.md, calibrated on wide coalesced reads; this kernel's reads are 16-byte pieces of "
                    "gathered 96-byte records, so it is an upper bound - uncorrected: %d) + WRITE_SIZE %d; algorithmic 128 B x 2^%d = %d"
                    % (job_stats["window_bits"], job_stats["chunk"], int(fetch + write), int(write), args.log_n, 128 << args.log_n))
            else:
                out["roofline"]["traffic_note"] = ("profiles/r6_final_pmc_accumulate.json was taken at log_n %s, c %s, K %s - not the plan "
                                                   "of this run (c %s, K %s): ignored" % (pmc.get("log_n"), pmc.get("window_bits"), pmc.get("chunk"),
                                                                                        job_stats.get("window_bits"), job_stats.get("chunk")))
        print(json.dumps(out), flush=True)
    if collective:
        dist.barrier()
        dist.destroy_process_group()
    worker.close()


if __name__ == "__main__":
    main()
