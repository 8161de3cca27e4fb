"""Turn the raw output of tools/gpu_final.sh (gpurun_out/<dir>) into the small files kept under profiles/:
   python tools/summarize_final.py gpurun_out/r2final r2_final
writes profiles/<tag>_bench.json, <tag>_bench_msm_kernel_stats.csv, <tag>_pmc_accumulate.json, <tag>_pmc_valu.json,
<tag>_<workload>_kernel_stats.csv, <tag>_sizes.txt, <tag>_fft.txt, <tag>_mimc.txt, <tag>_gputests.txt."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc_means(path, kernel_substr):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if kernel_substr in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    cp = lambda a, b: os.path.exists(os.path.join(src, a)) and shutil.copy(os.path.join(src, a), os.path.join(prof, tag + "_" + b))
    line = [ln for ln in open(os.path.join(src, "bench.json")) if ln.startswith("{")][-1]
    json.dump(json.loads(line), open(os.path.join(prof, tag + "_bench.json"), "w"), indent=1)
    cp("prof_bench/p_kernel_stats.csv", "bench_msm_kernel_stats.csv")
    cp("gputests.txt", "gputests.txt")
    cp("fft.txt", "fft.txt")
    cp("mimc.txt", "mimc.txt")
    cp("pmc_extra.json", "pmc_g2_pairs_and_fft.json")
    cp("proof.txt", "proof_2p20.txt")
    cp("boolean_mix.txt", "boolean_mix.txt")
    cp("proof_timeline.txt", "proof_timeline.txt")
    with open(os.path.join(prof, tag + "_sizes.txt"), "w") as f:
        for name in ("sizes_g1.txt", "sizes_g1_large.txt", "sizes_g2.txt"):
            if os.path.exists(os.path.join(src, name)):
                f.write(open(os.path.join(src, name)).read())
    for d in sorted(os.listdir(src)):
        if d.startswith("prof_") and d != "prof_bench" and os.path.isdir(os.path.join(src, d)):
            cp(d + "/p_kernel_stats.csv", d[5:] + "_kernel_stats.csv")
    K = "msm_accumulate_kernel<bh::FpOps, false>"
    f, w = pmc_means(os.path.join(src, "pmc_fetch/p_counter_collection.csv"), K), pmc_means(os.path.join(src, "pmc_write/p_counter_collection.csv"), K)
    # the plan the profiled command ran, as its own bench line reports it (bh_msm_wait_stats): bench.py refuses this file
    # for a run whose plan differs
    plan = json.loads(line)["roofline"]["alu"]["plan"]
    out = {
        "kernel": "msm_accumulate_kernel<FpOps,false>",
        "workload": "G1 MSM 2^20, signed digits c=%d (%d bucket sets of 2^%d buckets), K=%d (bench.py default plan)" %
                    (plan["window_bits"], plan["bucket_sets"], plan["window_bits"] - 1, plan["chunk"]),
        "log_n": 20, "window_bits": plan["window_bits"], "chunk": plan["chunk"],
        "FETCH_SIZE": {"per_launch_kb_mean": f["FETCH_SIZE"][0], "launches": f["FETCH_SIZE"][1]},
        "WRITE_SIZE": {"per_launch_kb_mean": w["WRITE_SIZE"][0], "launches": w["WRITE_SIZE"][1]},
        "algorithmic_bytes_per_launch": 128 << 20,
        "note": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace, separate passes, of `python bench.py --steps 10 --warmup 2 "
                "--no-cpu-baseline --no-proof --timed-steps-only`; units KB as reported.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half "
                "the bytes of a WIDE COALESCED read and is uncalibrated for other access widths; this kernel's reads are per-lane gathers "
                "of 96-byte records in 16-byte pieces (6 requests per 128-byte line pair), so the raw value is kept as a LOWER bound "
                "and the doubled value as an upper bound of the read traffic",
    }
    out["traffic_bytes_lower"] = int((out["FETCH_SIZE"]["per_launch_kb_mean"] + out["WRITE_SIZE"]["per_launch_kb_mean"]) * 1024)
    out["traffic_bytes_upper"] = int((2 * out["FETCH_SIZE"]["per_launch_kb_mean"] + out["WRITE_SIZE"]["per_launch_kb_mean"]) * 1024)
    json.dump(out, open(os.path.join(prof, tag + "_pmc_accumulate.json"), "w"), indent=1)
    v = pmc_means(os.path.join(src, "pmc_valu/p_counter_collection.csv"), K)
    m = {k: x[0] for k, x in v.items()}
    adds = json.loads(line)["roofline"]["alu"]["mixed_additions_per_launch"] / 64   # executed, counted on the device
    json.dump({
        "kernel": "msm_accumulate_kernel<FpOps,false>", "workload": out["workload"], "per_launch_mean": m,
        "derived": {
            "wave_level_mixed_additions": adds,
            "valu_instructions_per_mixed_addition": m["SQ_INSTS_VALU"] / adds,
            "int64_valu_instructions_per_mixed_addition": m.get("SQ_INSTS_VALU_INT64", 0) / adds,
            "VALUBusy_percent_4_cycles_per_instruction": 100 * m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (m["GRBM_GUI_ACTIVE"] / 8),
            "SIMD_cycles_per_VALU_instruction": (m["GRBM_GUI_ACTIVE"] / 8) / (m["SQ_INSTS_VALU"] / 1024),
        }}, open(os.path.join(prof, tag + "_pmc_valu.json"), "w"), indent=1)
    print(json.dumps(out, indent=1)[:600])


main()


def write_summary(tag):
    """profiles/<tag>_summary.md from the files main() wrote."""
    prof = os.path.join(ROOT, "profiles")
    d = json.load(open(os.path.join(prof, tag + "_bench.json")))
    cp = d["create_proof"]; r = cp["with_r1cs_resident_in_hbm"]; mm = d["create_proof_mimc"]; rf = d["roofline"]; dm = d["config"]["device_ms"]
    pmc = json.load(open(os.path.join(prof, tag + "_pmc_accumulate.json")))
    tests = [ln for ln in open(os.path.join(prof, tag + "_gputests.txt")) if " passed" in ln]
    L = []
    A = L.append
    A("# Final run of %s on one MI355X (tools/gpu_%s.sh, summarised by tools/summarize_final.py)\n" % (tag, tag.replace("_final", "_final") if tag.startswith("r") else "final"))
    A("`pytest tests -m gpu`: %s (`%s_gputests.txt`); `__graft_entry__.smoke()`: ok.\n" % (tests[-1].strip() if tests else "?", tag))
    A("## bench.py (default flags; `%s_bench.json`)\n" % tag)
    A("| item | value |\n|---|---|")
    A("| `value` (G1 MSM 2^20, inputs resident in HBM, mean of %d steps) | %.1f M scalar-mul/s, %.3f ms per step (median %.3f ms) |" % (d["steps"], d["value"], d["ms_per_step"], d["config"]["ms_per_step_median"]))
    A("| device stages (HIP events inside the library) | pipeline %.2f = digits+sort %.2f + accumulate %.2f + merge/reduce %.2f ms |" % (dm["pipeline"], dm["digits_sort"], dm["bucket_accumulate"], dm["merge_reduce"]))
    A("| two jobs in flight / host scalars (PCIe inclusive) | %.1f / %.1f M scalar-mul/s |" % (d["config"]["value_with_2_jobs_in_flight"], d["config"]["value_per_gpu_with_host_scalars_pcie_inclusive"]))
    A("| roofline (HBM) | %.1f GB/s algorithmic of 8000 = %.4f; traffic per launch %.2f GB read-corrected (upper bound) / %.2f GB uncorrected vs 0.134 GB algorithmic (`%s_pmc_accumulate.json`) |" % (rf["achieved"], rf["frac"], pmc["traffic_bytes_upper"] / 1e9, pmc["traffic_bytes_lower"] / 1e9, tag))
    A("| roofline (integer ALU) | %.2f T mad/s of %.1f measured peak = %.3f |" % (rf["alu"]["achieved"], rf["alu"]["peak"], rf["alu"]["frac"]))
    A("| CPU baseline (C restatement, %d window tasks) | %.2f M scalar-mul/s |" % (d["cpu_baseline"]["cores"], d["cpu_baseline"]["value"]))
    for s in d["msm_other_shapes"]:
        A("| %s MSM 2^%d | %.2f ms wall median (device %.2f: sort %.2f, accumulate %.2f, reduce %.2f) |" % (s["group"], s["log_n"], s["ms_median"], s["device_ms"]["pipeline"], s["device_ms"]["digits_sort"], s["device_ms"]["bucket_accumulate"], s["device_ms"]["merge_reduce"]))
    A("| create_proof MiMC-322 (C1) | %.2f ms median of %d (%.0f proofs/s); CPU restatement on %d threads: %.1f proofs/s |" % (mm["ms_median"], mm["samples"], mm["proofs_per_s"], mm["cpu_baseline"]["cores"], mm["cpu_baseline"]["value"]))
    f = d["fft"]
    A("| FFT 2^22 (C3) fft / ifft / coset_fft / icoset_fft | %.3f / %.3f / %.3f / %.3f ms = %.0f / %.0f / %.0f / %.0f GB/s algorithmic |" % tuple([f[k]["ms"] for k in ("fft", "ifft", "coset_fft", "icoset_fft")] + [f[k]["algorithmic_GBps"] for k in ("fft", "ifft", "coset_fft", "icoset_fft")]))
    A("| create_proof 2^20 constraints (C4), host synthesis as in the reference | %.2f proofs/s (%.1f ms: synthesis %.1f, issue %.1f, waits %.1f); 12 host threads: %.1f proofs/s |" % (cp["proofs_per_s"], cp["ms_total"], cp["ms_host_synthesis"], cp["ms_issue_7_multiexps_then_h_block_incl_uploads"], cp["ms_h_multiexp_and_waits"], cp["proofs_per_s_concurrent"]))
    A("| ... constraint matrices resident in HBM | %.2f proofs/s (%.1f ms: witness %.1f, issue %.1f, waits %.1f); 12 host threads: %.1f proofs/s |" % (r["proofs_per_s"], r["ms_total"], r["ms_host_witness"], r["ms_issue_7_multiexps_then_h_block_incl_uploads"], r["ms_h_multiexp_and_waits"], r["proofs_per_s_concurrent"]))
    A("| ... CPU restatement of the prover, %d host threads | %.4f proofs/s (%.2f s) |" % (cp["cpu_baseline"]["cores"], cp["cpu_baseline"]["value"], cp["cpu_baseline"]["seconds"]))
    A("\n## Per-size tables (`%s_sizes.txt`, tools/profile_suite.py sizes), FFT (`%s_fft.txt`), MiMC (`%s_mimc.txt`)\n" % (tag, tag, tag))
    A("```\n" + open(os.path.join(prof, tag + "_sizes.txt")).read() + open(os.path.join(prof, tag + "_fft.txt")).read() + open(os.path.join(prof, tag + "_mimc.txt")).read() + "```\n")
    extra = os.path.join(prof, tag + "_targets.md")
    if os.path.exists(extra):
        A(open(extra).read())
    open(os.path.join(prof, tag + "_summary.md"), "w").write("\n".join(L) + "\n")


if len(sys.argv) > 2:
    write_summary(sys.argv[2])
