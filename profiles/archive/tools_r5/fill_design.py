"""Fill the measurement table of DESIGN.md §6 (the @@TOKENS@@) from profiles/r5_final_bench.json + r5_final_fft.txt +
r5_final_gputests.txt:   python tools/r5/fill_design.py"""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = os.path.join(ROOT, "profiles")
d = json.load(open(os.path.join(P, "r5_final_bench.json")))
cfg, rf, cp, mm, f = d["config"], d["roofline"], d["create_proof"], d["create_proof_mimc"], d["fft"]
dm = cfg["device_ms"]
r = cp["with_r1cs_resident_in_hbm"]
cs = cp["drop_in_call_sites"]
shapes = {(s["group"], s["log_n"]): s for s in d["msm_other_shapes"]}
fft_txt = open(os.path.join(P, "r5_final_fft.txt")).read()
def fft_line(log_n):
    ms = [float(m) for m in re.findall(r"log_n=%d \S+\s+median ([\d.]+) ms" % log_n, fft_txt)]
    return " / ".join("%.3f" % x for x in ms)
tests = [ln.strip() for ln in open(os.path.join(P, "r5_final_gputests.txt")) if " passed" in ln]
c5 = d.get("create_proof_c5") or {}
sizes = open(os.path.join(P, "r5_final_sizes.txt")).read()
m26 = re.search(r"G1 log_n=26\s+wall median ([\d.]+) ms\s+device total ([\d.]+) ms\s+sort ([\d.]+)\s+accumulate ([\d.]+)\s+reduce ([\d.]+)", sizes)
tok = {
    "C2": "**%.1f M scalar-mul/s, %.3f ms per step** (median %.3f ms; 2 jobs in flight %.1f M/s); round 4: 260.8 M/s, 4.02 ms" % (
        d["value"], d["ms_per_step"], cfg["ms_per_step_median"], cfg["value_with_2_jobs_in_flight"] or 0),
    "STAGES": "pipeline %.2f = digits + sort %.2f + accumulate %.2f + merge / reduce %.2f ms (HIP events inside the library); the rest of a step is the host tail" % (
        dm["pipeline"], dm["digits_sort"], dm["bucket_accumulate"], dm["merge_reduce"]),
    "ROOF": "%.1f GB/s algorithmic of 8000 = **%.4f**; traffic %.2f GB per launch (read-corrected) = %.0f x the algorithmic bytes; %.2f T mad/s of 26.2 = **%.3f**" % (
        rf["achieved"], rf["frac"], (rf.get("traffic") or 0) / 1e9, (rf.get("traffic") or 0) / (128 << 20), rf["alu"]["achieved"], rf["alu"]["frac"]),
    "PCIE": "%.1f M scalar-mul/s (`value_incl_scalar_upload`)" % (d.get("value_incl_scalar_upload") or cfg.get("value_per_gpu_with_host_scalars_pcie_inclusive") or 0),
    "CPU": "%.2f M scalar-mul/s on %d window tasks (`cpu_baseline`, kind port); create_proof 2^20: %.4f proofs/s on %d threads" % (
        d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], cp["cpu_baseline"]["value"], cp["cpu_baseline"]["cores"]),
    "SHAPES": "%.2f / %.2f ms (accumulate %.2f / %.2f); G1 2^16 %.2f ms (reduce %.2f)" % (
        shapes[("G2", 19)]["ms_median"], shapes[("G2", 20)]["ms_median"], shapes[("G2", 19)]["device_ms"]["bucket_accumulate"],
        shapes[("G2", 20)]["device_ms"]["bucket_accumulate"], shapes[("G1", 16)]["ms_median"], shapes[("G1", 16)]["device_ms"]["merge_reduce"]),
    "FFT": "**%.3f / %.3f / %.3f / %.3f ms** = %.0f / %.0f / %.0f / %.0f GB/s algorithmic (%.1f-%.1f %% of 8 TB/s); round 4: 0.598-0.653 ms" % tuple(
        [f[k]["ms"] for k in ("fft", "ifft", "coset_fft", "icoset_fft")] + [f[k]["algorithmic_GBps"] for k in ("fft", "ifft", "coset_fft", "icoset_fft")] +
        [100 * min(f[k]["frac_of_8TBps"] for k in ("fft", "ifft", "coset_fft", "icoset_fft")), 100 * max(f[k]["frac_of_8TBps"] for k in ("fft", "ifft", "coset_fft", "icoset_fft"))]),
    "FFT2": "2^20: %s ms; 2^24: %s ms (fft / ifft / coset_fft / icoset_fft, median of 10 / 5)" % (fft_line(20), fft_line(24)),
    "MIMC": "%.2f ms median (%.0f proofs/s)" % (mm["ms_median"], mm["proofs_per_s"]),
    "C4": "%.2f proofs/s (%.1f ms: synthesis %.1f on one host thread, device part %.1f); two deep %.1f; 12 host threads %.1f proofs/s" % (
        cp["proofs_per_s"], cp["ms_total"], cp["ms_host_synthesis"], cp["ms_total"] - cp["ms_host_synthesis"], cp["proofs_per_s_one_caller_pipelined"], cp["proofs_per_s_concurrent"]),
    "C4R": "%.2f proofs/s (%.1f ms: witness %.1f); two deep %.1f; 12 host threads **%.1f proofs/s**; after synthesis %.1f proofs/s" % (
        r["proofs_per_s"], r["ms_total"], r["ms_host_witness"], r["proofs_per_s_one_caller_pipelined"], r["proofs_per_s_concurrent"], cp["proofs_per_s_excluding_host_synthesis"]),
    "CALLSITES": "**%.1f / %.1f / %.1f ms** (round 4: 21.9 / - / 155.5); the mirror's `bh_groth16_prove_assignment` %.1f ms; all four proofs bit-identical" % (
        cs["create_proof_via_patched_call_sites"]["ms_after_synthesis"], cs["create_proof_via_multiexp_and_fft_call_sites_only"]["ms_after_synthesis"],
        cs["create_proof_via_multiexp_and_fft_call_sites_only_round4_patch"]["ms_after_synthesis"], cs["bh_groth16_prove_assignment_same_inputs"]["ms_after_synthesis"]),
    "C5": "%s ms device (sort %s, accumulate %s, reduce %s; `r5_final_sizes.txt`); proof %.3f s (%.3f s host witness + **%.3f s GPU**)" % (
        (m26.group(2), m26.group(3), m26.group(4), m26.group(5)) + (c5.get("ms_total", 0) / 1e3, c5.get("ms_host_witness", 0) / 1e3, c5.get("ms_gpu_part", 0) / 1e3)) if m26 else "n/a",
    "TESTS": "%s (`r5_final_gputests.txt`); smoke ok" % (tests[-1] if tests else "?"),
}
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
for k, v in tok.items():
    s = s.replace("@@%s@@" % k, v)
left = re.findall(r"@@\w+@@", s)
open(p, "w").write(s)
print("filled", len(tok), "tokens; left:", left)
