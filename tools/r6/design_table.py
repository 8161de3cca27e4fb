#!/usr/bin/env python3
"""Prints the measurement table of DESIGN.md section 6 from profiles/<tag>_bench.json (+ <tag>_boolean_mix.txt, <tag>_sizes.txt,
<tag>_gputests.txt), so that every figure there is the file's.   usage: design_table.py [tag=r6_final]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r6_final"
P = lambda name: os.path.join(ROOT, "profiles", tag + "_" + name)
d = json.load(open(P("bench.json")))
c, rf, alu = d["config"], d["roofline"], d["roofline"]["alu"]
dm = c["device_ms"]
rows = []
A = lambda a, b: rows.append("| %s | %s |" % (a, b))
A("**C2** G1 multiexp 2^20, resident (`value`)", "**%.1f M scalar-mul/s, %.3f ms per step** (median %.3f; 2 jobs in flight %.1f M/s); the classic plan on the same handle (`BH_MSM_NO_TABLE`): %.1f M/s, %.3f ms"
  % (d["value"], d["ms_per_step"], c["ms_per_step_median"], c["value_with_2_jobs_in_flight"], c["value_classic_plan_no_table_per_gpu"], c["ms_per_step_classic_plan_no_table"]))
A("plan", c["plan"])
A("device stages of a step", "pipeline %.2f = digits + sort %.3f + accumulate %.2f + merge / reduce %.2f ms; the rest is the host tail" % (dm["pipeline"], dm["digits_sort"], dm["bucket_accumulate"], dm["merge_reduce"]))
tr = rf.get("traffic")
A("roofline (HBM / integer ALU)", "%.1f GB/s algorithmic of 8000 = **%.4f**; traffic %s; %.2f T mad/s of 26.2 = **%.3f** on %s executed additions (%s sorted entries, %s zero digits)"
  % (rf["achieved"], rf["frac"], ("%.2f GB per launch read-corrected = %.1f x the algorithmic bytes" % (tr / 1e9, tr / (128 << 20))) if tr else "null (PMC file of another plan)",
     alu["achieved"], alu["frac"], "{:,}".format(alu["mixed_additions_per_launch"]).replace(",", " "), "{:,}".format(alu["sorted_entries"]).replace(",", " "), alu["zero_digits"]))
A("incl. scalar upload (the SURVEY §8d wording; `config.value_incl_scalar_upload_survey_8d`)", "%.1f M scalar-mul/s" % c["value_incl_scalar_upload_survey_8d"])
mx = d["msm_boolean_heavy"]["mixes"]
A("**boolean-heavy, same bases** (`msm_boolean_heavy`): 50 % / 90 % booleans / all ones / 90 % < 2^8",
  "**%s ms = %s × the uniform rate**; merge + reduce %s ms" % (" / ".join("%.2f" % mx[k]["ms_median"] for k in ("bool50", "bool90", "ones", "small90")),
                                                              " / ".join("%.2f" % mx[k]["x_uniform_rate"] for k in ("bool50", "bool90", "ones", "small90")),
                                                              " / ".join("%.2f" % mx[k]["device_ms"]["merge_reduce"] for k in ("bool50", "bool90", "ones", "small90"))))
cb = d["cpu_baseline"]
cp = d["create_proof"]
A("CPU restatement of bellman's path", "%.2f M scalar-mul/s on %d window tasks (`cpu_baseline`, kind port); create_proof 2^20: %.4f proofs/s on %d threads"
  % (cb["value"], cb["cores"], cp["cpu_baseline"]["value"], cp["cpu_baseline"]["cores"]))
sh = {(s["group"], s["log_n"]): s for s in d["msm_other_shapes"]}
A("G2 2^19 / 2^20, G1 2^16 / 2^18", "%.2f / %.2f ms (merge + reduce %.2f); **%.3f / %.2f ms**" % (sh[("G2", 19)]["ms_median"], sh[("G2", 20)]["ms_median"], sh[("G2", 20)]["device_ms"]["merge_reduce"],
                                                                                              sh[("G1", 16)]["ms_median"], sh[("G1", 18)]["ms_median"]))
f = d["fft"]
A("**C3** FFT 2^22 fft / ifft / coset_fft / icoset_fft", "**%s ms** = %s GB/s algorithmic (%.1f-%.1f %% of 8 TB/s; %.2f-%.2f of the mad ceiling)"
  % (" / ".join("%.3f" % f[k]["ms"] for k in ("fft", "ifft", "coset_fft", "icoset_fft")), " / ".join("%.0f" % f[k]["algorithmic_GBps"] for k in ("fft", "ifft", "coset_fft", "icoset_fft")),
     100 * min(f[k]["frac_of_8TBps"] for k in ("fft", "ifft", "coset_fft", "icoset_fft")), 100 * max(f[k]["frac_of_8TBps"] for k in ("fft", "ifft", "coset_fft", "icoset_fft")),
     min(f[k]["frac_of_mad_ceiling"] for k in ("fft", "ifft", "coset_fft", "icoset_fft")), max(f[k]["frac_of_mad_ceiling"] for k in ("fft", "ifft", "coset_fft", "icoset_fft"))))
mi = d["create_proof_mimc"]
A("**C1** create_proof MiMC-322", "**%.3f ms median (%.0f proofs/s)**" % (mi["ms_median"], mi["proofs_per_s"]))
A("**C4** create_proof 2^20 constraints, host synthesis as in the reference", "%.2f proofs/s (%.1f ms: synthesis %.1f on one host thread, device part %.1f); two deep %.1f; %d host threads %.1f proofs/s"
  % (cp["proofs_per_s"], cp["ms_total"], cp["ms_host_synthesis"], cp["ms_total"] - cp["ms_host_synthesis"], cp["proofs_per_s_one_caller_pipelined"], cp["concurrent_host_threads"], cp["proofs_per_s_concurrent"]))
r = cp["with_r1cs_resident_in_hbm"]
ro = cp["roofline"]
A("… constraint matrices resident in HBM", "%.2f proofs/s (%.1f ms: witness %.1f, waits %.1f); two deep %.1f; %d host threads **%.1f proofs/s**; after synthesis %.1f proofs/s; `roofline`: %.1f T mad/s over the %.1f ms device part = %.2f"
  % (r["proofs_per_s"], r["ms_total"], r["ms_host_witness"], r["ms_h_multiexp_and_waits"], r["proofs_per_s_one_caller_pipelined"], cp["concurrent_host_threads"], r["proofs_per_s_concurrent"],
     cp["proofs_per_s_excluding_host_synthesis"], ro["achieved"], ro["device_ms"], ro["frac"]))
cs = cp["drop_in_call_sites"]
A("… after synthesis: prover.rs patched / multiexp + domain patched only / its round-4 form", "**%.1f / %.1f / %.0f ms** (C++ transcription of the Rust patch); all four proofs bit-identical"
  % (cs["create_proof_via_patched_call_sites"]["ms_after_synthesis"], cs["create_proof_via_multiexp_and_fft_call_sites_only"]["ms_after_synthesis"],
     cs["create_proof_via_multiexp_and_fft_call_sites_only_round4_patch"]["ms_after_synthesis"]))
b = d["create_proof_boolean"]
br = b["with_r1cs_resident_in_hbm"]
A("**boolean-heavy circuit**, 2^20-constraint domain (`create_proof_boolean`)", "host synthesis %.2f proofs/s (%.1f ms: synthesis %.1f); **R1CS resident %.1f proofs/s (%.1f ms: witness %.1f, device part %.1f)**"
  % (b["proofs_per_s"], b["ms_total"], b["ms_host_synthesis"], br["proofs_per_s"], br["ms_total"], br["ms_host_witness"], br["ms_device_part"]))
c5 = d["create_proof_c5"]
sizes = open(P("sizes.txt")).read()
m26 = re.search(r"G1 log_n=26 .*device total ([\d.]+) ms\s+sort ([\d.]+)\s+accumulate ([\d.]+)\s+reduce ([\d.]+)", sizes)
A("**C5** 2^26-term G1 multiexp; 2^24-constraint proof", "%s ms device (sort %s, accumulate %s, reduce %s; `%s_sizes.txt`); proof %.3f s (%.3f s host witness + %.3f s GPU)"
  % (m26.group(1), m26.group(2), m26.group(3), m26.group(4), tag, c5["ms_total"] / 1e3, c5["ms_host_witness"] / 1e3, c5["ms_gpu_part"] / 1e3))
g = open(P("gputests.txt")).read()
A("`pytest -m gpu`", "%s (`%s_gputests.txt`); smoke ok" % (re.findall(r"\d+ passed.*", g)[-1].strip(), tag))
print("| Item (final run of round 6; `profiles/%s_bench.json` unless noted) | Value |\n|---|---|" % tag)
print("\n".join(rows))
sm = d["scaling_model"]
print("\nscaling model:")
for n in ("1", "2", "4", "8"):
    a, pr = sm["msm_2p26_strong"][n], sm["proof_2p24_strong"][n]
    print("| %s | %.1f ms (%.0f M/s)%s | %.2f | %.0f ms (witness %.0f) | %.2f |" % (n, a["predicted_ms"], a["predicted_Mscalar_mul_per_s"], (", %d-row table" % a["window_table_rows"]) if a.get("window_table_rows") else "", a["predicted_speedup"],
                                                                   pr["predicted_ms"], pr["replicated_host_witness_ms"], pr["predicted_speedup"]))
print("\nsizes:")
for grp in ("G1", "G2"):
    print(grp, ", ".join("2^%s %.2f" % (m.group(1), float(m.group(2))) for m in re.finditer(r"%s log_n=(\d+)\s+wall median ([\d.]+)" % grp, sizes)))
