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
    with open(os.path.join(prof, tag + "_sizes.txt"), "w") as f:
        for name in ("sizes_g1.txt", "sizes_g1_large.txt", "sizes_g2.txt"):
            if os.path.exists(os.path.join(src, name)):
                f.write(open(os.path.join(src, name)).read())
    for d in sorted(os.listdir(src)):
        if d.startswith("prof_") and d != "prof_bench" and os.path.isdir(os.path.join(src, d)):
            cp(d + "/p_kernel_stats.csv", d[5:] + "_kernel_stats.csv")
    K = "msm_accumulate_kernel<bh::FpOps, false>"
    f, w = pmc_means(os.path.join(src, "pmc_fetch/p_counter_collection.csv"), K), pmc_means(os.path.join(src, "pmc_write/p_counter_collection.csv"), K)
    out = {
        "kernel": "msm_accumulate_kernel<FpOps,false>",
        "workload": "G1 MSM 2^20, signed digits c=16 (W=16, 2^15 buckets per window), K=32 (bench.py default)",
        "log_n": 20,
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
    adds = 16 * (1 << 20) / 64
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
