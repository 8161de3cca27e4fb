#!/bin/bash
# round 6, call 11: the reworked bench line + the bench smoke tests
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6c11
mkdir -p $OUT
t0=$(date +%s)
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"; tail -c 300 $OUT/bench.err
python - <<'PY'
import json, os
d = json.loads([l for l in open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r6c11/bench.json")) if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], d["config"]["device_ms"])
print("alu", {k: d["roofline"]["alu"][k] for k in ("achieved", "frac", "mixed_additions_per_launch", "bucket_and_chunk_openers", "plan")})
print("traffic", d["roofline"]["traffic"], d["roofline"].get("traffic_note", "")[:120])
for k, v in d["msm_boolean_heavy"]["mixes"].items(): print(k, v["ms_median"], v["x_uniform_rate"], v["device_ms"], v["merge_reduce_share_of_step"])
print("fft", {k: d["fft"][k]["ms"] for k in ("fft", "ifft", "coset_fft", "icoset_fft")}, d["fft"]["roofline"]["alu"]["frac"], d["fft"]["roofline"]["hbm"]["frac"])
cp = d["create_proof"]; print("C4", cp["proofs_per_s"], cp["ms_total"], cp["ms_host_synthesis"], cp["with_r1cs_resident_in_hbm"]["ms_total"], cp["roofline"]["frac"], cp["roofline"]["device_ms"])
print("bool proof", json.dumps(d["create_proof_boolean"])[:900])
print("scaling", json.dumps(d["scaling_model"])[:1500])
print("others", [(s["group"], s["log_n"], s["ms_median"]) for s in d["msm_other_shapes"]], d["create_proof_mimc"]["ms_median"])
PY
timeout 900 python -m pytest tests/test_gpu_bench_smoke.py -m gpu -x -q 2>&1 | tail -5
