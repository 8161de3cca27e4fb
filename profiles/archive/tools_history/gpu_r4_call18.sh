#!/bin/bash
# round 4, call 18: PARTIAL window tables for the G1 vectors that run the classic 16-bit plan (2^18 < n < 2^24): R rows (P, 2^128 P
# for R = 2) at the 128-byte stride, digit columns w and w + 16/R sharing a bucket set - parity first, then R = 0 (off) / 2 / 4 in
# separate processes: per-size MSM, a 2^20 proof, bench.py
# (the code these runs measured was removed afterwards; tools/experiments/r4_partial_g1_tables.patch restores it)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c18; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py -q -x -k "msm or 128_byte or large_host" ) > $O/parity.txt 2>&1; tail -2 $O/parity.txt
( timeout 600 python -m pytest tests/test_gpu_scale.py tests/test_gpu_groth16.py -q -x -k "(c5_scale and 1-23) or 2_23_density or chain_2_20" ) > $O/parity_scale.txt 2>&1; tail -2 $O/parity_scale.txt
( BELLMAN_HIP_TABLE_ROWS=4 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py -q -x -k "msm_2_20 or fuzz or large_host" ) > $O/parity_rows4.txt 2>&1; tail -2 $O/parity_rows4.txt
for r in 0 2 4 0 2; do
  BELLMAN_HIP_TABLE_ROWS=$r python tools/profile_suite.py sizes 1 19 22 2>&1 | grep "^G1" | sed "s/^/rows=$r /" >> $O/sizes.txt
done
cat $O/sizes.txt
for r in 0 2 4 0 2; do
  BELLMAN_HIP_TABLE_ROWS=$r python tools/profile_suite.py proof 20 7 12 2>&1 | grep create_proof | sed "s/^/rows=$r /" >> $O/proof_ab.txt
done
cat $O/proof_ab.txt
for r in 0 2 4; do
  BELLMAN_HIP_TABLE_ROWS=$r python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-proof > $O/bench_rows$r.json 2> $O/bench_rows$r.err
done
python - <<'PY'
import json
for tag in ("rows0", "rows2", "rows4"):
    try:
        d = json.loads([l for l in open("gpurun_out/r4c18/bench_%s.json" % tag) if l.startswith("{")][-1])
        print(tag, d["value"], d["ms_per_step"], d["config"]["device_ms"], d["config"].get("value_with_2_jobs_in_flight"), d.get("value_incl_scalar_upload"), d["roofline"]["alu"]["frac"])
    except Exception as e:
        print(tag, "no line", e)
PY
