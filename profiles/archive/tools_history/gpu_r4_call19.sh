#!/bin/bash
# round 4, call 19: the 2-row partial G1 table at 2^20 dense (192 MB, inside the 256 MB Infinity Cache) or at the 128-byte stride
# (256 MB) against no table: bench.py and the 2^20 proof with twelve threads, alternating
# (measures the removed partial tables: tools/experiments/r4_partial_g1_tables.patch)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c19; mkdir -p $O
for rep in 1 2; do
for cfg in "0 1" "2 1" "2 0"; do
  set -- $cfg
  BELLMAN_HIP_TABLE_ROWS=$1 BELLMAN_HIP_TABLE_PAD=$2 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-proof > $O/bench.json 2> $O/bench.err
  python - "$1" "$2" >> $O/ab.txt <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/r4c19/bench.json") if l.startswith("{")][-1])
print("rows=%s pad=%s bench" % (sys.argv[1], sys.argv[2]), d["value"], d["ms_per_step"], d["config"]["device_ms"], "2 jobs", d["config"].get("value_with_2_jobs_in_flight"), "incl upload", d.get("value_incl_scalar_upload"))
PY
  BELLMAN_HIP_TABLE_ROWS=$1 BELLMAN_HIP_TABLE_PAD=$2 python tools/profile_suite.py proof 20 7 12 2>&1 | grep create_proof | sed "s/^/rows=$1 pad=$2 /" >> $O/ab.txt
done
done
cat $O/ab.txt
