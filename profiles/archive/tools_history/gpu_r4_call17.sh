#!/bin/bash
# round 4, call 17: G1 window tables up to 2^22 points as the DEFAULT (16-bit rows to 2^20, 20-bit rows above, records at a
# 128-byte stride): parity of the new path, registration cost, and the A/B that decides it for proofs (the table plan's
# accumulation is slower per addition; inside a proof the host tail it removes was hidden anyway)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c17; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_round4.py -q -x -k "128_byte or large_host" ) > $O/parity.txt 2>&1; tail -2 $O/parity.txt
( timeout 600 python -m pytest tests/test_gpu_scale.py -q -x -k "c5_scale and (1-19 or 1-21)" ) > $O/parity_scale.txt 2>&1; tail -2 $O/parity_scale.txt
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "msm" ) > $O/parity_msm.txt 2>&1; tail -2 $O/parity_msm.txt
python - > $O/registration.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, "tools")
import bellman_amd
from bellman_amd import _lib
from profile_suite import make_bases
lib = _lib.load()
w = bellman_amd.Worker(0)
for log_n in (18, 19, 20, 21, 22):
    n = 1 << log_n
    dout = make_bases(w, lib, 1, n)
    for rep in range(2):
        t0 = time.perf_counter()
        b = bellman_amd.Bases.copy_device(w, 1, dout, n)
        dt = (time.perf_counter() - t0) * 1e3
        info = b.table_info()
        b.release()
    print("G1 2^%d registration (copy + window table c=%d, %d rows, %.2f GB): %.1f ms" % (log_n, info[0], info[1], info[2] / 1e9, dt), flush=True)
    w.free(dout)
PY
cat $O/registration.txt | grep -v amdgpu
for i in 1 2; do
  BELLMAN_HIP_TABLE_MAX_LOG2_G1=18 python tools/profile_suite.py proof 20 7 12 2>&1 | grep create_proof | sed 's/^/classic G1 plan: /' >> $O/proof_ab.txt
  python tools/profile_suite.py proof 20 7 12 2>&1 | grep create_proof | sed 's/^/G1 tables:      /' >> $O/proof_ab.txt
done
cat $O/proof_ab.txt
BELLMAN_HIP_TABLE_MAX_LOG2_G1=18 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-proof > $O/bench_classic.json 2> $O/bench_classic.err
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-proof > $O/bench_tables.json 2> $O/bench_tables.err
python - <<'PY'
import json
for tag in ("classic", "tables"):
    try:
        d = json.loads([l for l in open("gpurun_out/r4c17/bench_%s.json" % tag) if l.startswith("{")][-1])
        print(tag, d["value"], d["ms_per_step"], d["config"]["device_ms"], d["config"].get("value_with_2_jobs_in_flight"), d.get("value_incl_scalar_upload"), d["roofline"]["alu"]["frac"])
    except Exception as e:
        print(tag, "no line", e)
PY
