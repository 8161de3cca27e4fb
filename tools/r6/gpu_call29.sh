#!/bin/bash
# Round 6, call 29: chunk length of the 20-bit G1 window table (13 rows, 2^19 buckets in one set) at 2^19 ... 2^22, and the
# kernel statistics of its reduce phase at 2^20.
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c29_table20_chunks.txt
: > $out
timeout 900 python tools/profile_suite.py tsweep 1 19 22 20 0,32,48,64,96,128,192 >> $out 2>&1
timeout 300 python tools/profile_suite.py tsweep 1 20 20 16 0,16,32,64,128 >> $out 2>&1
cd /tmp && export TMPDIR=/tmp
cat > /tmp/one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
sys.argv = ["profile_suite.py", "tsweep", "1", "20", "20", "20", os.environ.get("KK", "0")]
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import runpy
runpy.run_path(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools", "profile_suite.py"), run_name="__main__")
PY
for kk in 0 96; do
  rm -rf /tmp/prof29
  KK=$kk timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof29 -o t20 -- python /tmp/one.py > /tmp/prof29.log 2>&1
  f=$(find /tmp/prof29 -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$GRAFT_REPO_ROOT/gpurun_out/r6c29_table20_K${kk}_kernel_stats.csv"
  tail -2 /tmp/prof29.log >> "$GRAFT_REPO_ROOT/$out"
done
