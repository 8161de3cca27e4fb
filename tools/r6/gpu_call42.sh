#!/bin/bash
# Round 6, call 42: stage one of the G2 two-stage sums on lane triples (16 per wavefront) against lane sextets (8), piece lengths
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c42_g2_stage1.txt
: > $out
run() { echo "## $1" >> $out; shift; env "$@" timeout 300 python tools/profile_suite.py tsweep 2 19 20 20 >> $out 2>&1; }
run "triples, len by rule" A=1
run "sextets" BELLMAN_HIP_SUM_G2_STAGE1_K6=1
run "triples, len 16" BELLMAN_HIP_SUM_TWO_LEN=16
run "triples, len 64" BELLMAN_HIP_SUM_TWO_LEN=64
run "one stage" BELLMAN_HIP_SUM_TWO_STAGE=0
export TMPDIR=/tmp
rm -rf /tmp/prof42
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof42 -o p -- python tools/profile_suite.py tsweep 2 20 20 20 > /tmp/prof42.log 2>&1
f=$(find /tmp/prof42 -name '*kernel_trace.csv' | head -1)
python tools/r6/trace_last_job.py $f | grep -v "scan_\|sort_" >> $out 2>&1
