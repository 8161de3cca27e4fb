#!/bin/bash
# Round-2 GPU call 11: kernel timeline of the 2^20 proof (R1CS resident), with and without the high-priority
# stream for the reduction phase
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c11
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_prio0 -o p -- python tools/profile_suite.py proof 20 3 1 > $OUT/trace_prio0.log 2>&1
BELLMAN_HIP_REDUCE_PRIORITY=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_prio1 -o p -- python tools/profile_suite.py proof 20 3 1 > $OUT/trace_prio1.log 2>&1
grep create_proof $OUT/trace_prio0.log $OUT/trace_prio1.log
for i in 1 2 3; do python tools/profile_suite.py proof 20 7 1 | grep create_proof; BELLMAN_HIP_REDUCE_PRIORITY=1 python tools/profile_suite.py proof 20 7 1 | grep create_proof; done > $OUT/prio_ab.txt 2>&1
cat $OUT/prio_ab.txt
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
# keep only the last 4000 dispatches of each trace (the timed proofs)
for d in trace_prio0 trace_prio1; do f=$(ls $OUT/$d/*kernel_trace.csv | head -1); (head -1 $f; tail -4000 $f) > $OUT/$d.csv; rm -rf $OUT/$d; done
ls -la $OUT
