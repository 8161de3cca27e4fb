#!/bin/bash
# Round-2 GPU call 21: kernel timeline of a tiny G2 multiexp (2^9) and of a MiMC-322 proof: launch gaps vs kernel time
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c21
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_g2_9 -o p -- python tools/profile_suite.py msm 2 9 5 > $OUT/trace_g2_9.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_mimc -o p -- python tools/profile_suite.py mimc 5 > $OUT/trace_mimc.log 2>&1
for d in trace_g2_9 trace_mimc; do f=$(ls $OUT/$d/*kernel_trace.csv | head -1); (head -1 $f; tail -600 $f) > $OUT/$d.csv; rm -rf $OUT/$d; done
tail -2 $OUT/trace_g2_9.log $OUT/trace_mimc.log
