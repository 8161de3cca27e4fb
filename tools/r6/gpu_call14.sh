#!/bin/bash
# round 6, call 14: where a MiMC-322 proof and the mid-size G2 / G1 jobs spend their time now
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6c14
mkdir -p $OUT
export TMPDIR=/tmp
for wl in "mimc 10" "msm 2 10 10" "msm 2 14 10" "msm 1 10 10" "msm 1 18 10"; do
  tag=$(echo $wl | tr ' ' '_')
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/tools/profile_suite.py $wl > $OUT/$tag.log 2>&1)
  grep -v "simple_timer\|rocprofv3\|Opened" $OUT/$tag.log | tail -2
  python tools/kstats.py $OUT/$tag | grep -v "fixed_base\|window_table\|rocclr\|gen_\|decode\|check" | head -14
  find $OUT/$tag -name '*.db' -delete; find $OUT/$tag -name '*_trace.csv' -delete; find $OUT/$tag -name '*agent_info.csv' -delete
done 2>&1 | tee $OUT/summary.txt
