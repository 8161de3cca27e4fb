#!/bin/bash
# round 6, call 7: merge_runs / merge_long on lane pairs (K2) against one lane per point, same box, alternating processes
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6c7
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for ln in 14 15 16 17 18; do
  for k2 in 1 0; do
    echo "## K2=$k2 2^$ln"
    BELLMAN_HIP_LONG_K2=$k2 MIX=uniform timeout 120 python tools/r6/boolean_mix.py 1 $ln 1 0 15 | tail -1
    BELLMAN_HIP_LONG_K2=$k2 MIX=bool50 timeout 120 python tools/r6/boolean_mix.py 1 $ln 1 0 15 | tail -1
  done
done
done 2>&1 | tee $OUT/k2_ab.txt
for k2 in 1 0; do
  tag=uniform_g1_16_k2$k2
  (cd /tmp && BELLMAN_HIP_LONG_K2=$k2 MIX=uniform timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o $tag -- python $GRAFT_REPO_ROOT/tools/r6/boolean_mix.py 1 16 1 0 10 > $OUT/$tag.log 2>&1)
  python tools/kstats.py $OUT/$tag | grep -v "fixed_base\|window_table\|rocclr" | head -12
  find $OUT/$tag -name '*.db' -delete; find $OUT/$tag -name '*_trace.csv' -delete; find $OUT/$tag -name '*agent_info.csv' -delete
done 2>&1 | tee -a $OUT/k2_ab.txt
