#!/bin/bash
# round 6, call 2: per-kernel view of the boolean-heavy mixes (where the reduce phase goes)
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6c2
mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "bool50 1 20 0 0" "ones 1 20 1 0" "bool50 1 16 1 0" "bool90 2 19 1 0" "uniform 2 19 1 0"; do
  set -- $cfg
  tag=$1_g$2_$3_t$4
  (cd /tmp && MIX=$1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o $tag -- python $GRAFT_REPO_ROOT/tools/r6/boolean_mix.py $2 $3 $4 $5 5 > $OUT/$tag.log 2>&1)
  tail -3 $OUT/$tag.log
  python tools/kstats.py $OUT/$tag | grep -v fixed_base | head -14
  find $OUT/$tag -name '*.db' -delete; find $OUT/$tag -name '*_trace.csv' -delete
done
