#!/bin/bash
# round 6, call 5: per-kernel view of the boolean-heavy mixes after the piece-wise long-run merge
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r6c5}
mkdir -p $OUT
export TMPDIR=/tmp
for cfg in ${CFGS:-"bool50,1,20,0,0" "bool90,1,20,0,0" "small90,1,20,0,0" "bool50,1,16,1,0" "bool90,2,19,1,0" "bool50,1,22,1,0"}; do
  IFS=, read mix g ln tb dn <<< "$cfg"
  tag=${mix}_g${g}_${ln}_t${tb}
  (cd /tmp && MIX=$mix timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o $tag -- python $GRAFT_REPO_ROOT/tools/r6/boolean_mix.py $g $ln $tb $dn 5 > $OUT/$tag.log 2>&1)
  grep -A2 "^mix" $OUT/$tag.log | tail -1
  python tools/kstats.py $OUT/$tag | grep -v "fixed_base\|window_table\|rocclr" | head -12
  find $OUT/$tag -name '*.db' -delete; find $OUT/$tag -name '*_trace.csv' -delete; find $OUT/$tag -name '*agent_info.csv' -delete
done 2>&1 | tee $OUT/summary.txt
