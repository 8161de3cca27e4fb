#!/bin/bash
# Round 6, call 57: soaks on the final code - table churn, the mixed soak of tools/soak.py, boolean-heavy multiexps from four host threads
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c57_soaks.txt
: > $out
timeout 600 python tools/r6/soak_tables.py 24 >> $out 2>&1
timeout 900 python tools/soak.py >> $out 2>&1
timeout 400 python tools/r6/soak_long_runs.py 120 4 >> $out 2>&1
