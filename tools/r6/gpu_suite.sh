#!/bin/bash
# the whole -m gpu suite with durations (what the driver runs at round end; its limit is 1200 s)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6suite
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r6suite/gputests.txt 2>&1
echo "rc=$? wall=$(( $(date +%s) - t0 )) s"
tail -40 gpurun_out/r6suite/gputests.txt
