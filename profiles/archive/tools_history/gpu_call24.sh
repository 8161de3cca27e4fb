#!/bin/bash
# Round-2 GPU call 24: does GPU_MAX_HW_QUEUES take effect when torch initialises HIP first (bench.py's situation)?
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c24
mkdir -p $OUT
export TMPDIR=/tmp
echo "A: harness alone"; python tools/profile_suite.py mimc 40
echo "B: env set, torch.cuda.init() first"; python - <<'P'
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
torch.cuda.init(); x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
sys.argv = ["profile_suite.py", "mimc", "40"]
sys.path.insert(0, "tools")
import runpy
runpy.run_path("tools/profile_suite.py", run_name="__main__")
P
echo "C: torch first, env NOT set beforehand"; python - <<'P'
import os, sys
import torch
torch.cuda.init(); x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
sys.argv = ["profile_suite.py", "mimc", "40"]
import runpy
runpy.run_path("tools/profile_suite.py", run_name="__main__")
P
echo "D: env=4"; GPU_MAX_HW_QUEUES=4 python tools/profile_suite.py mimc 40
