#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
STOP=0 ITERS=12 timeout -k 10 900 python tools/debug_graph_flake.py > gpurun_out/r05_flake3.log 2>&1; grep -E "iteration|step 7|step 6|grad|param" gpurun_out/r05_flake3.log | cut -c1-230 | head -70
echo "== HIP_LAUNCH_BLOCKING=1"
HIP_LAUNCH_BLOCKING=1 STOP=0 ITERS=12 timeout -k 10 900 python tools/debug_graph_flake.py > gpurun_out/r05_flake3b.log 2>&1; grep -E "iteration" gpurun_out/r05_flake3b.log | cut -c1-200 | head -20
