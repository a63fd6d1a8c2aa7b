#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
for v in "X=0" "X=0" "GUARD=1" "GUARD=1" "BSCLIP_TOWER_SERIAL=2" "BSCLIP_TOWER_SERIAL=2" "BSCLIP_TOWER_SERIAL=1" "BSCLIP_TOWER_SERIAL=1" "BSCLIP_TOWER_SERIAL=0" "BSCLIP_TOWER_SERIAL=0"; do
  echo "== $v"; env $v STOP=1 ITERS=10 timeout -k 10 600 python tools/debug_graph_flake.py > gpurun_out/r05_flake5.log 2>&1; grep -E "MISMATCH|GUARD|Error|error" gpurun_out/r05_flake5.log | cut -c1-200 | head -6; grep -c "equal;" gpurun_out/r05_flake5.log
done
