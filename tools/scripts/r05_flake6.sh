#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
for v in "GUARD=2" "GUARD=2"; do
  echo "== $v"; env $v STOP=1 ITERS=10 timeout -k 10 600 python tools/debug_graph_flake.py > gpurun_out/r05_flake6.log 2>&1; grep -E "MISMATCH|workspace|Error|error|Traceback" gpurun_out/r05_flake6.log | cut -c1-260 | head -60; grep -c "equal;" gpurun_out/r05_flake6.log
done
