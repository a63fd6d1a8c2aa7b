#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
for v in "BSCLIP_TOWER_STREAMS=0" "TEXT=0" "EXACT=0" "BSCLIP_EXACT_ATTN=2"; do
  echo "== $v"; env $v STOP=1 ITERS=12 timeout -k 10 600 python tools/debug_graph_flake.py > gpurun_out/r05_flake4.log 2>&1; grep -E "iteration|step [0-9]: .*differing: [1-9]|grad |Error|error" gpurun_out/r05_flake4.log | cut -c1-200 | head -24
done
