#!/bin/bash
# confidence run: graph replay against eager, three towers, 40 iterations each in the default and in the exact mode
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
for v in "EXACT=0" "EXACT=1"; do
  echo "== $v"; env $v STOP=1 ITERS=40 timeout -k 10 900 python tools/debug_graph_flake.py > gpurun_out/r05_m_$v.log 2>&1; grep -E "MISMATCH|Error|error" gpurun_out/r05_m_$v.log | cut -c1-200 | head -4; grep -c "equal;" gpurun_out/r05_m_$v.log
done
