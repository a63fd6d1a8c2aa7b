#!/bin/bash
# after the ds_add_f32 form of the cross-wave LDS sums: kernel tests, then the exact three-tower graph-vs-eager loop
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
timeout -k 10 900 python -m pytest tests/test_10_kernels_gpu.py tests/test_50_fullft_gpu.py -x -q -k "lora or test_50 or fullft or full or shared" > gpurun_out/r05_flake7_tests.log 2>&1; rc=$?; tail -4 gpurun_out/r05_flake7_tests.log
[ $rc -eq 0 ] || exit $rc
for v in 1 2 3 4; do
  echo "== run $v"; STOP=1 ITERS=12 timeout -k 10 600 python tools/debug_graph_flake.py > gpurun_out/r05_flake7.log 2>&1; grep -E "MISMATCH|Error|error" gpurun_out/r05_flake7.log | cut -c1-200 | head -4; grep -c "equal;" gpurun_out/r05_flake7.log
done
