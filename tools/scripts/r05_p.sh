#!/bin/bash
# slab reduce with 16-byte loads over 32 slab groups: tests, isolated timing, the step, the exact three-tower loop
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
O=gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
timeout -k 10 600 python -m pytest tests/test_10_kernels_gpu.py tests/test_45_fp8_gpu.py tests/test_50_fullft_gpu.py -x -q -k "lora or shared or fp8" > $O/r05_p_tests.log 2>&1; rc=$?; tail -3 $O/r05_p_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/lora_grad_bench.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/attn_bench.py 2>&1 | grep -E "LoRA partials"
for m in "" "--no-text" "" "--no-text"; do echo "== bench.py $m"; python bench.py $m --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "gpu:|rror" | tail -2; done
STOP=1 ITERS=12 timeout -k 10 600 python tools/debug_graph_flake.py > $O/r05_p_flake.log 2>&1; grep -E "MISMATCH|Error|error" $O/r05_p_flake.log | cut -c1-200 | head -4; grep -c "equal;" $O/r05_p_flake.log
