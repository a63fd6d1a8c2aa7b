#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
timeout -k 10 600 python -m pytest tests/test_10_kernels_gpu.py -x -q -k "lora_grad_from or attention" > gpurun_out/r05_q_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r05_q_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/attn_bench.py 2>&1 | grep -E "LoRA partials|bwd "
for m in "" "--no-text" "" "--no-text"; do echo "== bench.py $m"; python bench.py $m --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "gpu:|rror" | tail -2; done
