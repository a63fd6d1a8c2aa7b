#!/bin/bash
# LayerNorm forward with the LoRA projection on two rows per iteration (experiment, reverted: profiles/r05_k_ln_bench.log): tests, isolated timing, the step
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
O=gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
timeout -k 10 600 python -m pytest tests/test_10_kernels_gpu.py -x -q -k "layernorm or norm" > $O/r05_k_ln_tests.log 2>&1; rc=$?; tail -5 $O/r05_k_ln_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/ln_bench.py > $O/r05_k_ln_bench.log 2>&1; cat $O/r05_k_ln_bench.log
M=34048 timeout -k 10 200 python tools/ln_bench.py >> $O/r05_k_ln_bench.log 2>&1; tail -4 $O/r05_k_ln_bench.log
timeout -k 10 900 python -m pytest tests/test_20_encoders_gpu.py -x -q > $O/r05_k_encoder_tests.log 2>&1; rc=$?; tail -4 $O/r05_k_encoder_tests.log
[ $rc -eq 0 ] || exit $rc
for m in "" "--no-text" "" "--no-text"; do echo "== bench.py $m"; python bench.py $m --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "gpu:|rror" | tail -2; done
