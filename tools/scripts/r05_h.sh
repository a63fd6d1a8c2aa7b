#!/bin/bash
# LoRA partial sums out of the attention backward: kernel tests, encoder parity, isolated timings, the step with and without
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
O=gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
timeout -k 10 600 python -m pytest tests/test_10_kernels_gpu.py -x -q -k "lora_grad or attention" > $O/r05_h_kernel_tests.log 2>&1; rc=$?; tail -15 $O/r05_h_kernel_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/attn_bench.py > $O/r05_h_attn_bench.log 2>&1; cat $O/r05_h_attn_bench.log
rm -f $O/parity.jsonl
timeout -k 10 900 python -m pytest tests/test_20_encoders_gpu.py tests/test_40_dropout_gpu.py -x -q > $O/r05_h_encoder_tests.log 2>&1; rc=$?; tail -8 $O/r05_h_encoder_tests.log
[ $rc -eq 0 ] || exit $rc
cp $O/parity.jsonl $O/r05_h_parity.jsonl
for e in 1 0 1 0; do for m in "" "--no-text"; do echo "== BSCLIP_ATTN_LORA=$e bench.py $m"; BSCLIP_ATTN_LORA=$e python bench.py $m --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "gpu:|rror" | tail -2; done; done
