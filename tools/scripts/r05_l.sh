#!/bin/bash
# the splatted LoRA-gradient kernels: speed (lora_grad alone, attention + lora_grad_heads), the exact three-tower loop, the step
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
O=gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
timeout -k 10 200 python tools/lora_grad_bench.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/attn_bench.py 2>&1 | grep -E "LoRA partials"
for v in 1 2 3; do
  echo "== exact three-tower loop, run $v"; STOP=1 ITERS=12 timeout -k 10 600 python tools/debug_graph_flake.py > $O/r05_l_flake.log 2>&1; grep -E "MISMATCH|Error|error" $O/r05_l_flake.log | cut -c1-200 | head -4; grep -c "equal;" $O/r05_l_flake.log
done
for m in "" "--no-text" "" "--no-text"; do echo "== bench.py $m"; python bench.py $m --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "gpu:|rror" | tail -2; done
