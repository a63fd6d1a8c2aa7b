#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
for l in "lora512" "lora768" "lora512,lora768" "lora512,lora768,gemm,attn,ln"; do
  LOAD=$l N=2000 timeout -k 10 400 python tools/stress_lora_f32.py 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-220
done
