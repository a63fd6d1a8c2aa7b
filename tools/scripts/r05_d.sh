#!/bin/bash
# round 5, GPU run D: exact mode with LayerNorm-fused split operands and the wave-per-row f32 LoRA gradients -- every exact-mode test, LN kernel tests, step time + kernel stats
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/bioscan-clip_amd
O=gpurun_out
rm -f $O/parity.jsonl
timeout -k 10 900 python -m pytest tests/test_10_kernels_gpu.py tests/test_20_encoders_gpu.py tests/test_30_graph_gpu.py tests/test_40_dropout_gpu.py tests/test_90_dist_gpu.py -x -q -m gpu -k "exact or layernorm or lora_grad" > $O/r05_d_tests.log 2>&1 || { tail -40 $O/r05_d_tests.log; exit 1; }
tail -3 $O/r05_d_tests.log
cp $O/parity.jsonl $O/r05_d_parity_exact.jsonl
BSCLIP_PARITY=2 timeout -k 10 600 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>$O/r05_d_bench_exact.log | tail -1 > $O/r05_d_bench_exact.json || { tail $O/r05_d_bench_exact.log; exit 1; }
python -c "
import json; d=json.load(open('$O/r05_d_bench_exact.json')); print('exact mode ms/step', d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
BSCLIP_PARITY=2 timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/r05_d_prof -o x -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline --no-graph > $GRAFT_REPO_ROOT/$O/r05_d_prof.log 2>&1 || { tail $GRAFT_REPO_ROOT/$O/r05_d_prof.log; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(ls $O/r05_d_prof/*kernel_stats.csv $O/r05_d_prof/*/*kernel_stats.csv 2>/dev/null | head -1)
cp $f $O/r05_d_exact_mode_kernel_stats.csv; rm -rf $O/r05_d_prof
head -16 $O/r05_d_exact_mode_kernel_stats.csv | cut -c1-150
