#!/bin/bash
# round 5, GPU run C: exact-mode attention on split-bf16 operands -- kernel tests, timings, exact-mode encoder / trajectory tests, step time
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/bioscan-clip_amd
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_10_kernels_gpu.py -x -q -m gpu -k "exact" > $O/r05_c_tests.log 2>&1 || { tail -40 $O/r05_c_tests.log; exit 1; }
tail -3 $O/r05_c_tests.log
timeout -k 10 300 python tools/exact_attn_bench.py > $O/r05_c_exact_attn_bench.log 2>&1 || { tail $O/r05_c_exact_attn_bench.log; exit 1; }
cat $O/r05_c_exact_attn_bench.log
rm -f $O/parity.jsonl
timeout -k 10 900 python -m pytest tests/test_20_encoders_gpu.py tests/test_30_graph_gpu.py tests/test_40_dropout_gpu.py -x -q -m gpu -k "exact" > $O/r05_c_tests2.log 2>&1 || { tail -40 $O/r05_c_tests2.log; exit 1; }
tail -3 $O/r05_c_tests2.log
cp $O/parity.jsonl $O/r05_c_parity_exact.jsonl
BSCLIP_PARITY=2 timeout -k 10 600 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>$O/r05_c_bench_exact.log | tail -1 > $O/r05_c_bench_exact.json || { tail $O/r05_c_bench_exact.log; exit 1; }
python -c "
import json; d=json.load(open('$O/r05_c_bench_exact.json')); print('exact mode ms/step', d['ms_per_step'], d['config']['numerics'][:40])"
