#!/bin/bash
# persistent GEMM on shorter row panels (224 / 192) -- needs tools/experiments/gemm_row_panels.patch applied (the experiment was measured and
# not adopted: profiles/r05_i_*): tests, per-shape timings (tile 8 = auto height, 9 / 10 forced), the step with and without
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
O=gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
timeout -k 10 900 python -m pytest tests/test_10_kernels_gpu.py tests/test_40_dropout_gpu.py -x -q -k "gemm" > $O/r05_i_gemm_tests.log 2>&1; rc=$?; tail -8 $O/r05_i_gemm_tests.log
[ $rc -eq 0 ] || exit $rc
BSCLIP_GEMM_RT1=4 TILES=8,9,10 timeout -k 10 300 python tools/gemm_bench.py > $O/r05_i_gemm_panels.txt 2>&1; tail -18 $O/r05_i_gemm_panels.txt
TILES=0 timeout -k 10 300 python tools/gemm_bench.py > $O/r05_i_gemm_auto.txt 2>&1; tail -3 $O/r05_i_gemm_auto.txt
for e in 0 4 0 4; do for m in "" "--no-text"; do echo "== BSCLIP_GEMM_RT1=$e bench.py $m"; BSCLIP_GEMM_RT1=$e python bench.py $m --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "gpu:|rror" | tail -2; done; done
