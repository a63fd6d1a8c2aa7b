#!/bin/bash
# round-5 closing run on the final tree: the whole GPU suite (parity distances -> parity.jsonl), the default bench line, the GEMM yardstick
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
O=gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
rm -f $O/parity.jsonl
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/r05_g_gpu_tests.log 2>&1; rc=$?; tail -5 $O/r05_g_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
cp $O/parity.jsonl $O/r05_g_parity.jsonl
python bench.py > $O/r05_g_bench_line.json 2> $O/r05_g_bench.err || { tail -20 $O/r05_g_bench.err; exit 1; }
grep -E "gpu:|busiest" $O/r05_g_bench.err; cut -c1-300 $O/r05_g_bench_line.json
TILES=0 python tools/gemm_bench.py > $O/r05_g_gemm_yardstick.txt 2>&1; tail -18 $O/r05_g_gemm_yardstick.txt
