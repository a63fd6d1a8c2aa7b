#!/bin/bash
# dt partial sums as [heads][q | v][tokens][4] (whole lines per wave): tests, isolated timing, the step
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
O=gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
timeout -k 10 600 python -m pytest tests/test_10_kernels_gpu.py tests/test_50_fullft_gpu.py -x -q -k "lora_grad or shared" > $O/r05_n_tests.log 2>&1; rc=$?; tail -3 $O/r05_n_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/attn_bench.py 2>&1 | grep -E "LoRA partials"
for m in "" "--no-text" "" "--no-text"; do echo "== bench.py $m"; python bench.py $m --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "gpu:|rror" | tail -2; done
cd /tmp && export TMPDIR=/tmp
BSCLIP_TOWER_STREAMS=0 rocprofv3 --kernel-trace -d $R/$O/prof_r5n -o i -- python3 $R/bench.py --no-text --no-graph --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/$O/prof_r5n.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/$O/prof_r5n/*.db $R/$O/prof_r5n/*/*.db 2>/dev/null | head -1) $R/$O/r05_n_serial_kernel_stats.csv > /dev/null; rm -rf $R/$O/prof_r5n
grep -E "attn_bwd|lora_" $R/$O/r05_n_serial_kernel_stats.csv | cut -c1-120
