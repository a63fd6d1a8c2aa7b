#!/bin/bash
# round 5: kernel statistics of the exact mode (BSCLIP_PARITY=2) at the headline shape
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export BSCLIP_PARITY=2
timeout -k 10 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r5x -o x -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_r5x.log 2>&1 || { tail $R/gpurun_out/prof_r5x.log; exit 1; }
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/prof_r5x/*.db $R/gpurun_out/prof_r5x/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r05_e_exact_mode_kernel_stats.csv > /dev/null; rm -rf $R/gpurun_out/prof_r5x
head -30 $R/gpurun_out/r05_e_exact_mode_kernel_stats.csv | cut -c1-160
