# kernel statistics of the exact mode (BSCLIP_PARITY=2) at the headline shape: where its 240 ms go
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export BSCLIP_PARITY=2
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r4x -o x -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_r4x.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/prof_r4x/*.db $R/gpurun_out/prof_r4x/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r04_k_exact_mode_kernel_stats.csv > /dev/null; rm -rf $R/gpurun_out/prof_r4x
head -30 $R/gpurun_out/r04_k_exact_mode_kernel_stats.csv | cut -c1-160; tail -2 $R/gpurun_out/prof_r4x.log | cut -c1-300
