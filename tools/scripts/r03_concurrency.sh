# how much of a replayed step has 0 / 1 / 2 kernels in flight (rocprofv3 kernel trace of bench.py + tools/rocpd_concurrency.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_conc -o c -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_conc.log 2>&1
python3 $R/tools/rocpd_concurrency.py $(ls $R/gpurun_out/prof_conc/*.db $R/gpurun_out/prof_conc/*/*.db 2>/dev/null | head -1) 8
rm -rf $R/gpurun_out/prof_conc
