# round-3 profile run on the GPU box: serial (towers on one stream, eager) kernel stats + the two-stream graph run
R=$GRAFT_REPO_ROOT
TAG=${1:-r03a}
cd /tmp && export TMPDIR=/tmp
BSCLIP_TOWER_STREAMS=0 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_${TAG}_serial -o s -- python3 $R/bench.py --no-graph --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_${TAG}_serial.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/prof_${TAG}_serial/*.db $R/gpurun_out/prof_${TAG}_serial/*/*.db 2>/dev/null | head -1) $R/gpurun_out/${TAG}_serial_kernel_stats.csv > /dev/null
head -30 $R/gpurun_out/${TAG}_serial_kernel_stats.csv; tail -1 $R/gpurun_out/${TAG}_serial_kernel_stats.csv
rm -rf $R/gpurun_out/prof_${TAG}_serial
