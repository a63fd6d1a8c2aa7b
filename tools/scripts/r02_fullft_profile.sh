# kernel statistics of the full fine-tuning step (bench.py --full-ft), towers serialised so that durations are per kernel
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
BSCLIP_TOWER_STREAMS=0 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_ft -o ft -- python3 $R/bench.py --full-ft --no-graph --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_ft.log 2>&1
cd $R
DB=$(ls gpurun_out/prof_ft/*/*.db gpurun_out/prof_ft/*.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $DB gpurun_out/fullft_serial_kernel_stats.csv > /dev/null
head -40 gpurun_out/fullft_serial_kernel_stats.csv
tail -2 gpurun_out/fullft_serial_kernel_stats.csv
