cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE:mfma" "FETCH_SIZE:fetch" "WRITE_SIZE:write"; do
  ctr=${pass%%:*}; tag=${pass##*:}
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r2_$tag -o p -- python3 $R/tools/gemm_one.py fc1 fc1_dna tr_dna > $R/gpurun_out/pmc_r2_$tag.log 2>&1
done
ls $R/gpurun_out/pmc_r2_mfma | head
cd $R
for v in 0 1; do BSCLIP_FORCE_DIST=$v MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --no-graph --text --steps 20 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep "gpu:"; done
cd /tmp
BSCLIP_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29612 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r2d -o r2d -- python3 $R/bench.py --no-graph --text --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_r2d.log 2>&1
ls $R/gpurun_out/prof_r2d
