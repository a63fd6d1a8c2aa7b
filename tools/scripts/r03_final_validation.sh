# round-3 validation run on the GPU box: smoke, the bench line (with its side measurements and CPU baseline), other shapes, kernel stats
R=$GRAFT_REPO_ROOT
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r3.log 2>&1; tail -1 gpurun_out/smoke_r3.log
python bench.py > gpurun_out/bench_r3.json 2> gpurun_out/bench_r3.err; grep "gpu:" gpurun_out/bench_r3.err; cat gpurun_out/bench_r3.json
for m in "--batch 8" "--batch 64" "--batch 1024" "--no-graph" "--no-graph --batch 8" "--full-ft"; do echo "== bench.py $m"; python bench.py $m --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "gpu:|rror" | tail -2; done
for e in "BSCLIP_RESID_STREAM=f32 BSCLIP_GRAD_STREAM=f32 BSCLIP_PATCH_SPLIT=0" "BSCLIP_RESID_STREAM=f32" "BSCLIP_GRAD_STREAM=f32" "BSCLIP_GEMM_PERSISTENT=0" "BSCLIP_GEMM_NT=0" "BSCLIP_GEMM_PERSISTENT=0 BSCLIP_GEMM_NT=0" "BSCLIP_GEMM_GW=12"; do echo "== $e bench.py"; env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "gpu:|rror" | tail -1; done
echo "== BSCLIP_FORCE_DIST=1 (world_size 1 through RCCL, three captured graphs)"; BSCLIP_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29641 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --text --steps 20 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep -E "gpu:|rror" | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r3k -o h -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_r3k.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/prof_r3k/*.db $R/gpurun_out/prof_r3k/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r03_k_bench_b256_kernel_stats.csv > /dev/null; rm -rf $R/gpurun_out/prof_r3k
BSCLIP_TOWER_STREAMS=0 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r3l -o i -- python3 $R/bench.py --no-graph --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_r3l.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/prof_r3l/*.db $R/gpurun_out/prof_r3l/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r03_k_serial_kernel_stats.csv > /dev/null; rm -rf $R/gpurun_out/prof_r3l
head -24 $R/gpurun_out/r03_k_serial_kernel_stats.csv; tail -1 $R/gpurun_out/r03_k_serial_kernel_stats.csv; tail -1 $R/gpurun_out/r03_k_bench_b256_kernel_stats.csv
