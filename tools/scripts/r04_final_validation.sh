# round-4 validation run on the GPU box: smoke, the bench line (I+D+T headline, configs[1] / fp8 / parity as extras, CPU baseline),
# other shapes and switches, world_size 1 through RCCL on the per-tower-graph path, kernel stats of the replayed and the serialised run
R=$GRAFT_REPO_ROOT
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r4n.log 2>&1; tail -1 gpurun_out/smoke_r4n.log
python bench.py > gpurun_out/bench_r4n.json 2> gpurun_out/bench_r4n.err; grep -E "gpu:|busiest" gpurun_out/bench_r4n.err; cat gpurun_out/bench_r4n.json
for m in "--no-text" "--no-text --batch 8" "--no-text --batch 64" "--no-text --batch 1024" "--no-graph" "--no-text --full-ft"; do echo "== bench.py $m"; python bench.py $m --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "gpu:|rror" | tail -2; done
for e in "BSCLIP_PARITY=1" "BSCLIP_PARITY=2" "BSCLIP_GEMM_PERSISTENT=0"; do echo "== $e bench.py --no-text"; env $e python bench.py --no-text --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "gpu:|rror" | tail -1; done
echo "== BSCLIP_FORCE_DIST=1 (world_size 1 through RCCL, per-tower graphs)"; BSCLIP_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29641 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "gpu:|rror" | tail -1
cd /tmp && export TMPDIR=/tmp
# the headline command itself (I+D+T): kernel statistics of the replayed step
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r4h -o g -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_r4h.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/prof_r4h/*.db $R/gpurun_out/prof_r4h/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r04_k_bench_idt_b256_kernel_stats.csv > /dev/null; rm -rf $R/gpurun_out/prof_r4h
head -12 $R/gpurun_out/r04_k_bench_idt_b256_kernel_stats.csv; tail -1 $R/gpurun_out/r04_k_bench_idt_b256_kernel_stats.csv
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r4k -o h -- python3 $R/bench.py --no-text --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_r4k.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/prof_r4k/*.db $R/gpurun_out/prof_r4k/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r04_e_bench_b256_kernel_stats.csv > /dev/null; rm -rf $R/gpurun_out/prof_r4k
BSCLIP_TOWER_STREAMS=0 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r4l -o i -- python3 $R/bench.py --no-text --no-graph --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_r4l.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/gpurun_out/prof_r4l/*.db $R/gpurun_out/prof_r4l/*/*.db 2>/dev/null | head -1) $R/gpurun_out/r04_e_serial_kernel_stats.csv > /dev/null; rm -rf $R/gpurun_out/prof_r4l
head -22 $R/gpurun_out/r04_e_serial_kernel_stats.csv; tail -1 $R/gpurun_out/r04_e_serial_kernel_stats.csv; tail -1 $R/gpurun_out/r04_e_bench_b256_kernel_stats.csv
