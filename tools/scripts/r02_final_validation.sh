# round-2 validation run on the GPU box: full -m gpu suite, smoke, bench lines (bf16 / I+D+T / fp8 / other batch sizes), kernel stats
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -q > gpurun_out/gputest_r2h.log 2>&1; tail -4 gpurun_out/gputest_r2h.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r2h.log 2>&1; tail -1 gpurun_out/smoke_r2h.log
python bench.py > gpurun_out/bench_r2h.json 2> gpurun_out/bench_r2h.err; grep "gpu:" gpurun_out/bench_r2h.err; cat gpurun_out/bench_r2h.json
for m in "--text" "--fp8" "--fp8 --batch 512" "--batch 8" "--batch 64" "--batch 1024" "--no-graph" "--no-graph --batch 8" "--full-ft"; do echo "== bench.py $m"; python bench.py $m --steps 20 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep -E "gpu:|rror" | tail -2; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r2h -o r2h -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_r2h.log 2>&1
BSCLIP_TOWER_STREAMS=0 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r2i -o r2i -- python3 $R/bench.py --no-graph --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_r2i.log 2>&1
ls $R/gpurun_out/prof_r2h $R/gpurun_out/prof_r2i
