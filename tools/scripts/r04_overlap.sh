# round 4: the W > 1 launch path (per-tower captured graphs, each tower's all-gather / all-reduce issued from its stream) through
# the REAL process group at world_size 1: step time with and without the collectives, and a kernel + memory-copy trace in which
# the all-gathers (device copies at world_size 1) can be placed against the other towers' kernels (tools/rocpd_overlap.py)
R=$GRAFT_REPO_ROOT
cd $R
E="MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0"
echo "== plain (one captured graph), I+D+T B=256"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "gpu:|rror" | tail -1
echo "== BSCLIP_FORCE_DIST=1 (RCCL at world_size 1, per-tower graphs)"; env $E MASTER_PORT=29651 BSCLIP_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "gpu:|rror" | tail -1
cd /tmp && export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_PORT=29652 BSCLIP_FORCE_DIST=1
rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/prof_r4_overlap -o ov -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_r4_overlap.log 2>&1
db=$(ls $R/gpurun_out/prof_r4_overlap/*.db $R/gpurun_out/prof_r4_overlap/*/*.db 2>/dev/null | head -1)
echo "== trace: $db"
[ -n "$db" ] && python3 $R/tools/rocpd_overlap.py $db $R/gpurun_out/r04_overlap.txt | tail -30
[ -n "$db" ] && python3 $R/tools/rocpd_concurrency.py $db 2>&1 | tail -12
rm -rf $R/gpurun_out/prof_r4_overlap
