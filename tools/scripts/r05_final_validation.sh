#!/bin/bash
# round-5 validation run on the GPU box: smoke, the bench line (I+D+T headline; configs[1] / fp8 / parity / exact as extras, CPU baseline),
# other shapes and switches, world_size 1 through RCCL on the per-tower-graph path with both collective back ends, kernel stats of the
# replayed and the serialised run, fresh attention PMC passes, the host-thread experiment
T=${T:-f}   # tag of the output files (f: the first closing run of round 5; z: the run on the final tree)
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
O=gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_${T}_smoke.log 2>&1; tail -2 $O/r05_${T}_smoke.log
python bench.py > $O/r05_${T}_bench_line.json 2> $O/r05_${T}_bench.err || { tail -20 $O/r05_${T}_bench.err; exit 1; }
grep -E "gpu:|busiest" $O/r05_${T}_bench.err; cut -c1-400 $O/r05_${T}_bench_line.json
for m in "--no-text" "--no-text --batch 8" "--no-text --batch 64" "--no-text --batch 1024" "--no-graph" "--no-text --full-ft"; do echo "== bench.py $m"; python bench.py $m --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "gpu:|rror" | tail -2; done
for e in "BSCLIP_PARITY=1" "BSCLIP_PARITY=2" "BSCLIP_GEMM_PERSISTENT=0" "BSCLIP_ATTN_KEEP_BITS=0" "HSA_ENABLE_INTERRUPT=1" "HSA_ENABLE_INTERRUPT=0"; do echo "== $e bench.py --no-text"; env $e python bench.py --no-text --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "gpu:|rror|busiest" | tail -2; done
for nc in 0 1; do echo "== BSCLIP_FORCE_DIST=1 BSCLIP_NATIVE_COMM=$nc (world_size 1 through RCCL, per-tower graphs)"; BSCLIP_NATIVE_COMM=$nc BSCLIP_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2964$nc RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>$O/r05_${T}_dist_$nc.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['collectives'][:50], d.get('dist'))"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/$O/prof_r5h -o g -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/$O/prof_r5h.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/$O/prof_r5h/*.db $R/$O/prof_r5h/*/*.db 2>/dev/null | head -1) $R/$O/r05_${T}_bench_idt_b256_kernel_stats.csv > /dev/null; rm -rf $R/$O/prof_r5h
head -12 $R/$O/r05_${T}_bench_idt_b256_kernel_stats.csv | cut -c1-150
rocprofv3 --kernel-trace -d $R/$O/prof_r5k -o h -- python3 $R/bench.py --no-text --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/$O/prof_r5k.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/$O/prof_r5k/*.db $R/$O/prof_r5k/*/*.db 2>/dev/null | head -1) $R/$O/r05_${T}_bench_b256_kernel_stats.csv > /dev/null; rm -rf $R/$O/prof_r5k
BSCLIP_TOWER_STREAMS=0 rocprofv3 --kernel-trace -d $R/$O/prof_r5l -o i -- python3 $R/bench.py --no-text --no-graph --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/$O/prof_r5l.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/$O/prof_r5l/*.db $R/$O/prof_r5l/*/*.db 2>/dev/null | head -1) $R/$O/r05_${T}_serial_kernel_stats.csv > /dev/null; rm -rf $R/$O/prof_r5l
head -24 $R/$O/r05_${T}_serial_kernel_stats.csv | cut -c1-150
BSCLIP_PARITY=2 rocprofv3 --kernel-trace -d $R/$O/prof_r5x -o x -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $R/$O/prof_r5x.log 2>&1
python3 $R/tools/rocpd_stats.py $(ls $R/$O/prof_r5x/*.db $R/$O/prof_r5x/*/*.db 2>/dev/null | head -1) $R/$O/r05_${T}_exact_mode_kernel_stats.csv > /dev/null; rm -rf $R/$O/prof_r5x
cd $R
bash tools/scripts/r05_attn_pmc.sh > $O/r05_${T}_attn_pmc_raw.txt 2>&1; tail -30 $O/r05_${T}_attn_pmc_raw.txt
