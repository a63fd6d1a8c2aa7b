#!/bin/bash
# idle time inside the replayed step (I+D+T and I+D): kernel trace -> tools/rocpd_gaps.py
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
O=gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/$O/prof_r5j -o g -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/$O/prof_r5j.log 2>&1
DB=$(ls $R/$O/prof_r5j/*.db $R/$O/prof_r5j/*/*.db 2>/dev/null | head -1)
python3 $R/tools/rocpd_gaps.py $DB 0.5 40 loss_prep_kernel > $R/$O/r05_j_gaps_idt.txt; python3 $R/tools/rocpd_stats.py $DB $R/$O/r05_j_bench_idt_kernel_stats.csv > /dev/null; cp $DB $R/$O/r05_j_idt.db; rm -rf $R/$O/prof_r5j
cat $R/$O/r05_j_gaps_idt.txt
rocprofv3 --kernel-trace -d $R/$O/prof_r5k -o h -- python3 $R/bench.py --no-text --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/$O/prof_r5k.log 2>&1
DB=$(ls $R/$O/prof_r5k/*.db $R/$O/prof_r5k/*/*.db 2>/dev/null | head -1)
python3 $R/tools/rocpd_gaps.py $DB 0.5 40 loss_prep_kernel > $R/$O/r05_j_gaps_id.txt; rm -rf $R/$O/prof_r5k
cat $R/$O/r05_j_gaps_id.txt
