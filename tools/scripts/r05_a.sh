#!/bin/bash
# round 5, GPU run A: keep-bit / preload attention kernels -- tests, isolated timings, step A/B
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/bioscan-clip_amd
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_00_abi.py tests/test_10_kernels_gpu.py tests/test_40_dropout_gpu.py -x -q -m "gpu or not gpu" -k "abi or attention or dropout or derivative or train_vs_eval or small_ops" > $O/r05_a_tests.log 2>&1 || { tail -30 $O/r05_a_tests.log; exit 1; }
tail -3 $O/r05_a_tests.log
timeout -k 10 200 python tools/attn_bench.py > $O/r05_a_attn_bench.log 2>&1 && BSCLIP_ATTN_PRELOAD=0 timeout -k 10 200 python tools/attn_bench.py >> $O/r05_a_attn_bench.log 2>&1 && LAYOUT=head timeout -k 10 200 python tools/attn_bench.py >> $O/r05_a_attn_bench.log 2>&1
cat $O/r05_a_attn_bench.log
for rep in 1 2; do
  BSCLIP_ATTN_KEEP_BITS=0 BSCLIP_ATTN_PRELOAD=0 timeout -k 10 300 python bench.py --no-text --steps 20 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_a_bench_old_$rep.json || exit 1
  timeout -k 10 300 python bench.py --no-text --steps 20 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_a_bench_new_$rep.json || exit 1
  python - <<PY
import json
for k in ("old","new"):
    d=json.load(open("$O/r05_a_bench_%s_$rep.json"%k)); print(k, $rep, d["ms_per_step"], d["value"])
PY
done
