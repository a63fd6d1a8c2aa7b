#!/bin/bash
# round 5, GPU run B: full -m gpu suite on the new tree, the layout probe re-run, PMC passes of the dominant family, bench line
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/bioscan-clip_amd
O=gpurun_out
rm -f $O/parity.jsonl
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/r05_b_gpu_tests.log 2>&1; rc=$?
tail -5 $O/r05_b_gpu_tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert" $O/r05_b_gpu_tests.log | tail -20; exit 1; }
cp $O/parity.jsonl $O/r05_b_parity.jsonl
TREE=$(cat tools/scripts/.tree 2>/dev/null) bash tools/scripts/r05_dx_pmc.sh > $O/r05_b_pmc.log 2>&1 || { tail -20 $O/r05_b_pmc.log; exit 1; }
tail -2 $O/r05_b_pmc.log | cut -c1-600
cp $O/r05_dx_pmc.json $O/r05_fc1_pmc.json profiles/ 2>/dev/null
timeout -k 10 600 python bench.py --steps 20 > $O/r05_b_bench_line.json 2> $O/r05_b_bench.log || { tail -20 $O/r05_b_bench.log; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_b_bench_line.json"))
print("headline", d["ms_per_step"], d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["avg_launch_ms"], "lowest", d["roofline_lowest"]["frac"])
for k,v in d.get("extra",{}).items(): print(k, v if not isinstance(v,dict) else v.get("ms_per_step"))
print("parity", d.get("parity_mode_ms_per_step"), "exact", d.get("exact_mode_ms_per_step"))
for f in d.get("roofline_families",[]): print(f["family"], f["avg_launch_us"], f["frac_of_bf16_mfma_peak"])
PY
