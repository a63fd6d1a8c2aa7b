#!/bin/bash
# closing run of round 5 on the final tree: the whole GPU suite, then the validation set (smoke, bench line and variants, kernel stats, PMC)
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
O=gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
rm -f $O/parity.jsonl
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/r05_z_gpu_tests.log 2>&1; rc=$?; tail -5 $O/r05_z_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
cp $O/parity.jsonl $O/r05_z_parity.jsonl
T=z bash tools/scripts/r05_final_validation.sh > $O/r05_z_validation.log 2>&1; rc=$?; tail -60 $O/r05_z_validation.log | cut -c1-400
exit $rc
