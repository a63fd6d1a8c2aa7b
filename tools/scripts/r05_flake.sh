#!/bin/bash
# how often does graph replay differ from eager in the exact mode with three towers?  (one process, the test repeated)
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
timeout -k 10 1000 python - > gpurun_out/r05_flake.log 2>&1 <<'PY'
import pytest, sys
res = []
for i in range(8):
    rc = pytest.main(["tests/test_30_graph_gpu.py", "-x", "-q", "-k", "test_graph_replay_equals_eager_step and True-False-True", "-p", "no:cacheprovider"])
    res.append(int(rc))
    print("RUN", i, "rc", int(rc), flush=True)
print("RESULTS", res)
PY
grep -E "RUN|RESULTS|At index|diff:" gpurun_out/r05_flake.log | head -40
