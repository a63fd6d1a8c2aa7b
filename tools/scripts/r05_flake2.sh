#!/bin/bash
# the exact-mode graph-vs-eager comparison inside a long-lived process (the context it failed in once): suite prefix, then test_30 repeated
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export PYTHONPATH=$R:$R/bioscan-clip_amd
timeout -k 10 1100 python - > gpurun_out/r05_flake2.log 2>&1 <<'PY'
import pytest
rc = pytest.main(["tests/test_10_kernels_gpu.py", "tests/test_20_encoders_gpu.py", "-x", "-q", "-p", "no:cacheprovider"])
print("PREFIX rc", int(rc), flush=True)
res = []
for i in range(6):
    rc = pytest.main(["tests/test_30_graph_gpu.py", "-q", "-k", "test_graph_replay_equals_eager_step", "-p", "no:cacheprovider"])
    res.append(int(rc))
    print("RUN", i, "rc", int(rc), flush=True)
print("RESULTS", res)
PY
grep -E "PREFIX|RUN|RESULTS|At index|FAILED" gpurun_out/r05_flake2.log | head -40
