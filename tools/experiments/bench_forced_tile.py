import sys, os, runpy
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd")); sys.path.insert(0, ROOT)
from bioscanclip.hip import ops
ops.set_gemm_tile(int(os.environ.get("FORCE_TILE", "8")))
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extras"]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
