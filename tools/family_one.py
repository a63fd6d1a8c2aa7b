"""Launch ONE pass of a GEMM family's launch mix of the B = 256 step (for rocprofv3 --pmc): python tools/family_one.py dx|fc1
The mix is bench.py's (family_shapes): sums of a counter over the pass / launches = the mix-weighted per-launch average."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bioscan-clip_amd")]
import torch  # noqa: E402

from bench import family_shapes  # noqa: E402
from bioscanclip.hip import ops  # noqa: E402
from bioscanclip.hip.lib import EPI_BF16, EPI_GELU_BF16  # noqa: E402

which = sys.argv[1]
B = int(os.environ.get("B", "256"))
shapes = family_shapes(B)[which]
ops.init_tables()
torch.cuda.synchronize()
for M, N, K, cnt in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    kw = {}
    if which == "fc1":
        kw = {"bias": torch.randn(N, device="cuda"), "aux": torch.empty(M, N, device="cuda", dtype=torch.uint8)}
    torch.cuda.synchronize()
    for _ in range(cnt):
        ops.gemm(a, w, out, EPI_BF16 if which == "dx" else EPI_GELU_BF16, **kw)
    torch.cuda.synchronize()
print("launches", sum(c for *_, c in shapes))
