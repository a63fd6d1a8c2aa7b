"""Host-side cost of one kernel launch through the Python wrappers (shape checks + ctypes): python tools/host_overhead.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch  # noqa: E402

from bioscanclip.hip import ops  # noqa: E402
from bioscanclip.hip.lib import EPI_BF16, EPI_RESID_F32  # noqa: E402

a = torch.randn(256, 768, device="cuda").bfloat16()
w = torch.randn(768, 768, device="cuda").bfloat16()
out = torch.empty(256, 768, device="cuda", dtype=torch.bfloat16)
outf = torch.empty(256, 768, device="cuda")
bias = torch.randn(768, device="cuda")
x = torch.randn(256, 768, device="cuda")
g = torch.ones(768, device="cuda")
st = torch.empty(256, 2, device="cuda")
y = torch.empty(256, 768, device="cuda", dtype=torch.bfloat16)


def timeit(name, fn, n=2000):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{name:34s} {1e6 * (t1 - t0) / n:6.1f} us per call (host enqueue)", flush=True)


timeit("ops.gemm plain", lambda: ops.gemm(a, w, out, EPI_BF16))
timeit("ops.gemm bias+resid", lambda: ops.gemm(a, w, outf, EPI_RESID_F32, bias=bias, resid=x))
timeit("ops.layernorm_fwd", lambda: ops.layernorm_fwd(x, g, g, 1e-6, y_bf16=y, stats=st))
timeit("torch.empty (allocator)", lambda: torch.empty(256, 768, device="cuda"))
timeit("tensor.zero_()", lambda: outf.zero_())
