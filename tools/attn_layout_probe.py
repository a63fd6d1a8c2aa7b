import sys
import os, sys
sys.path.insert(0, "/root/repo/bioscan-clip_amd")
import torch
from bioscanclip.hip import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiments"))
import attn_sweep_ops as xo  # noqa: E402  (round-4 experiment kernels: diagnostic library only since ABI 9)
def t(fn, n=10):
    fn(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3
for S in (197, 133):
    for name, B, heads in (("token-major [B*S, 12 heads x 3 x 64] (the step's layout)", 256, 12), ("head-major: every (b, head) item contiguous", 3072, 1)):
        H = heads * 64
        qkv = (torch.randn(B * S, 3 * H, device="cuda") * 0.5).bfloat16()
        dctx = torch.randn(B * S, H, device="cuda").bfloat16()
        ctx = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16)
        ctx_lo = torch.empty_like(ctx); stats = torch.empty(B, heads, S, 4, device="cuda")
        dqkv = torch.empty(B * S, 3 * H, device="cuda", dtype=torch.bfloat16)
        lse = torch.empty(B, heads, S, device="cuda")
        f = t(lambda: ops.attn_fwd(qkv, B, S, heads, 0.125, ctx, lse))
        b = t(lambda: ops.attn_bwd(qkv, dctx, lse, B, S, heads, 0.125, dqkv))
        xo.attn_fwd2(qkv, B, S, heads, 0.125, ctx, ctx_lo, stats)
        b2 = t(lambda: xo.attn_bwd2(qkv, dctx, ctx, ctx_lo, stats, B, S, heads, 0.125, dqkv))
        print(f"S={S} {name}: fwd {f:6.1f} us  bwd (two-phase) {b:6.1f} us  bwd2 (persistent sweep) {b2:6.1f} us", flush=True)
