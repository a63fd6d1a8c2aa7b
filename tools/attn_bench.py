"""Times bsclip_attn_fwd / bsclip_attn_bwd (two-phase backward) and bsclip_attn_fwd2 / bsclip_attn_bwd2 (key-owner sweep) at the step's shapes (B=256: ViT S=197 no dropout, DNA S=133 dropout 0.1)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch  # noqa: E402

from bioscanclip.hip import ops  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiments"))
import attn_sweep_ops as xo  # noqa: E402  (round-4 experiment kernels: diagnostic library only since ABI 9)

B = int(os.environ.get("B", "256"))
for name, S, p in (("vit", 197, 0.0), ("dna", 133, 0.1), ("dna-nodrop", 133, 0.0)):
    heads, H = 12, 768
    qkv = (torch.randn(B * S, 3 * H, device="cuda") * 0.5).bfloat16()
    dctx = torch.randn(B * S, H, device="cuda").bfloat16()
    ctx = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16)
    dqkv = torch.empty(B * S, 3 * H, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, heads, S, device="cuda")
    drop = (p, 1234) if p else None
    ctx_lo = torch.empty_like(ctx)
    stats = torch.empty(B, heads, S, 4, device="cuda")
    res = {}
    for what, fn in (("fwd", lambda: ops.attn_fwd(qkv, B, S, heads, 0.125, ctx, lse, dropout=drop)),
                     ("bwd", lambda: ops.attn_bwd(qkv, dctx, lse, B, S, heads, 0.125, dqkv, dropout=drop)),
                     ("fwd2", lambda: xo.attn_fwd2(qkv, B, S, heads, 0.125, ctx, ctx_lo, stats, dropout=drop)),
                     ("bwd2", lambda: xo.attn_bwd2(qkv, dctx, ctx, ctx_lo, stats, B, S, heads, 0.125, dqkv, dropout=drop))):
        fn()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        res[what] = best
    fl = 4.0 * B * heads * S * S * 64
    print(f"{name:11s} S={S} p={p}: fwd {res['fwd']*1e3:7.1f} us ({fl/res['fwd']/1e9:6.1f} TF)   "
          f"bwd {res['bwd']*1e3:7.1f} us ({2.5*fl/res['bwd']/1e9:6.1f} TF at 5 products)   |  sweep pair: fwd2 {res['fwd2']*1e3:7.1f} us   "
          f"bwd2 {res['bwd2']*1e3:7.1f} us ({2.5*fl/res['bwd2']/1e9:6.1f} TF)", flush=True)
