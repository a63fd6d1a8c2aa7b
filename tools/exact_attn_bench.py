"""Times the exact-mode attention (bsclip_attn_fwd_f32 / bsclip_attn_bwd_f32) at the step's shapes (B = 256) in its implementations:
0 = split-bf16 operands on the bf16 matrix cores (csrc/attn_x3.hip, round 5), 2 = f32 operands on the matrix pipe (round 4)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch  # noqa: E402

from bioscanclip.hip import ops  # noqa: E402

B = int(os.environ.get("B", "256"))
for name, S, p in (("vit", 197, 0.0), ("dna", 133, 0.1)):
    heads, H = 12, 768
    qkv = torch.randn(B * S, 3 * H, device="cuda") * 0.5
    dctx = torch.randn(B * S, H, device="cuda")
    ctx = torch.empty(B * S, H, device="cuda")
    dqkv = torch.empty(B * S, 3 * H, device="cuda")
    lse = torch.empty(B, heads, S, device="cuda")
    drop = (p, 1234) if p else None
    line = f"{name:4s} S={S} p={p}:"
    for impl in (0, 2):
        ops.exact_attn_set_impl(impl)
        res = {}
        for what, fn in (("fwd", lambda: ops.attn_fwd_f32(qkv, B, S, heads, 0.125, ctx, lse, dropout=drop)),
                         ("bwd", lambda: ops.attn_bwd_f32(qkv, dctx, ctx, lse, B, S, heads, 0.125, dqkv, dropout=drop))):
            fn()
            best = 1e9
            for _ in range(2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 4)
            res[what] = best
        line += f"   impl {impl}: fwd {res['fwd']*1e3:7.1f} us  bwd {res['bwd']*1e3:7.1f} us"
    print(line, flush=True)
ops.exact_attn_set_impl(0)
