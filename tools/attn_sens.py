"""Sensitivity probe for bsclip_attn_bwd: relative error of dQ/dK/dV against float64 torch autograd on inputs where the keys
of a head are nearly parallel (softmax backward cancels to a small remainder) and with a tiny upstream gradient."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch  # noqa: E402

from bioscanclip.hip import ops  # noqa: E402

torch.manual_seed(0)
for S, B, spread, gscale in ((133, 8, 1.0, 1.0), (133, 8, 0.02, 1.0), (197, 4, 0.02, 1e-6), (197, 4, 4.0, 1.0), (197, 4, 12.0, 1.0), (197, 4, 30.0, 1.0)):
    heads, H = 12, 768
    base = torch.randn(B, 1, 3, heads, 64, device="cuda")
    qkv = (base + spread * torch.randn(B, S, 3, heads, 64, device="cuda")).reshape(B * S, 3 * H).bfloat16()
    dctx = (torch.randn(B * S, H, device="cuda") * gscale).bfloat16()
    ctx = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16)
    dqkv = torch.empty(B * S, 3 * H, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, heads, S, device="cuda")
    ops.attn_fwd(qkv, B, S, heads, 0.125, ctx, lse)
    ops.attn_bwd(qkv, dctx, lse, B, S, heads, 0.125, dqkv)
    x = qkv.double().reshape(B, S, 3, heads, 64).requires_grad_(True)
    q, k, v = [x[:, :, i].transpose(1, 2) for i in range(3)]
    o = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2).reshape(B * S, H)
    (g,) = torch.autograd.grad(o, x, dctx.double())
    g = g.reshape(B * S, 3 * H)
    errs = [((dqkv[:, i * H:(i + 1) * H].double() - g[:, i * H:(i + 1) * H]).norm() / g[:, i * H:(i + 1) * H].norm()).item()
            for i in range(3)]
    print(f"S={S} B={B} spread={spread} gscale={gscale}: rel err dq {errs[0]:.4f} dk {errs[1]:.4f} dv {errs[2]:.4f} "
          f"| |dq|={g[:, :H].norm().item():.3e}", flush=True)
