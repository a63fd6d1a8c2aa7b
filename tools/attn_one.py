"""Launches attn fwd + bwd a few times at one shape (for rocprofv3 --pmc): python tools/attn_one.py [S] [p]
With dropout the forward writes the keep-bit words and the backward reads them (the product path since round 5)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch  # noqa: E402

from bioscanclip.hip import ops  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 197
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
B, heads, H = 256, 12, 768
qkv = (torch.randn(B * S, 3 * H, device="cuda") * 0.5).bfloat16()
dctx = torch.randn(B * S, H, device="cuda").bfloat16()
ctx = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16)
dqkv = torch.empty(B * S, 3 * H, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, heads, S, device="cuda")
drop = (p, 1234) if p else None
bits = torch.zeros(B * heads * S * ops.KEEP_WORDS, device="cuda", dtype=torch.int32) if p else None
for _ in range(3):
    ops.attn_fwd(qkv, B, S, heads, 0.125, ctx, lse, dropout=drop, keep_bits=bits)
    ops.attn_bwd(qkv, dctx, lse, B, S, heads, 0.125, dqkv, dropout=drop, keep_bits=bits)
torch.cuda.synchronize()
