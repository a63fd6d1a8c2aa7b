"""Section times of attn_bwd_pers_kernel (diagnostic build, bsclip_attn_bwd_pers_diag): python tools/attn_pers_phases.py [S]
Stamps are taken on each workgroup's SECOND item (steady state: its tiles were prefetched during the first)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch  # noqa: E402

from bioscanclip.hip import lib, ops  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiments"))
import attn_sweep_ops as xo  # noqa: E402  (round-4 experiment kernels: diagnostic library only since ABI 9)

S = int(sys.argv[1]) if len(sys.argv) > 1 else 197
B, heads, H = int(os.environ.get("B", "256")), 12, 768
NB = (S + 31) // 32
NW = NB + 1
qkv = (torch.randn(B * S, 3 * H, device="cuda") * 0.5).bfloat16()
dctx = torch.randn(B * S, H, device="cuda").bfloat16()
ctx = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16)
ctx_lo = torch.empty_like(ctx)
dqkv = torch.empty(B * S, 3 * H, device="cuda", dtype=torch.bfloat16)
stats = torch.empty(B, heads, S, 4, device="cuda")
xo.attn_fwd2(qkv, B, S, heads, 0.125, ctx, ctx_lo, stats)
grid = min(256, B * heads)
diag = torch.zeros(grid * NW * 16, dtype=torch.int64, device="cuda")
h = lib.load_diag()
for _ in range(2):
    rc = h.bsclip_attn_bwd_pers_diag(qkv.data_ptr(), qkv.stride(0), dctx.data_ptr(), dctx.stride(0), ctx.data_ptr(), ctx_lo.data_ptr(),
                                     ctx.stride(0), stats.data_ptr(), B, S, heads, ctypes.c_float(0.125), dqkv.data_ptr(), dqkv.stride(0),
                                     diag.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
d = diag.view(grid, NW, 16).double().cpu() * 0.01  # us
ko = d[:, :NB]
names = [(0, 1, "issue prefetch + O loads"), (1, 2, "block 0"), (3, 4, "wait ring slot (block 1)"), (2, 5, f"blocks 1..{NB - 3}"),
         (5, 6, "confirm prefetch (vmcnt 0)"), (6, 7, f"block {NB - 2} + wait + delta"), (7, 8, f"block {NB - 1}"), (8, 9, "stores"),
         (9, 10, "item barrier"), (0, 10, "whole item")]
print(f"S={S} B={B}: second item of every workgroup, key-owner waves (mean over workgroups and waves, us)")
for a, b, n in names:
    print(f"    {n:32s} {(ko[:, :, b] - ko[:, :, a]).mean().item():7.2f}   (slowest wave {(ko[:, :, b] - ko[:, :, a]).mean(0).max().item():7.2f})")
dq = d[:, NB]
print("  dQ wave:")
for a, b, n in [(0, 1, "wait for block 0's dS tiles"), (1, 2, "dQ of block 0 + stores"), (2, 3, "wait for block 1's dS tiles"),
                (3, 4, "dQ of block 1 + stores"), (0, 9, "whole item"), (9, 10, "item barrier")]:
    print(f"    {n:32s} {(dq[:, b] - dq[:, a]).mean().item():7.2f}")
