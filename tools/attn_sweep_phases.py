"""Section times of attn_bwd_sweep_kernel from the diagnostic build (bsclip_attn_bwd2_diag): python tools/attn_sweep_phases.py [S]
B (env) * 12 heads workgroups: B=256 is the step's launch, B <= 21 leaves every workgroup alone on its CU."""
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
NW = (S + 31) // 32 + 1
qkv = (torch.randn(B * S, 3 * H, device="cuda") * 0.5).bfloat16()
dctx = torch.randn(B * S, H, device="cuda").bfloat16()
ctx = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16)
ctx_lo = torch.empty_like(ctx)
dqkv = torch.empty(B * S, 3 * H, device="cuda", dtype=torch.bfloat16)
stats = torch.empty(B, heads, S, 4, device="cuda")
xo.attn_fwd2(qkv, B, S, heads, 0.125, ctx, ctx_lo, stats)
diag = torch.zeros(B * heads * NW * 8, dtype=torch.int64, device="cuda")
h = lib.load_diag()
for _ in range(2):
    rc = h.bsclip_attn_bwd2_diag(qkv.data_ptr(), qkv.stride(0), dctx.data_ptr(), dctx.stride(0), ctx.data_ptr(), ctx_lo.data_ptr(),
                                 ctx.stride(0), stats.data_ptr(), B, S, heads, ctypes.c_float(0.125), dqkv.data_ptr(), dqkv.stride(0),
                                 diag.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
d = diag.view(B * heads, NW, 8).double().cpu() * 0.01  # us
names = ["delta loads + sums", "wait for tiles", "first query block", "other query blocks", "stores / tail"]
print(f"S={S} B={B}: kernel span {(d[:, :, 5].max() - d[:, :, 0].min()).item():.1f} us; per-workgroup "
      f"{(d[:, :, 5].max(1).values - d[:, :, 0].min(1).values).mean().item():.1f} us")
for w in range(NW):
    row = [f"{(d[:, w, i + 1] - d[:, w, i]).mean().item():6.2f}" for i in range(5)]
    print(f"  wave {w} ({'dQ' if w == NW - 1 else 'keys'}): " + "  ".join(f"{n}: {v}" for n, v in zip(names, row)))
