"""Section times of attn_bwd_kernel from the diagnostic build (bsclip_attn_bwd_diag): python tools/attn_phases.py [S]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch  # noqa: E402

from bioscanclip.hip import lib, ops  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 197
B, heads, H = int(os.environ.get("B", "256")), 12, 768
qkv = (torch.randn(B * S, 3 * H, device="cuda") * 0.5).bfloat16()
dctx = torch.randn(B * S, H, device="cuda").bfloat16()
ctx = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16)
dqkv = torch.empty(B * S, 3 * H, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, heads, S, device="cuda")
ops.attn_fwd(qkv, B, S, heads, 0.125, ctx, lse)
diag = torch.zeros(B * heads * 4 * 8, dtype=torch.int64, device="cuda")
h = lib.load_diag()
for _ in range(2):
    rc = h.bsclip_attn_bwd_diag(qkv.data_ptr(), qkv.stride(0), dctx.data_ptr(), dctx.stride(0), lse.data_ptr(), B, S, heads,
                                ctypes.c_float(0.125), dqkv.data_ptr(), dqkv.stride(0), diag.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
d = diag.view(B * heads, 4, 8).double().cpu() * 0.01  # us
names = ["stage K,V", "phase 1 (delta + dQ)", "wait at barrier", "stage Q,dO", "phase 2 (dK,dV)"]
print(f"S={S}: kernel span {(d[:, :, 5].max() - d[:, :, 0].min()).item():.1f} us; per-workgroup "
      f"{(d[:, :, 5].max(1).values - d[:, :, 0].min(1).values).mean().item():.1f} us")
for w in range(4):
    row = [f"{(d[:, w, i + 1] - d[:, w, i]).mean().item():6.1f}" for i in range(5)]
    print(f"  wave {w}: " + "  ".join(f"{n}: {v}" for n, v in zip(names, row)))
    print(f"          first block of phase 1: loads + pass 1 (delta) {(d[:, w, 6] - d[:, w, 1]).mean().item():5.1f}   "
          f"pass 2 (dQ) + store {(d[:, w, 7] - d[:, w, 6]).mean().item():5.1f}")
