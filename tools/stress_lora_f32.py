"""Is bsclip_lora_grad_f32 bit-reproducible while other streams keep the GPU busy?  (debugging the rare graph-vs-eager mismatch of the exact
mode with three towers: a few words of the dA / dB slabs off by ~1e-5.)  Reference = a quiet run; then N runs with load on two other streams."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch  # noqa: E402

from bioscanclip.hip import ops  # noqa: E402
from bioscanclip.hip.lib import EPI_F32  # noqa: E402

torch.manual_seed(0)
M, H = int(os.environ.get("M", 2128)), 768
N = int(os.environ.get("N", 300))
LOAD = os.environ.get("LOAD", "gemm,attn,ln").split(",")
dqkv = torch.randn(M, 3 * H, device="cuda") * 1e-3
y = torch.randn(M, H, device="cuda")
A = torch.randn(8, H, device="cuda") * 0.05
Bm = torch.randn(2, H, 4, device="cuda") * 0.05
key = (M, H, str(dqkv.device), None)


def run(stream):
    dA, dB = torch.zeros(8, H, device="cuda"), torch.zeros(2, H, 4, device="cuda")
    with torch.cuda.stream(stream):
        ops.lora_grad_f32(dqkv, y, M, H, A, Bm, dA, dB)
    return dA, dB


sa, sb, sc = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
dA0, dB0 = run(sa)
torch.cuda.synchronize()
wkey = [k for k in ops._LG32_WS if k[0] == M and k[1] == H][0]
ws0 = ops._LG32_WS[wkey].clone()
# load generators
Ml = 3152
a3 = torch.randn(Ml, 3 * 768, device="cuda").bfloat16()
w3 = torch.randn(768, 3 * 768, device="cuda").bfloat16()
out = torch.empty(Ml, 768, device="cuda")
qkv32 = torch.randn(16 * 197, 2304, device="cuda") * 0.5
ctx32 = torch.empty(16 * 197, 768, device="cuda")
lse = torch.empty(16, 12, 197, device="cuda")
dctx32 = torch.randn(16 * 197, 768, device="cuda")
dq32 = torch.empty(16 * 197, 2304, device="cuda")
g, b = torch.randn(768, device="cuda"), torch.randn(768, device="cuda")
yln = torch.empty(Ml, 768, device="cuda")
xln = torch.randn(Ml, 768, device="cuda")
st = torch.empty(Ml, 2, device="cuda")


M2, H2 = 320, 512
dqkv2 = torch.randn(M2, 3 * H2, device="cuda") * 1e-3
y2 = torch.randn(M2, H2, device="cuda")
A2 = torch.randn(8, H2, device="cuda") * 0.05
B2 = torch.randn(2, H2, 4, device="cuda") * 0.05
dA2, dB2 = torch.zeros(8, H2, device="cuda"), torch.zeros(2, H2, 4, device="cuda")
M3 = 3152
dqkv3 = torch.randn(M3, 3 * H, device="cuda") * 1e-3
y3 = torch.randn(M3, H, device="cuda")
dA3, dB3 = torch.zeros(8, H, device="cuda"), torch.zeros(2, H, 4, device="cuda")


def load(stream, n):
    with torch.cuda.stream(stream):
        for _ in range(n):
            if "lora512" in LOAD:
                ops.lora_grad_f32(dqkv2, y2, M2, H2, A2, B2, dA2, dB2)
            if "lora768" in LOAD:
                ops.lora_grad_f32(dqkv3, y3, M3, H, A, Bm, dA3, dB3)
            if "gemm" in LOAD:
                ops.gemm(a3, w3, out, EPI_F32)
            if "attn" in LOAD:
                ops.attn_fwd_f32(qkv32, 16, 197, 12, 0.125, ctx32, lse)
                ops.attn_bwd_f32(qkv32, dctx32, ctx32, lse, 16, 197, 12, 0.125, dq32)
            if "ln" in LOAD:
                ops.layernorm_fwd(xln, g, b, 1e-6, y_f32=yln, stats=st)


bad = 0
for it in range(N):
    load(sb, 3)
    load(sc, 3)
    dA, dB = run(sa)
    torch.cuda.synchronize()
    ws = ops._LG32_WS[wkey]
    if not (torch.equal(dA, dA0) and torch.equal(dB, dB0) and torch.equal(ws, ws0)):
        bad += 1
        d = (ws - ws0)
        idx = torch.nonzero(d != 0).flatten()
        Mp = (M + 3) // 4 * 4
        seg = lambda o: "t" if o < 8 * Mp else "dt" if o < 16 * Mp else "pa" if o < 16 * Mp + 768 * 8 * H else "pb"
        print(f"run {it}: {idx.numel()} workspace words differ (segments {sorted(set(seg(int(o)) for o in idx[:2000].tolist()))}), first {int(idx[0]) if idx.numel() else -1}, "
              f"max abs diff {d.abs().max().item():.3e}; dA equal {torch.equal(dA, dA0)} dB equal {torch.equal(dB, dB0)}", flush=True)
print(f"LOAD={LOAD}: {bad} of {N} runs differ from the quiet reference")
