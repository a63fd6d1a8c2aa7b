"""Times every GEMM shape of the training step for each tile configuration (interleaved rounds, one process)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch  # noqa: E402

from bioscanclip.hip import ops  # noqa: E402
from bioscanclip.hip.lib import EPI_BF16, EPI_DGELU_BF16, EPI_GELU_BF16, EPI_RESID_BF16, EPI_RESID_F32  # noqa: E402

EPI_R = EPI_RESID_F32 if os.environ.get('RESID', 'bf16') == 'f32' else EPI_RESID_BF16   # the residual stream's dtype (default: the engines')

B = int(os.environ.get("B", "256"))
TILES = tuple(int(t) for t in os.environ.get("TILES", "1,2,3,4,5").split(","))
SHAPES = []
for name, M in (("vit", B * 197), ("dna", B * 133)):
    SHAPES += [(f"{name}.qkv", M, 2304, 832, EPI_BF16), (f"{name}.proj", M, 768, 768, EPI_R),
               (f"{name}.fc1", M, 3072, 768, EPI_GELU_BF16), (f"{name}.fc2", M, 768, 3072, EPI_R),
               (f"{name}.dfc2", M, 3072, 768, EPI_DGELU_BF16), (f"{name}.dfc1", M, 768, 3072, EPI_BF16),
               (f"{name}.dproj", M, 768, 768, EPI_BF16), (f"{name}.dqkv", M, 768, 2304, EPI_BF16)]


def run(M, N, K, epi, iters):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda")
    f32 = epi == EPI_RESID_F32
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    kw = {}
    if epi in (EPI_RESID_F32, EPI_RESID_BF16):
        kw["resid"] = torch.randn(M, N, device="cuda").to(out.dtype)
    if epi in (EPI_GELU_BF16, EPI_DGELU_BF16):
        kw["aux"] = torch.randint(0, 256, (M, N), device="cuda", dtype=torch.uint8)
    res = {}
    for rnd in range(3):
        for tile in TILES:
            ops.set_gemm_tile(tile)
            ops.gemm(a, w, out, epi, bias=bias, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.gemm(a, w, out, epi, bias=bias, **kw)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(tile, []).append(e0.elapsed_time(e1) / iters)
    ops.set_gemm_tile(0)
    # library reference (hipBLASLt / rocBLAS through torch): plain bf16 NT GEMM without any epilogue
    o2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    wt = w.t()
    lib = []
    for rnd in range(3):
        torch.matmul(a, wt, out=o2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            torch.matmul(a, wt, out=o2)
        e1.record()
        torch.cuda.synchronize()
        lib.append(e0.elapsed_time(e1) / iters)
    res["lib"] = lib
    return {t: min(v) for t, v in res.items()}


tot = {**{t: 0.0 for t in TILES}, "best": 0.0, "lib": 0.0}
for name, M, N, K, epi in SHAPES:
    r = run(M, N, K, epi, 10)
    fl = 2.0 * M * N * K
    best = min((t for t in r if t != 'lib'), key=r.get)
    print(f"{name:10s} M={M:6d} N={N:5d} K={K:5d}  " + "  ".join(
        f"t{t}: {r[t]*1e3:7.1f}us {fl/r[t]/1e9:7.1f}TF" for t in TILES + ("lib",)) + f"   best=t{best}", flush=True)
    for t in TILES + ("lib",):
        tot[t] += r[t]
    tot["best"] += r[best]
print("sum per layer (ms):", {k: round(v, 3) for k, v in tot.items()})
