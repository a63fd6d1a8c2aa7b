"""Forward GEMM shapes of the step: bf16 ping-pong kernel vs the fp8 forms (interleaved rounds, one process, random data)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch  # noqa: E402

from bioscanclip.hip import ops  # noqa: E402
from bioscanclip.hip.lib import EPI_BF16, EPI_GELU_BF16, EPI_GELU_FP8, EPI_RESID_F32  # noqa: E402

B = int(os.environ.get("B", "256"))
ops.init_tables()
SHAPES = []
for name, M in (("vit", B * 197), ("dna", B * 133)):
    SHAPES += [(f"{name}.qkv", M, 2304, 768, "qkv"), (f"{name}.proj", M, 768, 768, "resid"), (f"{name}.fc1", M, 3072, 768, "gelu"),
               (f"{name}.fc2", M, 768, 3072, "resid")]


def timed(fn, iters=10):
    best = 1e9
    for _ in range(3):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


tot = {"bf16": 0.0, "fp8 form1": 0.0, "fp8 form2": 0.0}
for name, M, N, K, kind in SHAPES:
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.03
    bias = torch.randn(N, device="cuda")
    a16, w16 = a.bfloat16(), w.bfloat16()
    a8 = a.to(ops.FP8)
    w8, ws = ops.quantize_rows_fp8(w.contiguous())
    resid = torch.randn(M, N, device="cuda") if kind == "resid" else None
    aux = torch.empty(M, N, device="cuda", dtype=torch.uint8) if kind == "gelu" else None
    aug_a16 = torch.cat([a16, torch.zeros(M, 64, device="cuda", dtype=torch.bfloat16)], 1) if kind == "qkv" else None
    aug_w16 = torch.cat([w16, torch.zeros(N, 64, device="cuda", dtype=torch.bfloat16)], 1) if kind == "qkv" else None
    t_aug = torch.zeros(M, 64, device="cuda", dtype=torch.bfloat16) if kind == "qkv" else None
    b_aug = torch.zeros(N, 64, device="cuda", dtype=torch.bfloat16) if kind == "qkv" else None
    if kind == "qkv":
        o16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        f16 = lambda: ops.gemm(aug_a16, aug_w16, o16, EPI_BF16, bias=bias)
        f8 = lambda form: ops.gemm_fp8(a8, w8, o16, ws, bias, EPI_BF16, a_aug=t_aug, b_aug=b_aug, form=form)
    elif kind == "resid":
        o32 = torch.empty(M, N, device="cuda")
        f16 = lambda: ops.gemm(a16, w16, o32, EPI_RESID_F32, bias=bias, resid=resid)
        f8 = lambda form: ops.gemm_fp8(a8, w8, o32, ws, bias, EPI_RESID_F32, resid=resid, form=form)
    else:
        o16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        o8 = torch.empty(M, N, device="cuda", dtype=ops.FP8)
        f16 = lambda: ops.gemm(a16, w16, o16, EPI_GELU_BF16, bias=bias, aux=aux)
        f8 = lambda form: ops.gemm_fp8(a8, w8, o8, ws, bias, EPI_GELU_FP8, aux=aux, form=form)
    r = {"bf16": timed(f16), "fp8 form1": timed(lambda: f8(1)), "fp8 form2": timed(lambda: f8(2))}
    fl = 2.0 * M * N * K
    print(f"{name:9s} M={M:6d} N={N:5d} K={K:5d}  " + "  ".join(f"{k}: {v * 1e3:7.1f}us {fl / v / 1e9:7.1f}TF" for k, v in r.items()),
          flush=True)
    for k, v in r.items():
        tot[k] += v
print("sum of the four forward GEMMs of one layer, both towers (ms):", {k: round(v, 3) for k, v in tot.items()})
