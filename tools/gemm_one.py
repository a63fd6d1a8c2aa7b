"""Run one GEMM shape a few times (for rocprofv3 --pmc).  usage: gemm_one.py NAME"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch
from bioscanclip.hip import ops
from bioscanclip.hip.lib import EPI_BF16, EPI_DGELU_BF16, EPI_GELU_BF16, EPI_RESID_F32
SH = {"qkv": (2304, 832, EPI_BF16), "dfc1": (768, 3072, EPI_BF16), "fc1": (3072, 768, EPI_GELU_BF16),
      "dfc2": (3072, 768, EPI_DGELU_BF16), "fc1_dna": (3072, 768, EPI_GELU_BF16), "tr_dna": (768, 768, EPI_GELU_BF16), "fc2": (768, 3072, EPI_RESID_F32), "proj": (768, 768, EPI_RESID_F32)}
for name in sys.argv[1:]:
    M = 256 * 133 if name.endswith("_dna") else 256 * 197
    N, K, epi = SH[name]
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == EPI_RESID_F32 else torch.bfloat16)
    kw = {}
    if epi == EPI_RESID_F32: kw["resid"] = torch.randn(M, N, device="cuda")
    if epi in (EPI_GELU_BF16, EPI_DGELU_BF16): kw["aux"] = torch.randint(0, 256, (M, N), device="cuda", dtype=torch.uint8)
    for _ in range(3):
        ops.gemm(a, w, out, epi, bias=bias, **kw)
    torch.cuda.synchronize()
