"""Weight-gradient GEMM dW = dY^T X at the ViT / BarcodeBERT shapes of a B=256 step: transposes + NT split-K against the TN
split-K kernel that reads the row-major activations (bioscanclip/hip/engine_ft.py:_dw)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch  # noqa: E402

from bioscanclip.hip import ops  # noqa: E402
from bioscanclip.hip.engine import split_plan  # noqa: E402


def t_us(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


partial = torch.empty(512 * 65536, device="cuda")
for name, M, N, K in (("vit.qkv", 50432, 2304, 768), ("vit.proj", 50432, 768, 768), ("vit.fc1", 50432, 3072, 768), ("vit.fc2", 50432, 768, 3072),
                      ("dna.qkv", 34048, 2304, 768), ("dna.fc1", 34048, 3072, 768), ("dna.fc2", 34048, 768, 3072)):
    S, Mp = split_plan(M, N, K)
    dY = torch.zeros(Mp, N, device="cuda", dtype=torch.bfloat16)
    X = torch.zeros(Mp, K, device="cuda", dtype=torch.bfloat16)
    dY[:M] = torch.randn(M, N, device="cuda").bfloat16()
    X[:M] = torch.randn(M, K, device="cuda").bfloat16()
    tA, tB = torch.zeros(N, Mp, device="cuda", dtype=torch.bfloat16), torch.zeros(K, Mp, device="cuda", dtype=torch.bfloat16)
    gw1, gw2, gb = torch.zeros(N, K, device="cuda"), torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")

    def nt():
        ops.transpose_colsum_bf16(dY, M, N, tA, gb)
        ops.transpose_bf16(X, M, K, tB)
        ops.gemm_splitk_f32(tA, tB, gw1, S, partial, K=Mp)

    def nt_gemm_only():
        ops.gemm_splitk_f32(tA, tB, gw1, S, partial, K=Mp)

    def tn():
        ops.gemm_tn_splitk_f32(dY, X, gw2, S, partial, Kp=Mp)
    nt()
    gw1.zero_(); gw2.zero_()
    nt_gemm_only(); tn()
    err = ((gw1 - gw2).norm() / gw1.norm()).item()
    a, b, c = t_us(nt), t_us(nt_gemm_only), t_us(tn)
    fl = 2.0 * M * N * K
    print(f"{name:9s} M={M} N={N} K={K} S={S}: transposes+NT {a:7.1f} us (GEMM alone {b:7.1f} us = {fl / b / 1e6:6.0f} TF)   TN {c:7.1f} us = {fl / c / 1e6:6.0f} TF   rel diff {err:.1e}")
