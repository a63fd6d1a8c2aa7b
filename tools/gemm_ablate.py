"""Ablation of the 256x256 ping-pong K loop: the diagnostic EPI_BF16 build with parts of the loop compiled out (MFMA, LDS
fragment reads, LDS-DMA, barriers) -- which resource sets the 1.45 us per K-tile?  Results of ablated runs are garbage; only
the K-loop time between the phase stamps matters.  python tools/gemm_ablate.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch  # noqa: E402

from bioscanclip.hip import lib as L  # noqa: E402
from bioscanclip.hip.lib import EPI_BF16, EpiArgs  # noqa: E402

h = L.load_diag()
M = 256 * 197
NAMES = {0: "full loop", 1: "no MFMA", 2: "no LDS reads", 4: "no DMA", 8: "no barriers", 6: "no LDS reads, no DMA",
         3: "no MFMA, no LDS reads", 5: "no MFMA, no DMA", 9: "no MFMA, no barriers"}
for name, N, K in (("qkv", 2304, 832), ("dfc1", 768, 3072)):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    args = EpiArgs()
    grid = (M // 256) * (N // 256)
    nk = K // 64
    print(f"{name}: M={M} N={N} K={K} ({nk} K-tiles, {grid} workgroups)")
    for mask in (0, 1, 2, 4, 8, 6, 3, 5, 9):
        assert h.bsclip_gemm_diag_ablate(mask) == 0
        diag = torch.zeros(grid * 16, dtype=torch.int64, device="cuda")
        for _ in range(3):
            rc = h.bsclip_gemm_diag(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, EPI_BF16, ctypes.byref(args),
                                    diag.data_ptr(), None)
            assert rc == 0, L.last_error()
        torch.cuda.synchronize()
        d = diag.cpu().reshape(grid, 2, 8).double() / 100.0
        loop = (d[:, :, 2] - d[:, :, 1]).median().item()
        total = (d[:, :, 3].max() - d[:, :, 0].min()).item()
        print(f"   {NAMES[mask]:26s} K loop {loop:7.2f} us = {loop / nk:5.2f} us per K-tile    kernel {total:7.1f} us", flush=True)
h.bsclip_gemm_diag_ablate(0)
