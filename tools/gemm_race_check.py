"""Bitwise comparison of an experimental K-loop schedule of the 256x256 kernel (tile mode on the command line) with the proven
one (tile 4) over many launches and shapes.  MFMA arithmetic is deterministic and identical in both, so ANY mismatch is a
synchronisation bug (a fragment read racing an LDS-DMA write).  python tools/gemm_race_check.py 7 [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch  # noqa: E402

from bioscanclip.hip import ops  # noqa: E402
from bioscanclip.hip.lib import EPI_BF16, EPI_F32  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 7
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
shapes = [(50432, 768, 768), (50432, 2304, 832), (34048, 768, 3072), (5000, 1024, 64), (256 * 7 + 13, 512, 128),
          (34048, 3072, 768), (20000, 256, 1920)]
bad = 0
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    for epi, dt in ((EPI_BF16, torch.bfloat16), (EPI_F32, torch.float32)):
        ref = torch.empty(M, N, device="cuda", dtype=dt)
        ops.set_gemm_tile(4)
        ops.gemm(a, w, ref, epi)
        out = torch.empty_like(ref)
        ops.set_gemm_tile(mode)
        for r in range(rounds):
            out.zero_()
            ops.gemm(a, w, out, epi)
            if not torch.equal(out, ref):
                bad += 1
                d = (out.float() - ref.float()).abs()
                print(f"MISMATCH M={M} N={N} K={K} epi={epi} round {r}: {int((d > 0).sum())} elements, max {d.max().item():.3e}",
                      flush=True)
                break
    print(f"M={M} N={N} K={K}: done", flush=True)
ops.set_gemm_tile(0)
print("mismatching (shape, epilogue) pairs:", bad)
sys.exit(1 if bad else 0)
