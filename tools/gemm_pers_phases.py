"""Section times of the persistent 256x256 GEMM kernel (diagnostic build, per-workgroup stamps of its first two tiles)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch
from bioscanclip.hip import lib as L
from bioscanclip.hip.lib import EPI_BF16, EPI_DGELU_BF16, EPI_GELU_BF16, EPI_RESID_BF16, EPI_RESID_F32, EpiArgs
h = L.load_diag()
h.bsclip_init_tables(None)
M = int(os.environ.get("M", 256 * 197))
WGS = int(os.environ.get("WGS", 256))
for name, N, K, epi in (("qkv", 2304, 832, EPI_BF16), ("fc1", 3072, 768, EPI_GELU_BF16), ("dfc2", 3072, 768, EPI_DGELU_BF16),
                        ("proj", 768, 768, EPI_RESID_BF16), ("fc2", 768, 3072, EPI_RESID_BF16), ("fc2_f32", 768, 3072, EPI_RESID_F32)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == EPI_RESID_F32 else torch.bfloat16)
    args = EpiArgs(); args.bias = bias.data_ptr()
    keep = []
    if epi in (EPI_RESID_F32, EPI_RESID_BF16):
        r = torch.randn(M, N, device="cuda").to(out.dtype); keep.append(r); args.resid = r.data_ptr(); args.ld_resid = N
    if epi in (EPI_GELU_BF16, EPI_DGELU_BF16):
        z = torch.randint(0, 256, (M, N), device="cuda", dtype=torch.uint8); keep.append(z); args.aux = z.data_ptr(); args.ld_aux = N
    nt = ((M + 255) // 256) * (N // 256)
    grid = min(nt, WGS)
    diag = torch.zeros(grid * 32, dtype=torch.int64, device="cuda")
    for _ in range(3):
        rc = h.bsclip_gemm_pers_diag(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, epi, ctypes.byref(args),
                                     diag.data_ptr(), WGS, None)
        assert rc == 0, L.last_error()
    torch.cuda.synchronize()
    raw = diag.cpu().reshape(grid, 32)
    d = raw.double() / 100.0  # us
    two = raw[:, 2] >= 2
    total = d[:, 1].max() - d[:, 0].min()
    nk = K // 64
    med = lambda x: float(x.median())
    pro0 = med(d[:, 14] - d[:, 0]); k0 = med(d[:, 15] - d[:, 14])
    line = f"{name:8s} tiles {nt:5d} on {grid} WGs (rounds {nt / grid:5.2f}) nk {nk:3d}: kernel {total:7.1f} us | tile 0: prologue {pro0:5.2f} K-loop {k0:6.2f} ({k0 / nk:5.2f}/K-tile)"
    if two.any():
        t = d[two]
        gap = med(t[:, 4] - t[:, 15])                 # tile 0's epilogue (+ the hand-over)
        k1 = med(t[:, 5] - t[:, 4])
        secs = [med(t[:, 6 + i] - t[:, 5 + i]) for i in range(8)]
        line += (f" | tile 0 epilogue {gap:5.2f} | tile 1: K-loop {k1:6.2f} ({k1 / nk:5.2f}/K-tile) epilogue {sum(secs):5.2f} = "
                 + " ".join(f"{x:4.2f}" for x in secs) + "  (stage q | store q, q = 0..3)")
    if two.any() and os.environ.get("KTILES"):
        t = d[two]
        n = min(nk, 16)
        ends = [t[:, 4]] + [t[:, 16 + i] for i in range(n)]
        line += " | tile 1 K-tiles: " + " ".join(f"{med(ends[i + 1] - ends[i]):4.2f}" for i in range(n))
    clk = raw[:, 3].double() / ((raw[:, 1] - raw[:, 0]).double() * 10.0)   # cycles per ns = GHz (100 MHz wall clock ticks)
    line += f" | s_memtime clock {float(clk.median()):.2f} GHz"
    per_wg = d[:, 1] - d[:, 0]
    line += f" | WG lifetime median {med(per_wg):6.1f} max {float(per_wg.max()):6.1f}"
    print(line, flush=True)
