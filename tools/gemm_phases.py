"""Where does a 256x256 GEMM workgroup spend its time?  Runs the diagnostic build (phase stamps) on the step's shapes."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch
from bioscanclip.hip import lib as L
from bioscanclip.hip.lib import EPI_BF16, EPI_DGELU_BF16, EPI_GELU_BF16, EPI_RESID_F32, EpiArgs
h = L.load_diag()
M = 256 * 197
for name, N, K, epi in (("qkv", 2304, 832, EPI_BF16), ("dfc1", 768, 3072, EPI_BF16), ("fc1", 3072, 768, EPI_GELU_BF16),
                        ("dfc2", 3072, 768, EPI_DGELU_BF16), ("fc2", 768, 3072, EPI_RESID_F32), ("proj", 768, 768, EPI_RESID_F32)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == EPI_RESID_F32 else torch.bfloat16)
    args = EpiArgs(); args.bias = bias.data_ptr()
    keep = []
    if epi == EPI_RESID_F32:
        r = torch.randn(M, N, device="cuda"); keep.append(r); args.resid = r.data_ptr(); args.ld_resid = N
    if epi in (EPI_GELU_BF16, EPI_DGELU_BF16):
        z = torch.randint(0, 256, (M, N), device="cuda", dtype=torch.uint8); keep.append(z); args.aux = z.data_ptr(); args.ld_aux = N
    grid = (M // 256) * (N // 256)
    diag = torch.zeros(grid * 16, dtype=torch.int64, device="cuda")
    for _ in range(3):
        rc = h.bsclip_gemm_diag(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, epi, ctypes.byref(args),
                                diag.data_ptr(), None)
        assert rc == 0, L.last_error()
    torch.cuda.synchronize()
    d = diag.cpu().reshape(grid, 2, 8).double() / 100.0  # us
    t0 = d[:, :, 0].min()
    pro = (d[:, :, 1] - d[:, :, 0]); loop = (d[:, :, 2] - d[:, :, 1]); epi_t = (d[:, :, 3] - d[:, :, 2])
    total = d[:, :, 3].max() - t0
    nk = K // 64
    print(f"{name:5s} grid {grid:5d} nk {nk:3d}: kernel {total:7.1f} us | per WG median: prologue {pro.median():5.2f}  K-loop {loop.median():6.2f} "
          f"({loop.median() / nk:5.2f}/tile)  epilogue {epi_t.median():6.2f}  | WG total {(d[:, :, 3] - d[:, :, 0]).median():6.2f}  "
          f"rounds {grid / 256:.2f}  sum-over-rounds {(d[:, :, 3] - d[:, :, 0]).median() * -(-grid // 256):7.1f}", flush=True)
    seq = [2, 4, 5, 6, 7, 3]
    parts = [(d[:, :, b] - d[:, :, a]).median().item() for a, b in zip(seq[:-1], seq[1:])]
    print("        epilogue sections (us): stage slab 0 " + f"{parts[0]:5.2f}  rows/consume 0 {parts[1]:5.2f}  stage slab 1 {parts[2]:5.2f}  "
          f"rows/consume 1 {parts[3]:5.2f}  tail {parts[4]:5.2f}", flush=True)
