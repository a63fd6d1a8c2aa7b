"""Section times and co-residency of the 256x128 duo GEMM kernel (diagnostic build, per-workgroup stamps)."""
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
    grid = (M // 256) * (N // 128)
    diag = torch.zeros(grid * 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        rc = h.bsclip_gemm_duo_diag(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, epi, ctypes.byref(args),
                                    diag.data_ptr(), None)
        assert rc == 0, L.last_error()
    torch.cuda.synchronize()
    raw = diag.cpu().reshape(grid, 8)
    d = raw[:, :4].double() / 100.0  # us
    t0 = d[:, 0].min()
    total = d[:, 3].max() - t0
    nk = K // 64
    pro, loop, ep = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2]
    # co-residency: CU key = (xcc, se, sh, cu); count, at each workgroup's midpoint, how many workgroups are alive on its CU
    hw, xcc = raw[:, 4], raw[:, 5] & 0xF
    key = (xcc << 16) | (hw & 0xFF00)      # cu_id 11:8, sh_id 12, se_id 15:13
    mid = (d[:, 0] + d[:, 3]) / 2
    alive = []
    import collections
    by = collections.defaultdict(list)
    for i in range(grid):
        by[int(key[i])].append(i)
    for k, idx in by.items():
        s = d[idx, 0]; e = d[idx, 3]
        for i in idx:
            alive.append(int(((s <= mid[i]) & (e >= mid[i])).sum()))
    alive = torch.tensor(alive).double()
    print(f"{name:5s} grid {grid:5d} nk {nk:3d}: kernel {total:7.1f} us | per WG median: prologue {pro.median():5.2f}  K-loop {loop.median():6.2f} "
          f"({loop.median() / nk:5.2f}/tile)  epilogue {ep.median():6.2f} | WG total {(d[:, 3] - d[:, 0]).median():6.2f} | CUs seen {len(by)} "
          f"workgroups alive per CU (mean at WG midpoints) {alive.mean():.2f}  max {int(alive.max())}", flush=True)
