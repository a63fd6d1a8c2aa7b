"""InfoNCE forward + backward time, fused epilogues vs f32 logits slabs, at the loss sizes of BASELINE configs[1..3]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch
from bioscanclip.hip import ops
for N, nmod, nl in ((256, 2, 256), (2048, 3, 256), (8192, 3, 1024)):
    g = torch.Generator().manual_seed(N)
    zs = [torch.nn.functional.normalize(torch.randn(N, 768, generator=g), dim=-1).cuda() for _ in range(nmod)]
    label = torch.arange(N).cuda()
    ws = torch.empty(ops.infonce_workspace_floats(N, nmod), device="cuda")
    out = {}
    for impl in (2, 1, 0):   # fused (forced), logits slabs, by size (the default)
        ops.infonce_set_impl(impl)
        loss = torch.zeros(1, device="cuda")
        dz = [torch.empty(nl, 768, device="cuda") for _ in range(nmod)]
        for _ in range(2):
            ops.infonce_fwd_bwd(zs, label, 1 / 0.07, loss, dz, row0=0, n_local=nl, workspace=ws)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.infonce_fwd_bwd(zs, label, 1 / 0.07, loss, dz, row0=0, n_local=nl, workspace=ws)
        e1.record()
        torch.cuda.synchronize()
        out[impl] = (e0.elapsed_time(e1) / 10, loss.item())
    ops.infonce_set_impl(0)
    print(f"N={N:5d} modalities={nmod} local rows={nl:5d}: fused {out[2][0]:8.3f} ms   logits slabs {out[1][0]:8.3f} ms   "
          f"default (by size) {out[0][0]:8.3f} ms   (loss {out[2][1]:.6f} / {out[1][1]:.6f})", flush=True)
