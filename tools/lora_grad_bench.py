import sys
sys.path.insert(0, "/root/repo/bioscan-clip_amd")
import torch
from bioscanclip.hip import ops
for name, M in (("vit", 50432), ("dna", 34048)):
    H = 768
    dqkv = torch.randn(M, 3 * H, device="cuda").bfloat16()
    h = torch.randn(M, H + 64, device="cuda").bfloat16()
    lb = torch.randn(2, H, 4, device="cuda") * 0.02
    dt = torch.empty(M, 8, device="cuda")
    dA, dBq, dBv = torch.zeros(8, H, device="cuda"), torch.zeros(H, 4, device="cuda"), torch.zeros(H, 4, device="cuda")
    fn = lambda: ops.lora_grad(dqkv, h, M, H, lb, dt, dA, dBq, dBv)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: lora_grad (3 kernels) {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
