"""Loss and embedding spread per step of the bench's synthetic workload (random-init towers, uniform-noise images) in the LoRA
regime and under full fine-tuning, to see whether a flat loss is the optimisation (collapsed embeddings carry no InfoNCE
gradient) or the gradients.   python tools/ft_dynamics_probe.py [B] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from bioscanclip.hip.optim import FusedAdamW  # noqa: E402
from bioscanclip.model.loss_func import ContrastiveLoss  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
dev = torch.device("cuda", 0)
for full_ft, lr in ((False, 1e-3), (True, 5e-5), (True, 1e-3), (True, 1e-6)):
    model = bench.build_model(False, dev, full_ft=full_ft).train()
    image, dna, _ = bench.synthetic_batch(B, False, dev, seed=1234)
    label = torch.arange(B, device=dev)
    opt = FusedAdamW(model.parameters(), lr=lr)
    crit = ContrastiveLoss(torch.nn.CrossEntropyLoss(), 1 / 0.07)
    out = []
    for s in range(steps):
        opt.zero_grad()
        io, do, _ = model(image, dna, None)
        loss = crit(io, do, None, label)
        loss.backward()
        if s == 0:
            opt.attach(model)
            gn = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in model.parameters() if p.grad is not None)).item()
        opt.step()
        if s in (0, 1, 2, 4, 9, steps - 1):
            spread = lambda z: (1 - (z @ z.t()).mean()).item()          # 0 = all embeddings identical
            out.append(f"s{s}: loss {loss.item():.5f} spread img {spread(io.detach()):.2e} dna {spread(do.detach()):.2e}")
    print(f"full_ft={full_ft} lr={lr:g} |grad|_0={gn:.3e}  " + " | ".join(out), flush=True)
    del model, opt
    torch.cuda.empty_cache()
