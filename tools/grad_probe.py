"""First-step gradients of the BASELINE configs[0] trajectory (I+D, B=8) against the CPU oracle: full-tensor relative L2
error per trainable tensor (the golden fixtures only hold fingerprints).  Diagnostic; uses oracle/ as the checker."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from oracle import refcpu, synth  # noqa: E402
from helpers import load_golden  # noqa: E402
import test_20_encoders_gpu as T  # noqa: E402
from bioscanclip.model.loss_func import ContrastiveLoss  # noqa: E402

g = load_golden("trajectory_id")
model, sd = T._build_clip(False, g["weight_seed"])
model.to("cuda").train()
crit = ContrastiveLoss(criterion=torch.nn.CrossEntropyLoss(), logit_scale=1 / 0.07)
image, dna, text, label = synth.synth_batch(g["B"], seed=g["batch_seed0"], with_text=False)
io, do, to = model(image.cuda(), dna.cuda(), None)
loss = crit(io, do, to, label.cuda())
loss.backward()
named = dict(model.named_parameters())
torch.set_num_threads(16)
state = {k: v.clone() for k, v in sd.items()}
keys = [k for k in state if refcpu.is_trainable_key(k) and state[k].is_floating_point()]
for k in keys:
    state[k].requires_grad_(True)
fi, fd, ft = refcpu.simple_clip_forward(state, image, dna, None)
ref_loss = refcpu.contrastive_loss(fi, fd, ft, label)
grads = torch.autograd.grad(ref_loss, [state[k] for k in keys])
print("loss", loss.item(), ref_loss.item())
errs = []
for k, gr in zip(keys, grads):
    mine = named[k].grad.detach().cpu().double()
    e = ((mine - gr.double()).norm() / gr.double().norm().clamp_min(1e-30)).item()
    errs.append((e, k, gr.norm().item()))
errs.sort(reverse=True)
for e, k, n in errs[:12]:
    print(f"{e:8.4f}  |g|={n:.3e}  {k}")
import statistics
print("median", statistics.median(e for e, _, _ in errs), "n", len(errs))
q = [e for e, k, _ in errs if "query" in k or "_q." in k]
print("Q-LoRA tensors: mean", sum(q) / len(q), "max", max(q))
