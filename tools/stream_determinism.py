"""Runs the same training step with the tower streams on and off and lists every gradient tensor that is not bitwise equal
between runs (how the shared lora_grad workspace race was found).  Diagnostic; the pytest version is
tests/test_20_encoders_gpu.py::test_tower_streams_join_before_gradients_are_read."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bioscan-clip_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import test_20_encoders_gpu as T
from oracle import synth
import bioscanclip.model.simple_clip as sc
from bioscanclip.model.loss_func import ContrastiveLoss
model, _ = T._build_clip(False, 77)
model.to("cuda").train()
crit = ContrastiveLoss(criterion=torch.nn.CrossEntropyLoss(), logit_scale=1 / 0.07)
image, dna, _, label = synth.synth_batch(16, seed=5)
image, dna, label = image.cuda(), dna.cuda(), label.cuda()
res = []
for mode in (True, False, True, False):
    sc._TOWER_STREAMS = mode
    for p in model.parameters():
        if p.grad is not None:
            p.grad.zero_()
    io, do, to = model(image, dna, None)
    loss = crit(io, do, to, label)
    loss.backward()
    torch.cuda.synchronize()
    res.append(({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, io.detach().clone(), do.detach().clone(), loss.item()))
for a, b, name in ((0, 2, "streams vs streams"), (1, 3, "serial vs serial"), (0, 1, "streams vs serial")):
    print(name, "loss", res[a][3], res[b][3], "io equal", torch.equal(res[a][1], res[b][1]), "do equal", torch.equal(res[a][2], res[b][2]))
    bad = [(k, (res[a][0][k] - res[b][0][k]).abs().max().item(), res[a][0][k].abs().max().item()) for k in res[a][0] if not torch.equal(res[a][0][k], res[b][0][k])]
    print("  differing tensors:", len(bad), bad[:6])
