"""Debug driver for the captured step: python tools/graph_probe.py [streams 0/1] [text 0/1] [train 0/1]
Runs a graphed model and an eager twin in lockstep and reports the first step / tensor where they part."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bioscan-clip_amd"), os.path.join(ROOT, "tests")]
import faulthandler
faulthandler.enable()
import torch
import bioscanclip.model.simple_clip as sc
sc._TOWER_STREAMS = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
with_text = (sys.argv[2] if len(sys.argv) > 2 else "0") == "1"
train = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
from test_30_graph_gpu import _build
from oracle import synth
from bioscanclip.hip.graph import GraphedStep
from bioscanclip.hip.optim import FusedAdamW
from bioscanclip.model.loss_func import ContrastiveLoss
crit = ContrastiveLoss(torch.nn.CrossEntropyLoss(), 1 / 0.07)
ma, mb = _build(91, with_text), _build(91, with_text)
if not train:
    ma.eval(); mb.eval()
oa, ob = FusedAdamW(ma.parameters(), lr=1e-3), FusedAdamW(mb.parameters(), lr=1e-3)
ob.enable_device_hyper(True)
g = GraphedStep(ma, oa, crit, warmup=2)
if not train:
    g_body = g._body
image, dna, text, label = synth.synth_batch(16, seed=300, with_text=with_text)
image, dna, label = image.cuda(), dna.cuda(), label.cuda()
text = None if text is None else {k: v.cuda() for k, v in text.items()}
fresh = (sys.argv[4] if len(sys.argv) > 4 else "0") == "1"
for s in range(6):
    if fresh:   # new input tensors every step: GraphedStep copies them into its static buffers
        i2, d2, _, l2 = synth.synth_batch(16, seed=300 + s, with_text=False)
        image, dna, label = i2.cuda(), d2.cuda(), l2.cuda()
    la = g(image, dna, text, label)
    if os.environ.get("PTRS"):
        torch.cuda.synchronize()
        print("   ptrs: loss_buf %x fresh image %x dna %x label %x | static image %x dna %x label %x | loss val %r" % (
            g.loss_buf.data_ptr(), image.data_ptr(), dna.data_ptr(), label.data_ptr(), g.static[0].data_ptr(),
            g.static[1].data_ptr(), g.static[3].data_ptr(), g.loss_buf.item()), flush=True)
    ob.zero_grad()
    lb = crit(*mb(image, dna, text), label)
    lb.backward()
    if ob.needs_attach():
        ob.attach(mb)
    ob.step()
    torch.cuda.synchronize()
    pa = {k: p for k, p in ma.named_parameters() if p.requires_grad}
    pb = {k: p for k, p in mb.named_parameters() if p.requires_grad}
    badp = [k for k in pa if not torch.equal(pa[k], pb[k])]
    badg = [k for k in pa if pa[k].grad is not None and not torch.equal(pa[k].grad, pb[k].grad)]
    print(f"step {s} {'graph' if g.graph is not None else 'eager'}: loss {la.item():.6f} vs eager twin {lb.item():.6f}; "
          f"{len(badg)} grads / {len(badp)} params differ {badg[:3]} {[(k, pa[k].grad.abs().max().item(), pb[k].grad.abs().max().item()) for k in badg[:2]]}", flush=True)
