"""The captured-step path (bioscanclip/hip/graph.py) must be the eager step: same losses and parameters step for step, with
dropout at the HF defaults (fresh masks on every replay, the backward regenerating the forward's), a moving learning rate
(the schedule reaches the AdamW nodes through device memory) and the towers on their own streams."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import rel_err  # noqa: E402
from oracle import synth  # noqa: E402


def _build(seed, with_text):
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    from bioscanclip.model.language_encoder import LoRA_bert
    from bioscanclip.model.simple_clip import SimpleCLIP
    torch.manual_seed(seed)
    model = SimpleCLIP(LoRA_ViT_timm(arch.VisionTransformerParams(depth=3), r=4, num_classes=768),
                       LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=3)), r=4, num_classes=768),
                       LoRA_bert(arch.BertModelParams(arch.bert_small_config(num_hidden_layers=2)), r=4, num_classes=768)
                       if with_text else None)
    model.load_state_dict(synth.synth_state_dict(synth.shapes_of(model), seed=seed))
    return model.cuda().train()     # HF dropout 0.1 / 0.1 active


@pytest.mark.parametrize("with_text", [False, True])
def test_graph_replay_equals_eager_step(with_text):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bioscanclip.hip.graph import GraphedStep
    from bioscanclip.hip.optim import FusedAdamW
    from bioscanclip.model.loss_func import ContrastiveLoss
    steps = 8
    batches = [synth.synth_batch(16, seed=300 + s % 3, with_text=with_text) for s in range(steps)]
    cuda = lambda t: None if t is None else ({k: v.cuda() for k, v in t.items()} if isinstance(t, dict) else t.cuda())
    runs = {}
    for mode in ("eager", "graph"):
        model = _build(91, with_text)
        opt = FusedAdamW(model.parameters(), lr=1e-3)
        opt.enable_device_hyper(True)   # both modes on bsclip_adamw_step_dev (checked against the host-argument form below)
        sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=3e-3, total_steps=steps, pct_start=0.3, anneal_strategy="cos",
                                                    cycle_momentum=False)
        crit = ContrastiveLoss(torch.nn.CrossEntropyLoss(), 1 / 0.07)
        g = GraphedStep(model, opt, crit, warmup=2) if mode == "graph" else None
        losses = []
        for s in range(steps):
            image, dna, text, label = (cuda(t) for t in batches[s])
            if g is not None:
                loss = g(image, dna, text, label)
            else:
                opt.zero_grad()
                loss = crit(*model(image, dna, text), label)
                loss.backward()
                if opt.needs_attach():
                    opt.attach(model)
                opt.step()
            sched.step()
            losses.append(loss.item())
        if g is not None:
            assert g.graph is not None
        runs[mode] = (losses, {k: p.detach().clone() for k, p in model.named_parameters() if p.requires_grad})
    le, lg = runs["eager"][0], runs["graph"][0]
    assert len(set(round(x, 6) for x in le)) == steps          # dropout + new batches: every step differs
    assert le == lg, (le, lg)                                   # same kernels, same seeds, same order: bit for bit
    for k, v in runs["eager"][1].items():
        assert torch.equal(v, runs["graph"][1][k]), k


def test_adamw_device_hyper_equals_host_arguments():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bioscanclip.hip import ops
    g = torch.Generator().manual_seed(1)
    n = 100003
    p0 = torch.randn(n, generator=g).cuda()
    pa, pb = p0.clone(), p0.clone()
    ma, va, mb, vb = (torch.zeros(n, device="cuda") for _ in range(4))
    hyper = torch.zeros(2, device="cuda")
    for step in range(1, 40):
        gr = torch.randn(n, generator=g).cuda()
        lr = 1e-3 * (1 + 0.1 * step)
        ops.adamw_step(pa, gr, ma, va, lr, 0.9, 0.999, 1e-8, 0.01, step)
        hyper.copy_(torch.tensor([lr, float(step)]))
        ops.adamw_step_dev(pb, gr, mb, vb, hyper, 0.9, 0.999, 1e-8, 0.01)
    assert rel_err(pb, pa) < 1e-7
