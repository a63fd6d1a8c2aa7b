"""SURVEY 8f-4 -- full fine-tuning (``disable_lora: true``, reference bioscanclip/model/simple_clip.py:151-201): gradients of
EVERY parameter of each tower from the HIP full fine-tuning engines (hip/engine_ft.py) against the f32 oracle and against
the fixtures the imported reference produced (tests/golden/fullft.json, oracle/gen_golden.py:gen_fullft), then one whole
SimpleCLIP step (loss, backward, FusedAdamW over all parameters) against the oracle's step.

Depth 2, dropout 0 (the deterministic path).  Tolerances are <= 2x the values measured on the MI355X (gpurun_out/parity.jsonl):
the operands of every GEMM -- weight-gradient ones included -- are bf16, as in the LoRA regime.
"""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import check_summary, load_golden, rel_err  # noqa: E402
from oracle import refcpu, synth  # noqa: E402

NODROP = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
# (embedding, worst parameter-gradient tensor) vs the f32 oracle / the reference fixtures
TOL = {"dna": (1.1e-2, 3.3e-2), "txt": (6.3e-3, 1.9e-2), "vit": (2.2e-2, 6e-2)}   # measured 6.0e-3/1.6e-2, 3.1e-3/9.5e-3, 1.1e-2/3.0e-2
# tensors whose gradient is rounding noise around an exact zero: a key bias shifts all scores of a query equally
NOISE = ("attention.self.key.bias",)
LR_FT = 2e-5
# depth 12, oracle only: (embedding, worst gradient tensor); set from the measured values below (gpurun_out/parity.jsonl)
FULL_DEPTH_TOL = {"dna": (1.5e-2, 4e-2), "vit": (4.3e-2, 0.16)}   # measured 7.6e-3 / 2.0e-2 and 2.2e-2 / 7.9e-2


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _log(rec):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity.jsonl", "a") as f:
        f.write(json.dumps(rec) + "\n")


def _all_grads(sd, fn, cot):
    sd = {k: v.clone() for k, v in sd.items()}
    keys = [k for k in sd if sd[k].is_floating_point()]
    for k in keys:
        sd[k].requires_grad_(True)
    y = fn(sd)
    (y * cot).sum().backward()
    return y.detach(), {k: sd[k].grad for k in keys if sd[k].grad is not None}


def _cases():
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    from bioscanclip.model.language_encoder import LoRA_bert
    dna = synth.synth_batch(2, seed=21)[1]
    text = synth.synth_batch(4, seed=22, with_text=True)[2]
    image = synth.synth_batch(2, seed=23)[0]
    return {
        "dna": (lambda: LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=2, **NODROP)),
                                          r=4, num_classes=768, lora_layer=[]), "dna_encoder.", 11,
                lambda sd: refcpu.barcode_bert_encoder(sd, dna), lambda: dna.cuda()),
        "txt": (lambda: LoRA_bert(arch.BertModelParams(arch.bert_small_config(num_hidden_layers=2, **NODROP)), r=4,
                                  num_classes=768, lora_layer=[]), "language_encoder.", 12,
                lambda sd: refcpu.bert_text_encoder(sd, text), lambda: {k: v.cuda() for k, v in text.items()}),
        "vit": (lambda: LoRA_ViT_timm(arch.VisionTransformerParams(depth=2), r=4, num_classes=768, lora_layer=[]),
                "image_encoder.", 13, lambda sd: refcpu.vit_encoder(sd, image), lambda: image.cuda()),
    }


def _is_noise(k):
    # the BERT pooler is not on the path (BertModel's sequence output is mean-pooled, language_encoder.py:84-86); the tied MLM
    # decoder was replaced (dna_encoder.py:93-95): neither receives a gradient on either side
    return k.endswith(NOISE)


@pytest.mark.parametrize("name", ["dna", "txt", "vit"])
def test_all_parameter_gradients(name):
    build, prefix, seed, oracle_fn, hip_in = _cases()[name]
    m = build()
    sd = synth.synth_state_dict({prefix + k: v for k, v in synth.shapes_of(m).items()}, seed=seed)
    m.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    for p in m.parameters():
        p.requires_grad = True
    m.hip_full_ft = True
    m.to("cuda").train()
    y = m(hip_in())
    cot = synth.synth_tensor(f"{name}.cot.ft", y.shape, seed=5)
    (y * cot.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert type(m._engine).__name__.endswith("EngineFT")
    yo, go = _all_grads(sd, oracle_fn, cot)
    tol_emb, tol_grad = TOL[name]
    named = dict(m.named_parameters())
    rec = {"test": f"fullft_{name}", "emb_vs_f32_oracle": rel_err(y, yo), "grads": {}}
    worst, worst_key = 0.0, None
    for k, g in go.items():
        p = named[k[len(prefix):]]
        assert p.grad is not None, k
        assert torch.isfinite(p.grad).all(), k
        if _is_noise(k):
            assert p.grad.norm().item() < 1e-3 * max(1.0, float(max(v.norm() for v in go.values()))), k
            continue
        e = rel_err(p.grad, g)
        rec["grads"][k] = e
        if e > worst:
            worst, worst_key = e, k
    rec["worst_grad"], rec["worst_key"] = worst, worst_key
    # parameters off the path get no gradient from the oracle: the engine must leave theirs empty or zero
    for k, p in named.items():
        if prefix + k not in go and p.grad is not None:
            assert p.grad.abs().max().item() == 0.0, k
    _log(rec)
    assert rec["emb_vs_f32_oracle"] < tol_emb, rec
    assert worst < tol_grad, {k: v for k, v in rec.items() if k != "grads"} | {"top": sorted(rec["grads"].items(), key=lambda kv: -kv[1])[:8]}
    gold = load_golden("fullft")[name]   # the reference's own numbers
    check_summary(f"{name}.out.ft", y, gold["out"], tol_emb, what=name + " ")
    for k, gs in gold["grads"].items():
        if _is_noise(k):
            continue
        # * the fixture's projection vector is drawn by the synthetic value rule of the tensor's own key, which for LayerNorm gains
        #   is 1 + noise: there it is 50 x (sum of the elements) + a projection, and the sum of a zero-mean gradient is all
        #   cancellation.  Gains are held to norm + leading elements here and elementwise to the oracle above.
        # * gradient mass is concentrated (pos_embed row 0 is the cls token's, embedding rows of frequent ids), so the leading
        #   elements carry more than the even share check_summary assumes: slack 4.
        gain = k.endswith(("LayerNorm.weight", "norm.weight", "norm1.weight", "norm2.weight"))
        check_summary(k, named[k[len(prefix):]].grad, gs, tol_grad, what=name + " ", first_slack=4.0, probe=not gain)


@pytest.mark.parametrize("name", ["dna", "vit"])
def test_gradients_are_bitwise_reproducible(name):
    """No float atomics in this regime either: split-K slabs, embedding rows, bias and LayerNorm sums all add in a fixed order."""
    build, prefix, seed, _, hip_in = _cases()[name]
    m = build()
    sd = synth.synth_state_dict({prefix + k: v for k, v in synth.shapes_of(m).items()}, seed=seed)
    m.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    for p in m.parameters():
        p.requires_grad = True
    m.hip_full_ft = True
    m.to("cuda").train()
    x = hip_in()
    x = torch.cat([x] * 4) if torch.is_tensor(x) else x          # 8 samples: M = 1 064 / 1 576 rows, the split-K path
    runs = []
    for _ in range(2):
        for p in m.parameters():
            p.grad = None
        y = m(x)
        (y * synth.synth_tensor(f"{name}.cot.rep", y.shape, seed=5).cuda()).sum().backward()
        torch.cuda.synchronize()
        runs.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    assert len(runs[0]) > 30 and all(torch.equal(runs[0][k], runs[1][k]) for k in runs[0])


def test_weight_update_reaches_every_tensor():
    """Six FusedAdamW steps in the full fine-tuning regime (per-tensor launches: the optimizer is not attached to the engine):
    every tensor that has a gradient moves, the next forward uses the moved weights (the bf16 operand copies are re-packed
    from the f32 masters), and the loss follows the oracle's under torch.optim.AdamW."""
    from bioscanclip.hip.optim import FusedAdamW
    build, prefix, seed, oracle_fn, hip_in = _cases()["dna"]
    m = build()
    sd = synth.synth_state_dict({prefix + k: v for k, v in synth.shapes_of(m).items()}, seed=seed)
    m.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    for p in m.parameters():
        p.requires_grad = True
    m.hip_full_ft = True
    m.to("cuda").train()
    x = hip_in()
    target = synth.synth_tensor("ft.target", (2, 768), seed=9).cuda() / 0.02     # a fixed linear functional of the embedding
    opt = FusedAdamW([p for p in m.parameters()], lr=LR_FT, weight_decay=0.0)
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = (m(x) * target).sum()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    moved = {k: (p.detach() - before[k]).abs().max().item() for k, p in m.named_parameters()}
    on_path = [k for k, p in m.named_parameters() if p.grad is not None and p.grad.abs().max().item() > 0]
    assert len(on_path) > 30
    assert all(moved[k] > 0 for k in on_path), [k for k in on_path if moved[k] == 0][:5]
    # oracle: same six AdamW steps in f32
    sdo = {k: v.clone() for k, v in sd.items()}
    keys = [k for k in sdo if sdo[k].is_floating_point()]
    for k in keys:
        sdo[k].requires_grad_(True)
    oopt = torch.optim.AdamW([sdo[k] for k in keys], lr=LR_FT, weight_decay=0.0)
    olosses = []
    for _ in range(6):
        oopt.zero_grad()
        lo = (oracle_fn(sdo) * target.cpu()).sum()
        lo.backward()
        oopt.step()
        olosses.append(lo.item())
    _log({"test": "fullft_dna_adamw6", "hip": losses, "oracle": olosses})
    assert olosses[-1] < olosses[0] and losses[-1] < losses[0], (losses, olosses)
    fall, ofall = losses[0] - losses[-1], olosses[0] - olosses[-1]
    assert abs(fall - ofall) <= 0.1 * ofall, (losses, olosses)


def test_simple_clip_full_ft_trajectory():
    """The whole step in the full fine-tuning regime (I+D+T at depth 2, B=8, four steps over two batches): SimpleCLIP forward,
    ContrastiveLoss, backward through all three towers into every parameter, FusedAdamW -- against the oracle's train_step
    (reference train_epoch.py:28-42) with every floating-point tensor trainable (simple_clip.py:199-201)."""
    from bioscanclip.hip.optim import FusedAdamW
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    from bioscanclip.model.language_encoder import LoRA_bert
    from bioscanclip.model.loss_func import ContrastiveLoss
    from bioscanclip.model.simple_clip import SimpleCLIP, enable_full_fine_tuning
    model = SimpleCLIP(LoRA_ViT_timm(arch.VisionTransformerParams(depth=2), r=4, num_classes=768, lora_layer=[]),
                       LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=2, **NODROP)), r=4,
                                         num_classes=768, lora_layer=[]),
                       LoRA_bert(arch.BertModelParams(arch.bert_small_config(num_hidden_layers=2, **NODROP)), r=4, num_classes=768,
                                 lora_layer=[]))
    sd = synth.synth_state_dict(synth.shapes_of(model), seed=31)
    model.load_state_dict(sd)
    enable_full_fine_tuning(model)
    model.to("cuda").train()
    lr, steps, B = 1e-4, 4, 8
    opt = FusedAdamW(model.parameters(), lr=lr)
    crit = ContrastiveLoss(criterion=torch.nn.CrossEntropyLoss(), logit_scale=1 / 0.07)
    state = refcpu.StepState(sd)
    state.train_keys = [k for k in state.sd if state.sd[k].is_floating_point()]     # every tensor is a leaf here
    for k in state.train_keys:
        state.sd[k].requires_grad_(True)
    state.m = {k: torch.zeros_like(state.sd[k]) for k in state.train_keys}
    state.v = {k: torch.zeros_like(state.sd[k]) for k in state.train_keys}
    losses, olosses = [], []
    first = None
    for s in range(steps):
        image, dna, text, label = synth.synth_batch(B, seed=40 + s % 2, with_text=True)
        opt.zero_grad()
        io, do, to = model(image.cuda(), dna.cuda(), {k: v.cuda() for k, v in text.items()})
        loss = crit(io, do, to, label.cuda())
        loss.backward()
        lo, outs, grads = refcpu.train_step(state, image, dna, text, label, lr=lr)
        if s == 0:
            named = dict(model.named_parameters())
            gd = {k: rel_err(named[k].grad, g) for k, g in grads.items()
                  if g is not None and not _is_noise(k) and g.norm().item() > 0}
            first = {"emb": [rel_err(a, b) for a, b in zip((io, do, to), outs)], "worst_grad": max(gd.values()),
                     "worst_key": max(gd, key=gd.get), "n_grad_tensors": len(gd)}
            opt.attach(model)
        opt.step()
        losses.append(loss.item())
        olosses.append(lo.item())
    named = dict(model.named_parameters())
    # parameters after the last step: distance travelled from the initial weights, HIP vs oracle (the weights themselves are
    # O(1) and move by lr per step, so a plain relative error of the tensors would hide the update entirely)
    moved = {}
    for k in state.train_keys:
        if state.m[k].abs().max().item() == 0 or _is_noise(k):
            continue
        d_o = state.sd[k].detach() - sd[k]
        d_h = named[k].detach().cpu() - sd[k]
        moved[k] = ((d_h - d_o).norm() / d_o.norm().clamp_min(1e-30)).item()
    rec = {"test": "fullft_clip_trajectory", "first": first, "losses": losses, "oracle": olosses,
           "loss_rel_err": [abs(a - b) / abs(b) for a, b in zip(losses, olosses)],
           "update_rel_err_worst": max(moved.values()), "update_rel_err_worst_key": max(moved, key=moved.get),
           "update_rel_err_median": sorted(moved.values())[len(moved) // 2]}
    _log(rec)
    assert first["n_grad_tensors"] > 100, first
    assert max(first["emb"]) < 2.3e-2, rec
    assert first["worst_grad"] < 0.11, rec               # measured 5.5e-2
    assert max(rec["loss_rel_err"]) < 4e-3, rec          # measured 4.3e-4 (round 2), 1.0e-3 (round 3: gelu' codes round to nearest
    #                                                      even, the N = 8 loss runs on the slab form), 2.7e-3 (round 3, 256-entry GELU
    #                                                      table) -- grows step by step from 1.7e-4 at step 1: a trajectory, not one step
    assert olosses[-1] < olosses[0] and losses[-1] < losses[0], rec
    assert rec["update_rel_err_median"] < 0.18, rec      # measured 8.9e-2: AdamW normalises by sqrt(v), small gradients flip sign


def test_train_cl_full_fine_tuning_config(tmp_path, capsys):
    """scripts/train_cl.py with the full fine-tuning config (disable_lora: true, one-cycle LR; reference
    config/model_config/full_fine_tuning/one_cycle/*) at full depth with the HF dropout defaults active: the epoch runs, every
    parameter on the path has moved in the checkpoint, the loss is finite."""
    import sys as _sys
    scripts = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bioscan-clip_amd", "scripts")
    _sys.path.insert(0, scripts)
    import train_cl
    losses = train_cl.main(["model_config=full_fine_tuning/one_cycle/image_dna_one_cycle", "model_config.batch_size=8",
                            "model_config.epochs=1", "synthetic_steps_per_epoch=4", "save_ckpt=true", "debug_flag=false",
                            f"project_root_path={tmp_path}"])
    capsys.readouterr()
    assert len(losses) == 1 and all(torch.isfinite(torch.tensor(float(v))) for v in losses)
    ck = [os.path.join(r, f) for r, _, fs in os.walk(str(tmp_path)) for f in fs if f.endswith("last.pth")]
    sd = torch.load(ck[0], map_location="cpu")
    # no LoRA pairs in the BERT tower, LoRA on every ViT block
    assert not any(".w_a." in k or ".w_b." in k for k in sd) and sum(".linear_a_q." in k for k in sd) == 12
    assert all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())


@pytest.mark.parametrize("name", ["dna", "vit"])
def test_all_parameter_gradients_at_full_depth(name):
    """The same comparison at the reference's depth (12 layers / blocks), oracle only (the fixtures are depth 2): the chain runs
    through every layer down to the embeddings / patch filters, and the error does not grow out of the depth-2 band by more than
    the LoRA regime's does."""
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    if name == "dna":
        m = LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(**NODROP)), r=4, num_classes=768, lora_layer=[])
        prefix, seed = "dna_encoder.", 11
        x = synth.synth_batch(2, seed=21)[1]
        fn = lambda sd: refcpu.barcode_bert_encoder(sd, x)
    else:
        m = LoRA_ViT_timm(arch.vit_base_patch16_224(), r=4, num_classes=768, lora_layer=[])
        prefix, seed = "image_encoder.", 13
        x = synth.synth_batch(2, seed=23)[0]
        fn = lambda sd: refcpu.vit_encoder(sd, x)
    sd = synth.synth_state_dict({prefix + k: v for k, v in synth.shapes_of(m).items()}, seed=seed)
    m.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    for p in m.parameters():
        p.requires_grad = True
    m.hip_full_ft = True
    m.to("cuda").train()
    y = m(x.cuda())
    cot = synth.synth_tensor(f"{name}.cot.ft12", y.shape, seed=5)
    (y * cot.cuda()).sum().backward()
    torch.cuda.synchronize()
    yo, go = _all_grads(sd, fn, cot)
    named = dict(m.named_parameters())
    errs = {k: rel_err(named[k[len(prefix):]].grad, g) for k, g in go.items() if not _is_noise(k)}
    worst = max(errs, key=errs.get)
    rec = {"test": f"fullft_{name}_L12", "emb_vs_f32_oracle": rel_err(y, yo), "worst_grad": errs[worst], "worst_key": worst,
           "median_grad": sorted(errs.values())[len(errs) // 2], "n": len(errs)}
    _log(rec)
    assert len(errs) > 150 and rec["emb_vs_f32_oracle"] < FULL_DEPTH_TOL[name][0] and errs[worst] < FULL_DEPTH_TOL[name][1], rec


def test_full_fine_tuning_overfits_a_fixed_batch_with_dropout():
    """Dropout active (HF defaults) in the full fine-tuning regime: 40 AdamW steps over all parameters on one fixed batch drive
    the contrastive loss down -- the weight, LayerNorm and embedding gradients are taken under the same masks as the forward."""
    from bioscanclip.hip.optim import FusedAdamW
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    from bioscanclip.model.loss_func import ContrastiveLoss
    from bioscanclip.model.simple_clip import SimpleCLIP, enable_full_fine_tuning
    torch.manual_seed(5)
    model = SimpleCLIP(LoRA_ViT_timm(arch.VisionTransformerParams(depth=3), r=4, num_classes=768, lora_layer=[]),
                       LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=3)), r=4,
                                         num_classes=768, lora_layer=[]), None)
    enable_full_fine_tuning(model)
    model.to("cuda").train()
    image, dna, _, label = synth.synth_batch(16, seed=78)
    image, dna, label = image.cuda(), dna.cuda(), label.cuda()
    crit = ContrastiveLoss(criterion=torch.nn.CrossEntropyLoss(), logit_scale=1 / 0.07)
    opt = FusedAdamW(model.parameters(), lr=5e-5)
    losses = []
    for s in range(40):
        opt.zero_grad()
        loss = crit(*model(image, dna, None), label)
        loss.backward()
        if s == 0:
            opt.attach(model)
        opt.step()
        if s % 10 == 0 or s == 39:
            losses.append(loss.item())
    _log({"test": "fullft fixed-batch training with dropout", "losses": losses})
    # (round 3 draws other masks -- the step word goes through the mixer: 2.77 -> 1.67 here, 0.60 of the start; the bar is "falls
    # steadily under dropout", not a particular trajectory)
    assert all(l == l for l in losses) and losses[-1] < 0.7 * losses[0] and losses[3] < losses[1], losses
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_shared_gradients_equal_the_lora_regime(monkeypatch):
    """The ViT carries LoRA on every block in both regimes (App. B-3): with the same weights and inputs the full fine-tuning engine
    must produce the LoRA regime's embeddings and its LoRA / head gradients exactly -- same kernels on the shared part; the
    regime only adds gradients (and keeps per-layer copies of two activations).  Full fine-tuning keeps the f32 residual /
    residual-gradient streams and the plain bf16 patch GEMM (round 3 moved the LoRA regime's defaults to bf16 streams and the
    split-bf16 patch embedding), so the LoRA regime is switched to the same settings for this comparison."""
    from bioscanclip.hip import engine
    from bioscanclip.model import arch
    monkeypatch.setattr(engine, "RESID_STREAM_BF16", False)
    monkeypatch.setattr(engine, "GRAD_STREAM_BF16", False)
    monkeypatch.setattr(engine, "PATCH_SPLIT", False)
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    x = synth.synth_batch(4, seed=23)[0].cuda()
    outs = {}
    for regime in ("lora", "full"):
        m = LoRA_ViT_timm(arch.VisionTransformerParams(depth=3), r=4, num_classes=768, lora_layer=[])
        sd = synth.synth_state_dict({"image_encoder." + k: v for k, v in synth.shapes_of(m).items()}, seed=13)
        m.load_state_dict({k[len("image_encoder."):]: v for k, v in sd.items()})
        if regime == "full":
            for p in m.parameters():
                p.requires_grad = True
            m.hip_full_ft = True
        m.to("cuda").train()
        y = m(x)
        (y * synth.synth_tensor("vit.cot.shared", y.shape, seed=5).cuda()).sum().backward()
        torch.cuda.synchronize()
        outs[regime] = (y.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    (yl, gl), (yf, gf) = outs["lora"], outs["full"]
    assert torch.equal(yl, yf)
    assert len(gl) == 3 * 4 + 2 and len(gf) > 40 and set(gl) <= set(gf)
    for k in gl:
        assert torch.equal(gl[k], gf[k]), k
