"""End-to-end parity of the HIP encoders / loss / step against (a) the CPU oracle on identical seeded weights and
inputs and (b) the golden fixtures produced by the imported reference (tests/golden, oracle/gen_golden.py).

The fixtures are non-degenerate on purpose (oracle/synth.py): attention rows are peaked, embeddings of different samples
are distinct, the 10-step loss trajectory falls from above ln N to a few percent of it.  In that regime bf16 GEMM operands
cost more than on near-uniform attention, because the softmax turns an absolute score error into a relative probability
error (scores are O(3-10)).  Three distances are measured per encoder (normwise relative L2):
  * HIP vs the f32 oracle / the reference's golden vectors: what north_star's "1e-3" is about.  A single bf16-operand GEMM
    is already at 2.4e-3..5.3e-3 (patch embed, tools/vit_bisect.py), 12 layers land at 1e-2..2e-2; the bf16-rounding-aware
    oracle sits at the SAME distance from f32 at every sub-layer, i.e. this is the price of the prescribed
    operand dtype, not of the kernels.  Tolerances below are 1.3x the measured values (round 3: 2x).
  * HIP vs the oracle that rounds where the kernels round (emulate_bf16=True).
  * that oracle vs ITSELF with f64 instead of f32 accumulation (same rounding points, different last bits): the
    resolution of the emulation.  Values near a bf16 rounding boundary flip, the softmax amplifies the flips, and the two
    evaluations drift apart by 3e-4 per attention layer (tools/emu_sensitivity.py).  The HIP path must be as close to the
    emulating oracle as the emulating oracle is to itself (factor 1.5): that is the parity criterion with teeth.
Measured values are appended to gpurun_out/parity.jsonl so DESIGN.md can quote them.
"""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import check_summary, load_golden, rel_err, skip_param_init  # noqa: E402
from oracle import refcpu, synth  # noqa: E402

NODROP = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)  # parity = deterministic path
# (embedding vs f32 oracle / golden, worst trainable-gradient tensor vs f32 oracle / golden)
# (gpurun_out/parity.jsonl, copied into DESIGN.md 4)
# round 3 (bf16 residual stream + split-bf16 patch embedding), measured: dna_L2 7.2e-3 / 2.4e-2, dna_L12 1.5e-2 / 6.1e-2,
# txt_L4 4.5e-3 / 2.1e-2, vit_L2 9.1e-3 / 1.9e-2, vit_L12 1.7e-2 / 7.2e-2 (round 2: vit_L12 2.1e-2 / 9.3e-2, tolerance 4.3e-2 / 1.9e-1)
# round 4: the kernels are bitwise reproducible, so a box-to-box band is not needed: tolerances are 1.3 x the measured values
# (gpurun_out/parity.jsonl of the round-4 suite run: dna_L2 7.18e-3 / 2.27e-2, dna_L12 9.37e-3 / 6.55e-2, txt_L4 4.49e-3 / 2.21e-2,
# vit_L2 9.44e-3 / 2.06e-2, vit_L12 1.70e-2 / 6.84e-2); round 3 allowed 2 x
TOL = {"dna_L2": (9.4e-3, 3.0e-2), "dna_L12": (1.22e-2, 8.6e-2), "txt_L4": (5.9e-3, 2.9e-2),
       "vit_L2": (1.23e-2, 2.7e-2), "vit_L12": (2.22e-2, 8.9e-2)}
SELF_FACTOR = 1.5   # HIP-vs-emulating-oracle <= SELF_FACTOR x (emulating oracle f32-accumulate vs f64-accumulate), embedding AND gradients
# (round 5, gradients on the GPU box, profiles/r05_b_parity.jsonl: HIP vs emulation / emulation's own drift, worst tensor: dna_L2 1.37e-2 /
# 1.07e-2, dna_L12 5.49e-2 / 5.08e-2, txt_L4 1.39e-2 / 1.23e-2, vit_L2 1.42e-2 / 1.18e-2, vit_L12 6.03e-2 / 5.42e-2: ratios 1.08 - 1.29)


def _log(rec):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity.jsonl", "a") as f:
        f.write(json.dumps(rec) + "\n")


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _load(module, prefix, seed):
    sd = synth.synth_state_dict({prefix + k: v for k, v in synth.shapes_of(module).items()}, seed)
    module.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    return sd


def _oracle_grads(sd, fn):
    sd = {k: v.clone() for k, v in sd.items()}
    keys = [k for k in sd if refcpu.is_trainable_key(k) and sd[k].is_floating_point()]
    for k in keys:
        sd[k].requires_grad_(True)
    y = fn(sd)
    return sd, keys, y


_ORACLE_F32 = {}   # name -> (embedding, {key: gradient}) of the f32 oracle: the default-mode and the exact-mode test of one encoder share it


def _oracle_f32(name, sd, fn, w):
    if name not in _ORACLE_F32:
        sdo, keys, yo = _oracle_grads(sd, fn)
        (yo * w).sum().backward()
        _ORACLE_F32[name] = (yo.detach(), {k: sdo[k].grad for k in keys})
    return _ORACLE_F32[name]


def _f64(sd):
    return {k: (v.detach().double() if v.is_floating_point() else v) for k, v in sd.items()}


def _compare_encoder(name, module, prefix, sd, hip_in, oracle_fn, cot_key, gold, log_name=None):
    tol_emb, tol_grad = TOL[name]
    module.to("cuda")
    module.train()
    y = module(hip_in)
    w = synth.synth_tensor(cot_key, y.shape, seed=5)
    (y * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    yo, go = _oracle_f32(name, sd, oracle_fn, w)
    keys = list(go)
    sde, _, y_emu = _oracle_grads(sd, lambda s: oracle_fn(s, emulate=True))   # rounds where the kernels round (forward)
    (y_emu * w).sum().backward()
    # the same rounding points with f64 accumulation, forward AND backward: how far two equally valid evaluations of the rounded
    # network drift apart, in the embedding and in every gradient tensor
    sde64, _, y_emu64 = _oracle_grads(_f64(sd), lambda s: oracle_fn(s, emulate=True, f64=True))
    (y_emu64 * w.double()).sum().backward()
    y_emu64 = y_emu64.detach()
    e_f32, e_emu, e_self = rel_err(y, yo), rel_err(y, y_emu), rel_err(y_emu, y_emu64)
    rec = {"test": log_name or name, "emb_vs_f32_oracle": e_f32, "emb_vs_bf16_emulating_oracle": e_emu,
           "emulating_oracle_f32acc_vs_f64acc": e_self, "emulating_oracle_vs_f32_oracle": rel_err(y_emu, yo), "grads": {}}
    named = dict(module.named_parameters())
    worst = worst_emu = floor = self_g = 0.0
    for k in keys:
        p = named[k[len(prefix):]]
        assert p.grad is not None, k
        e = rel_err(p.grad, go[k])
        rec["grads"][k] = e
        worst = max(worst, e)
        worst_emu = max(worst_emu, rel_err(p.grad, sde[k].grad))
        floor = max(floor, rel_err(sde[k].grad, go[k]))
        self_g = max(self_g, rel_err(sde[k].grad, sde64[k].grad))
    rec["worst_grad"] = worst
    rec["worst_grad_vs_emulating_oracle"] = worst_emu
    rec["worst_grad_emulating_vs_f32_oracle"] = floor   # what rounding the FORWARD operands alone does to the gradients
    rec["worst_grad_emulating_oracle_f32acc_vs_f64acc"] = self_g
    _log(rec)
    assert torch.isfinite(y).all()
    assert e_emu < max(SELF_FACTOR * e_self, 5e-4), rec
    # the gradient criterion with teeth: the hand-written BACKWARD (bf16 dX operands, recomputed attention
    # probabilities, 8-bit gelu' codes) must sit as close to the emulating oracle's autograd as two evaluations of that oracle sit
    # to each other -- a backward regression inside the calibrated f32 band below fails here
    assert worst_emu < max(SELF_FACTOR * self_g, 2e-3), rec
    assert e_f32 < tol_emb, rec
    assert worst < tol_grad, rec
    if gold is not None:  # fixtures from the imported reference
        check_summary(gold[0], y, gold[1]["out"], tol_emb, what=name + " ")
        for k in keys:
            check_summary(k, named[k[len(prefix):]].grad, gold[1]["grads"][k], tol_grad, what=name + " ")


@pytest.mark.parametrize("layers", [2, 12])
def test_dna_encoder(layers):
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    with skip_param_init():
        m = LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=layers, **NODROP)), r=4,
                              num_classes=768)
    sd = _load(m, "dna_encoder.", 11)
    _, dna, _, _ = synth.synth_batch(2, seed=21)
    fn = lambda s, emulate=False, f64=False: refcpu.barcode_bert_encoder(s, dna, emulate_bf16=emulate)
    _compare_encoder(f"dna_L{layers}", m, "dna_encoder.", sd, dna.cuda(), fn, f"dna.cot.{layers}",
                     (f"dna.out.{layers}", load_golden("encoders")[f"dna_L{layers}"]))


def test_text_encoder():
    from bioscanclip.model import arch
    from bioscanclip.model.language_encoder import LoRA_bert
    m = LoRA_bert(arch.BertModelParams(arch.bert_small_config(**NODROP)), r=4, num_classes=768)
    sd = _load(m, "language_encoder.", 12)
    _, _, text, _ = synth.synth_batch(4, seed=22, with_text=True)
    fn = lambda s, emulate=False, f64=False: refcpu.bert_text_encoder(s, text, emulate_bf16=emulate)
    _compare_encoder("txt_L4", m, "language_encoder.", sd, {k: v.cuda() for k, v in text.items()}, fn, "txt.cot",
                     ("txt.out", load_golden("encoders")["txt_L4"]))


@pytest.mark.parametrize("depth", [2, 12])
def test_vit_encoder(depth):
    from bioscanclip.model import arch
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    with skip_param_init():
        m = LoRA_ViT_timm(arch.VisionTransformerParams(depth=depth), r=4, num_classes=768)
    sd = _load(m, "image_encoder.", 13)
    image, _, _, _ = synth.synth_batch(2, seed=23)
    fn = lambda s, emulate=False, f64=False: refcpu.vit_encoder(s, image.double() if f64 else image, emulate_bf16=emulate)
    _compare_encoder(f"vit_L{depth}", m, "image_encoder.", sd, image.cuda(), fn, f"vit.cot.{depth}",
                     (f"vit.out.{depth}", load_golden("encoders")[f"vit_L{depth}"]))


@pytest.mark.parametrize("which", ["vit", "dna"])
def test_parity_mode_against_the_f32_oracle(which, monkeypatch):
    """BSCLIP_PARITY=1 (hip/engine.py set_parity_mode): f32 residual and residual-gradient streams, gated at 1.3 x its own measured
    distance to the f32 oracle at depth 12 (round 4: ViT 1.22e-2 / worst gradient 4.46e-2 -- closer than the default's 1.70e-2 /
    6.84e-2; BarcodeBERT 1.25e-2 / 3.89e-2 -- the embedding is NOT closer than the default's 0.94e-2, the gradients are: at this
    level the distance is the bf16 GEMM / attention operands', which both configurations share, not the streams').
    bench.py prices the mode (`parity_mode_ms_per_step`)."""
    from bioscanclip.hip import engine
    from bioscanclip.model import arch
    monkeypatch.setattr(engine, "RESID_STREAM_BF16", False)
    monkeypatch.setattr(engine, "GRAD_STREAM_BF16", False)
    if which == "vit":
        from bioscanclip.model.image_encoder import LoRA_ViT_timm
        m, prefix, seed = LoRA_ViT_timm(arch.VisionTransformerParams(depth=12), r=4, num_classes=768), "image_encoder.", 13
        x, _, _, _ = synth.synth_batch(2, seed=23)
        fn, cot, name = (lambda s: refcpu.vit_encoder(s, x)), "vit.cot.12", "vit_L12"
    else:
        from bioscanclip.model.dna_encoder import LoRA_barcode_bert
        m = LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=12, **NODROP)), r=4, num_classes=768)
        prefix, seed = "dna_encoder.", 11
        _, x, _, _ = synth.synth_batch(2, seed=21)
        fn, cot, name = (lambda s: refcpu.barcode_bert_encoder(s, x)), "dna.cot.12", "dna_L12"
    sd = _load(m, prefix, seed)
    m.to("cuda").train()
    y = m(x.cuda())
    w = synth.synth_tensor(cot, y.shape, seed=5)
    (y * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    ws = m._engine.ws
    assert ws["resid_bf16"] is False and ws["grad_bf16"] is False
    yo, go = _oracle_f32(name, sd, fn, w)
    named = dict(m.named_parameters())
    e = rel_err(y, yo)
    worst = max(rel_err(named[k[len(prefix):]].grad, go[k]) for k in go)
    _log({"test": f"parity_mode_{name}", "emb_vs_f32_oracle": e, "worst_grad": worst})
    tol_e, tol_g = {"vit_L12": (1.59e-2, 5.8e-2), "dna_L12": (1.62e-2, 5.1e-2)}[name]
    assert e < tol_e and worst < tol_g, (e, worst)
    assert worst < TOL[name][1] / 1.3, worst    # in both towers the gradients are closer than the default configuration's


@pytest.mark.parametrize("which", ["vit_L12", "dna_L12", "txt_L4", "vit_L2"])
def test_exact_forward_meets_north_star_tolerance(which, monkeypatch):
    """BSCLIP_PARITY=2 (hip/engine.py EXACT_FORWARD, csrc/exact.hip): every trunk GEMM on split-bf16 operands (hi + lo, K tripled),
    LoRA folded in f32, exact-erf GELU, f32 attention, f32 streams.  north_star: "outputs (embeddings ...) match the reference CPU
    path within 1e-3" -- here the embeddings of the full-depth encoders against the f32 oracle AND the golden fixtures the imported
    reference produced, at 1e-3 (measured: see the log line / DESIGN.md 4).  The backward is exact as well (f32 gradients, dX / dW
    GEMMs on split operands, f32 attention backward and LoRA gradients): EVERY trainable tensor's gradient within 1e-3 of the
    oracle's autograd."""
    from bioscanclip.hip import engine
    from bioscanclip.model import arch
    monkeypatch.setattr(engine, "RESID_STREAM_BF16", False)
    monkeypatch.setattr(engine, "GRAD_STREAM_BF16", False)
    monkeypatch.setattr(engine, "EXACT_FORWARD", True)
    gold_all = load_golden("encoders")
    if which.startswith("vit"):
        depth = int(which[5:])
        from bioscanclip.model.image_encoder import LoRA_ViT_timm
        m, prefix, seed = LoRA_ViT_timm(arch.VisionTransformerParams(depth=depth), r=4, num_classes=768), "image_encoder.", 13
        x, _, _, _ = synth.synth_batch(2, seed=23)
        fn, cot, hip_in, gkey = (lambda s: refcpu.vit_encoder(s, x)), f"vit.cot.{depth}", x.cuda(), f"vit.out.{depth}"
    elif which == "dna_L12":
        from bioscanclip.model.dna_encoder import LoRA_barcode_bert
        m = LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=12, **NODROP)), r=4, num_classes=768)
        prefix, seed = "dna_encoder.", 11
        _, x, _, _ = synth.synth_batch(2, seed=21)
        fn, cot, hip_in, gkey = (lambda s: refcpu.barcode_bert_encoder(s, x)), "dna.cot.12", x.cuda(), "dna.out.12"
    else:
        from bioscanclip.model.language_encoder import LoRA_bert
        m = LoRA_bert(arch.BertModelParams(arch.bert_small_config(**NODROP)), r=4, num_classes=768)
        prefix, seed = "language_encoder.", 12
        _, _, text, _ = synth.synth_batch(4, seed=22, with_text=True)
        fn, cot, hip_in, gkey = (lambda s: refcpu.bert_text_encoder(s, text)), "txt.cot", {k: v.cuda() for k, v in text.items()}, "txt.out"
    sd = _load(m, prefix, seed)
    m.to("cuda").train()
    y = m(hip_in)
    assert m._engine.exact()
    w = synth.synth_tensor(cot, y.shape, seed=5)
    (y * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    yo, go = _oracle_f32(which, sd, fn, w)
    keys = list(go)
    named = dict(m.named_parameters())
    e = rel_err(y, yo)
    worst = max(rel_err(named[k[len(prefix):]].grad, go[k]) for k in keys)
    _log({"test": f"exact_forward_{which}", "emb_vs_f32_oracle": e, "worst_grad": worst})
    assert e < 1e-3, e                                                   # north_star's tolerance, on the embeddings
    check_summary(gkey, y, gold_all[which]["out"], 1e-3, what=which + " exact ")   # ... and against the imported reference's own output
    assert worst < 1e-3, worst                                           # ... and on every gradient tensor
    for k in keys:                                                       # ... which the imported reference's autograd confirms
        check_summary(k, named[k[len(prefix):]].grad, gold_all[which]["grads"][k], 1e-3, what=which + " exact ")


def _build_clip(*a, **k):
    from helpers import skip_param_init
    with skip_param_init():   # every tensor is loaded from oracle.synth right after construction
        return _build_clip_inner(*a, **k)


def _build_clip_inner(with_text, seed):
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    from bioscanclip.model.language_encoder import LoRA_bert
    from bioscanclip.model.simple_clip import SimpleCLIP
    img = LoRA_ViT_timm(arch.vit_base_patch16_224(), r=4, num_classes=768)
    dna = LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(**NODROP)), r=4, num_classes=768)
    txt = LoRA_bert(arch.BertModelParams(arch.bert_small_config(**NODROP)), r=4, num_classes=768) if with_text else None
    model = SimpleCLIP(img, dna, txt)
    sd = _load(model, "", seed)
    return model, sd


# trajectory tolerances (2x measured, gpurun_out/parity.jsonl): first-step embeddings / gradient fingerprints vs the
# reference's, loss per step (relative), trainable parameters after the last step
# loss: the golden loss falls 50x over the run and every step amplifies the difference of the step before (measured per step,
# two equally exact GELU tables: 0.0005 ... 0.013 in profiles/r03_h_parity.jsonl, 0.0009 ... 0.034 after round 3's table change),
# so the late steps bound the tolerance, not the arithmetic of one step
TRAJ_TOL = {False: dict(emb=1.4e-2, grad=0.12, loss=6e-2, params=6e-2), True: dict(emb=2e-2, grad=0.15, loss=6e-2, params=0.1)}


@pytest.mark.parametrize("with_text", [False, True])
def test_training_trajectory_matches_reference(with_text):
    """BASELINE config 1 (I+D, B=8, 10 steps over a two-batch epoch) and a 6-step I+D+T run: loss per step and trainable
    parameters after the last step vs the trajectory the imported reference produced with torch.optim.AdamW.  The golden
    loss falls from 2.24 (> ln 8) to 0.044: only the right embeddings, loss, gradients and optimizer reproduce it."""
    _run_trajectory(with_text, TRAJ_TOL[with_text], f"trajectory text={with_text}")


# the same trajectories in the exact mode: north_star's "outputs (embeddings, loss, gradients) match the reference CPU path within
# 1e-3" on the first step's embeddings and gradient fingerprints and on the loss of EVERY step (the late steps carry whatever the
# early ones differed by, amplified by a loss that falls 50 x; measured values in gpurun_out/parity.jsonl)
TRAJ_TOL_EXACT = dict(emb=1e-3, grad=1e-3, loss=1e-3, params=1e-3)


@pytest.mark.parametrize("with_text", [False, True])
def test_training_trajectory_exact_mode_within_north_star_tolerance(with_text, monkeypatch):
    """BSCLIP_PARITY=2, whole training steps: configs[0] (I+D, B=8, 10 steps) and the 6-step I+D+T run against the trajectory the
    imported reference produced -- embeddings, gradients, per-step loss and the parameters after the last AdamW step within 1e-3."""
    from bioscanclip.hip import engine
    monkeypatch.setattr(engine, "RESID_STREAM_BF16", False)
    monkeypatch.setattr(engine, "GRAD_STREAM_BF16", False)
    monkeypatch.setattr(engine, "EXACT_FORWARD", True)
    _run_trajectory(with_text, TRAJ_TOL_EXACT, f"trajectory_exact text={with_text}")


def _run_trajectory(with_text, tol, log_name):
    from helpers import summary_distance
    from bioscanclip.hip.optim import FusedAdamW
    from bioscanclip.model.loss_func import ContrastiveLoss
    g = load_golden("trajectory_idt" if with_text else "trajectory_id")
    assert g["losses"][-1] < 0.6 * g["losses"][0] and abs(g["losses"][0] - torch.tensor(float(g["B"])).log().item()) > 5e-3
    model, sd = _build_clip(with_text, g["weight_seed"])
    model.to("cuda").train()
    opt = FusedAdamW(model.parameters(), lr=g["lr"])
    crit = ContrastiveLoss(criterion=torch.nn.CrossEntropyLoss(), logit_scale=1 / 0.07)
    losses = []
    rec = {"test": log_name}
    for s in range(g["steps"]):
        image, dna, text, label = synth.synth_batch(g["B"], seed=g["batch_seed0"] + s % g["n_batches"], with_text=with_text)
        opt.zero_grad()
        text = None if text is None else {k: v.cuda() for k, v in text.items()}
        io, do, to = model(image.cuda(), dna.cuda(), text)
        loss = crit(io, do, to, label.cuda())
        loss.backward()
        if s == 0:
            named = dict(model.named_parameters())
            rec["emb"] = {"img": summary_distance("traj.img", io, g["first_step"]["image_out"]),
                          "dna": summary_distance("traj.dna", do, g["first_step"]["dna_out"])}
            if with_text:
                rec["emb"]["txt"] = summary_distance("traj.txt", to, g["first_step"]["text_out"])
            gd = {k: summary_distance(k, named[k].grad, gs) for k, gs in g["first_step"]["grads"].items()}
            rec["first_step_grad_worst"] = {m: max(d[m] for d in gd.values()) for m in ("norm", "probe", "first")}
            rec["first_step_grad_worst_key"] = max(gd, key=lambda k: max(gd[k].values()))
            opt.attach(model)
        opt.step()
        losses.append(loss.item())
    rec["losses"], rec["ref"] = losses, g["losses"]
    rec["loss_rel_err"] = [abs(a - b) / abs(b) for a, b in zip(losses, g["losses"])]
    named = dict(model.named_parameters())
    pd = {k: summary_distance(k, named[k], gs) for k, gs in g["params_after"].items()}
    rec["params_after_worst"] = {m: max(d[m] for d in pd.values()) for m in ("norm", "probe", "first")}
    _log(rec)
    assert max(max(d.values()) for d in rec["emb"].values()) < tol["emb"], rec
    assert max(rec["first_step_grad_worst"].values()) < tol["grad"], rec
    assert max(rec["loss_rel_err"]) < tol["loss"], rec
    # AdamW divides by sqrt(v): on near-zero gradient elements a small absolute error flips the update's sign, so
    # parameters agree with the f32 reference to a fraction of lr*steps per element, not to bf16 epsilon.
    assert max(rec["params_after_worst"].values()) < tol["params"], rec


def test_tower_streams_join_before_gradients_are_read():
    """The towers run forward/backward on their own HIP streams; the gradients they write from inside the kernels must be
    complete on the caller's stream when backward() returns (autograd does not know about them).  Kernels are deterministic,
    so the flat gradients must equal the single-stream run bit for bit."""
    import bioscanclip.model.simple_clip as sc
    from bioscanclip.model.loss_func import ContrastiveLoss
    model, _ = _build_clip(False, 77)
    model.to("cuda").train()
    crit = ContrastiveLoss(criterion=torch.nn.CrossEntropyLoss(), logit_scale=1 / 0.07)
    image, dna, _, label = synth.synth_batch(16, seed=5)
    image, dna, label = image.cuda(), dna.cuda(), label.cuda()
    grads = {}
    saved = sc._TOWER_STREAMS
    try:
        for mode in (True, False, True):
            sc._TOWER_STREAMS = mode
            for p in model.parameters():
                if p.grad is not None:
                    p.grad.zero_()
            io, do, to = model(image, dna, None)
            crit(io, do, to, label).backward()
            # read on the caller's stream right away, without a device-wide synchronize
            got = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None]).clone()
            grads.setdefault(mode, []).append(got)
    finally:
        sc._TOWER_STREAMS = saved
    torch.cuda.synchronize()
    for g in grads[True]:
        assert torch.equal(g, grads[False][0])


def test_checkpoint_loaded_after_first_forward_repacks_frozen_weights():
    """The engines pack the frozen weights into their own bf16 layouts at the first forward.  Loading a checkpoint
    afterwards (in place) must not leave those copies stale."""
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert

    def build(seed):
        m = LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=2, **NODROP)), r=4,
                              num_classes=768)
        _load(m, "dna_encoder.", seed)
        return m.to("cuda").eval()

    _, dna, _, _ = synth.synth_batch(4, seed=3)
    dna = dna.cuda()
    a, b = build(21), build(22)
    with torch.no_grad():
        ya, yb = a(dna).clone(), b(dna).clone()
        assert not torch.allclose(ya, yb)
        b.load_state_dict(a.state_dict())  # in place, after b's engine exists
        assert torch.equal(b(dna), ya)
        fresh = build(22)
        fresh.load_state_dict({k: v.cpu() for k, v in a.state_dict().items()})  # before any forward
        assert torch.equal(fresh(dna), ya)


def test_train_cl_epoch_with_eval_phase(tmp_path, capsys):
    """scripts/train_cl.py end to end on synthetic data: one short epoch, checkpoint written with the reference's key
    names, eval_phase (feature extraction + GPU top-k retrieval + accuracy table) runs natively."""
    import sys as _sys
    scripts = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bioscan-clip_amd", "scripts")
    _sys.path.insert(0, scripts)
    import train_cl
    losses = train_cl.main(["model_config=lora_vit_lora_barcode_bert_ssl", "model_config.batch_size=8",
                            "model_config.epochs=1", "synthetic_steps_per_epoch=2", "synthetic_eval=true",
                            "synthetic_eval_batches=1", "save_ckpt=true", "debug_flag=false",
                            f"project_root_path={tmp_path}"])
    out = capsys.readouterr().out
    assert len(losses) == 1 and "overall_acc" in out and "Query_feature: encoded_image_feature" in out
    ck = [os.path.join(r, f) for r, _, fs in os.walk(str(tmp_path)) for f in fs if f.endswith(".pth")]
    assert any(p.endswith("last.pth") for p in ck) and any(p.endswith("best.pth") for p in ck)
    keys = set(torch.load(ck[0], map_location="cpu").keys())
    assert keys == {k for k in load_golden("state_dict_keys")["keys"] if not k.startswith("language_encoder.")}


def test_inference_and_eval_script_loads_the_training_checkpoint(tmp_path, capsys):
    """scripts/inference_and_eval.py as an entry point (reference :786-871): the checkpoint train_cl.py wrote is loaded (checked),
    features of the three splits are extracted, the accuracy table is printed; ``save_inference`` / ``load_inference`` reuse the
    cached features and give the same table; ``model_config.load_ckpt=false`` skips the checkpoint; a checkpoint of another
    configuration is refused."""
    import sys as _sys
    scripts = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bioscan-clip_amd", "scripts")
    _sys.path.insert(0, scripts)
    import inference_and_eval
    import train_cl
    common = ["model_config=lora_vit_lora_barcode_bert_ssl", f"project_root_path={tmp_path}", "debug_flag=false"]
    train_cl.main(common + ["model_config.batch_size=8", "model_config.epochs=1", "synthetic_steps_per_epoch=2", "save_ckpt=true"])
    ck = [os.path.join(r, f) for r, _, fs in os.walk(str(tmp_path)) for f in fs if f.endswith("last.pth")][0]
    capsys.readouterr()
    acc, _, _ = inference_and_eval.main(common + [f"model_config.ckpt_path={ck}", "save_inference=true"])
    out1 = capsys.readouterr().out
    assert "Initialize model" in out1 and "micro_acc top-1" in out1 and "macro_acc top-5" in out1
    a1 = acc["encoded_image_feature"]["encoded_dna_feature"]["seen"]["micro_acc"]
    acc2, _, _ = inference_and_eval.main(common + [f"model_config.ckpt_path={ck}", "load_inference=true"])
    out2 = capsys.readouterr().out
    assert "Initialize model" not in out2                         # features came from the cache
    assert acc2["encoded_image_feature"]["encoded_dna_feature"]["seen"]["micro_acc"] == a1
    inference_and_eval.main(common + ["model_config.load_ckpt=false"])
    with pytest.raises(RuntimeError, match="does not match the parameter tree"):
        inference_and_eval.main(["model_config=lora_vit_lora_bert_ssl", f"project_root_path={tmp_path}", f"model_config.ckpt_path={ck}"])


def test_train_cl_image_text_config(tmp_path, capsys):
    """The reference's two-tower Image+Text configuration (model_config/lora_vit_lora_bert_ssl.yaml): no DNA tower is built, the
    loss runs on the one pair, the checkpoint carries image and language keys only."""
    import sys as _sys
    scripts = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bioscan-clip_amd", "scripts")
    _sys.path.insert(0, scripts)
    import train_cl
    losses = train_cl.main(["model_config=lora_vit_lora_bert_ssl", "model_config.batch_size=8", "model_config.epochs=1",
                            "synthetic_steps_per_epoch=3", "save_ckpt=true", "debug_flag=false", f"project_root_path={tmp_path}"])
    capsys.readouterr()
    assert len(losses) == 1 and losses[0] == losses[0]
    ck = [os.path.join(r, f) for r, _, fs in os.walk(str(tmp_path)) for f in fs if f.endswith("last.pth")]
    keys = set(torch.load(ck[0], map_location="cpu").keys())
    assert keys and not any(k.startswith("dna_encoder.") for k in keys)
    assert any(k.startswith("image_encoder.") for k in keys) and any(k.startswith("language_encoder.") for k in keys)


def test_training_on_a_fixed_batch_drives_the_loss_down():
    """End-to-end sanity beyond step-wise parity: 60 AdamW steps on one fixed synthetic batch (dropout at the HF defaults, i.e.
    the benchmark configuration) must overfit it -- the loss falls well below its start and no gradient or activation turns
    non-finite; the flat buffers and workspaces are reused from step to step (allocated memory stays put)."""
    from bioscanclip.hip.optim import FusedAdamW
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    from bioscanclip.model.loss_func import ContrastiveLoss
    from bioscanclip.model.simple_clip import SimpleCLIP
    import gc
    gc.collect()   # garbage of earlier tests (engines in reference cycles) must not be freed in the middle of the memory samples
    torch.manual_seed(5)
    model = SimpleCLIP(LoRA_ViT_timm(arch.VisionTransformerParams(depth=4), r=4, num_classes=768),
                       LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=4)), r=4,
                                         num_classes=768), None).to("cuda").train()
    image, dna, _, label = synth.synth_batch(32, seed=77)
    image, dna, label = image.cuda(), dna.cuda(), label.cuda()
    crit = ContrastiveLoss(criterion=torch.nn.CrossEntropyLoss(), logit_scale=1 / 0.07)
    opt = FusedAdamW(model.parameters(), lr=1e-3)
    losses, mem = [], []
    for s in range(60):
        opt.zero_grad()
        loss = crit(*model(image, dna, None), label)
        loss.backward()
        if s == 0:
            opt.attach(model)
        opt.step()
        if s % 10 == 0 or s == 59:
            losses.append(loss.item())
            mem.append(torch.cuda.memory_allocated())
    _log({"test": "fixed-batch training", "losses": losses})
    assert all(l == l for l in losses) and losses[-1] < 0.5 * losses[0], losses
    assert max(mem[1:]) - mem[1] < 64 * 2 ** 20, mem   # no growth from step to step (a collector run may only lower it)
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_backward_after_a_later_forward_is_refused():
    """An engine holds one workspace of saved activations per encoder.  forward(a), forward(b), backward(a) would silently
    differentiate b's activations; it raises instead.  backward(b) still works, and so does the usual one-forward-one-backward."""
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    m = LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=2, **NODROP)), r=4, num_classes=768)
    _load(m, "dna_encoder.", 11)
    m.to("cuda").train()
    a = synth.synth_batch(2, seed=21)[1].cuda()
    b = synth.synth_batch(2, seed=22)[1].cuda()
    ya = m(a)
    yb = m(b)
    with pytest.raises(RuntimeError, match="overwritten by a later forward"):
        ya.sum().backward()
    yb.sum().backward()
    ya = m(a)
    with torch.no_grad():
        m(b)                                   # an evaluation pass in between overwrites the workspace just the same
    with pytest.raises(RuntimeError, match="overwritten by a later forward"):
        ya.sum().backward()


def test_requires_gpu_inputs():
    from bioscanclip.model import arch
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    m = LoRA_ViT_timm(arch.VisionTransformerParams(depth=1), r=4, num_classes=768)
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(1, 3, 224, 224))


def test_detect_anomaly_names_where_non_finite_values_appear(monkeypatch):
    """BSCLIP_DETECT_ANOMALY (hip/engine.py): the counterpart of the reference loop's ``torch.autograd.set_detect_anomaly(True)``
    (train_epoch.py:12).  A NaN planted in block 1's LoRA-A makes the forward raise and name block 1's QKV output (block 0's
    tensors are clean); an infinite cotangent makes the backward raise and name gradient tensors; clean runs pass untouched."""
    from bioscanclip.hip import engine, ops
    from bioscanclip.model import arch
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    c = torch.zeros(1, dtype=torch.int32, device="cuda")
    t = torch.randn(100_003, device="cuda")
    t[5], t[77_777], t[100_002] = float("nan"), float("inf"), float("-inf")
    ops.count_nonfinite(t, c)
    ops.count_nonfinite(t.bfloat16(), c)
    ops.count_nonfinite(torch.randn(4096, device="cuda"), c)
    assert c.item() == 6
    monkeypatch.setattr(engine, "DETECT_ANOMALY", True)
    m = LoRA_ViT_timm(arch.VisionTransformerParams(depth=3), r=4, num_classes=768)
    _load(m, "image_encoder.", 13)
    m.to("cuda").train()
    x = synth.synth_batch(2, seed=23)[0].cuda()
    y = m(x)
    (y * torch.randn_like(y)).sum().backward()          # clean: nothing raised
    with pytest.raises(RuntimeError, match="backward: non-finite gradient values in .*weight"):
        y = m(x)
        (y * float("inf")).sum().backward()
    with torch.no_grad():
        m.lora_vit.blocks[1].attn.qkv.linear_a_q.weight[0, 0] = float("nan")
    with pytest.raises(RuntimeError, match=r"ViTEngine forward: non-finite values first appear in blocks\.1\.attn\.qkv output"):
        m(x)
    # the BERT engines (post-LN trunk, MLM softmax-mean head), in the default and in the exact mode
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    for exact in (False, True):
        monkeypatch.setattr(engine, "RESID_STREAM_BF16", not exact)
        monkeypatch.setattr(engine, "GRAD_STREAM_BF16", not exact)
        monkeypatch.setattr(engine, "EXACT_FORWARD", exact)
        d = LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=3, **NODROP)), r=4, num_classes=768)
        _load(d, "dna_encoder.", 11)
        d.to("cuda").train()
        ids = synth.synth_batch(2, seed=21)[1].cuda()
        y = d(ids)
        (y * torch.randn_like(y)).sum().backward()      # clean
        with torch.no_grad():
            d.lora_barcode_bert.bert.encoder.layer[1].attention.self.value.w_a.weight[0, 0] = float("inf")
        with pytest.raises(RuntimeError, match=r"BertEngine forward: non-finite values first appear in encoder\.layer\.1\.attention\.self q / k / v"):
            d(ids)


@pytest.mark.parametrize("exact", [False, True])
def test_train_epoch_entry_point_reproduces_the_reference_trajectory(exact, monkeypatch):
    """``bioscanclip.epoch.train_epoch.train_epoch`` ITSELF (reference signature, the captured-graph launch path it
    takes by default, the loss read one step late) over configs[0] -- five epochs of the two-batch loader = the golden run's ten
    steps: the epoch means of the loss and the trainable parameters after the last epoch against the imported reference's
    trajectory, at the trajectory tolerances of the default mode and at north_star's 1e-3 in the exact mode."""
    from helpers import summary_distance
    from bioscanclip.epoch.train_epoch import train_epoch
    from bioscanclip.hip import engine
    from bioscanclip.hip.optim import FusedAdamW
    from bioscanclip.model.loss_func import ContrastiveLoss
    if exact:
        monkeypatch.setattr(engine, "RESID_STREAM_BF16", False)
        monkeypatch.setattr(engine, "GRAD_STREAM_BF16", False)
        monkeypatch.setattr(engine, "EXACT_FORWARD", True)
    g = load_golden("trajectory_id")
    tol = TRAJ_TOL_EXACT if exact else TRAJ_TOL[False]
    model, _ = _build_clip(False, g["weight_seed"])
    model.to("cuda")
    opt = FusedAdamW(model.parameters(), lr=g["lr"])
    crit = ContrastiveLoss(criterion=torch.nn.CrossEntropyLoss(), logit_scale=1 / 0.07)
    loader = []
    for i in range(g["n_batches"]):   # the reference's 7-tuple: (processid, image, dna, input_ids, token_type_ids, attention_mask, label)
        image, dna, _, label = synth.synth_batch(g["B"], seed=g["batch_seed0"] + i)
        loader.append((None, image, dna, None, None, None, label))
    epochs = g["steps"] // g["n_batches"]
    means = [train_epoch(False, epochs, e, loader, model, opt, crit, "cuda", scheduler=None, rank=0) for e in range(epochs)]
    torch.cuda.synchronize()
    ref = [sum(g["losses"][e * g["n_batches"]:(e + 1) * g["n_batches"]]) / g["n_batches"] for e in range(epochs)]
    err = [abs(a - b) / abs(b) for a, b in zip(means, ref)]
    named = dict(model.named_parameters())
    pd = {k: summary_distance(k, named[k], gs) for k, gs in g["params_after"].items()}
    worst = max(max(d.values()) for d in pd.values())
    _log({"test": f"train_epoch_trajectory exact={exact}", "epoch_mean_loss": means, "ref": ref, "loss_rel_err": err, "params_after_worst": worst})
    assert getattr(model, "_bsclip_graphed", None) is not None and model.image_encoder._engine.exact() == exact   # the captured path ran
    assert max(err) < tol["loss"] and worst < tol["params"], (err, worst)
