"""fp8 GEMM family (BASELINE.json configs[4]) against exact emulation: the products of two e4m3 values are exact in f32, so the
kernel must agree with a float matmul of the DEQUANTISED operands to accumulation-order precision -- for both MFMA forms."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import rel_err  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bioscanclip.hip import ops as o
    o.init_tables()
    return o


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def gelu(x):
    return torch.nn.functional.gelu(x)


@pytest.mark.parametrize("form", [1, 2])
@pytest.mark.parametrize("M,N,K,aug", [(300, 256, 128, False), (1000, 768, 768, True), (2600, 2304, 768, True),
                                       (5120, 768, 3072, False), (197 * 16, 3072, 768, False)])
def test_gemm_fp8_matches_dequantised_matmul(ops, form, M, N, K, aug):
    from bioscanclip.hip.lib import EPI_BF16, EPI_F32, EPI_GELU_FP8, EPI_RESID_F32
    FP8 = ops.FP8
    a = rnd(M, K, seed=1).cuda()
    a8 = a.to(FP8)                                            # activations: scale 1 (LayerNorm / GELU outputs are O(1))
    w = rnd(N, K, seed=2, scale=0.05).cuda()
    w8, ws = ops.quantize_rows_fp8(w.contiguous())
    # the quantiser itself: row amax -> 448, round to nearest e4m3
    ref_s = w.abs().amax(dim=1) / 448
    assert torch.allclose(ws, ref_s, rtol=1e-6)
    assert torch.equal(w8.view(torch.uint8), (w / ref_s[:, None]).to(FP8).view(torch.uint8))
    bias = rnd(N, seed=3).cuda()
    alpha = ws.clone()                                        # activation scale 1 x weight-row scale
    ref = (a8.float() @ w8.float().t()) * alpha + bias
    kw = {}
    if aug:
        t = torch.zeros(M, 64, dtype=torch.bfloat16, device="cuda")
        t[:, :8] = rnd(M, 8, seed=4, scale=0.3).cuda().bfloat16()
        b = torch.zeros(N, 64, dtype=torch.bfloat16, device="cuda")
        b[:, :8] = (rnd(N, 8, seed=5, scale=0.05).cuda() / alpha[:, None]).bfloat16()
        ref = ref + (t.float() @ b.float().t()) * alpha
        kw = dict(a_aug=t, b_aug=b)
    out32 = torch.full((M + 3, N), float("nan"), device="cuda")
    ops.gemm_fp8(a8, w8, out32, alpha, bias, EPI_F32, M=M, form=form, **kw)
    assert rel_err(out32[:M], ref) < 2e-5
    assert torch.isnan(out32[M:]).all()
    out16 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_fp8(a8, w8, out16, alpha, bias, EPI_BF16, form=form, **kw)
    assert rel_err(out16.float(), ref) < 4e-3
    r = rnd(M, N, seed=6).cuda()
    ops.gemm_fp8(a8, w8, out32, alpha, bias, EPI_RESID_F32, resid=r, M=M, form=form, **kw)
    assert rel_err(out32[:M], ref + r) < 2e-5
    out8 = torch.empty(M, N, dtype=FP8, device="cuda")
    z = torch.empty(M, N, dtype=torch.uint8, device="cuda")
    ops.gemm_fp8(a8, w8, out8, alpha, bias, EPI_GELU_FP8, aux=z, form=form, **kw)
    want = gelu(ref)
    # e4m3 has 3 mantissa bits: compare codes -- equal, or one code apart where the value sits on a rounding boundary
    got, exp = out8.float(), want.to(FP8).float()
    assert ((got - exp).abs() <= 0.126 * exp.abs() + 2e-3).all()
    assert (got != exp).float().mean().item() < 2e-3
    x = ref.double()
    dg = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
    assert ((z.float() * (1.26 / 255) - 0.13) - dg.float()).abs().max().item() < 0.5 * 1.26 / 255 + 6e-4


def test_gemm_fp8_forms_agree_and_integer_exact(ops):
    """Small integers are exact in e4m3 and their dot products exact in f32: both MFMA forms must return the integer matrix
    product exactly (catches any k-slot mismatch between the A and B fragments)."""
    from bioscanclip.hip.lib import EPI_F32
    FP8 = ops.FP8
    g = torch.Generator().manual_seed(7)
    M, N, K = 512, 256, 384
    a = torch.randint(-4, 5, (M, K), generator=g).float().cuda()
    w = torch.randint(-3, 4, (N, K), generator=g).float().cuda()
    w[:, ::7] = 0          # asymmetric structure along k
    ref = a @ w.t()
    one, zero = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    for form in (1, 2):
        out = torch.empty(M, N, device="cuda")
        ops.gemm_fp8(a.to(FP8), w.to(FP8), out, one, zero, EPI_F32, form=form)
        assert torch.equal(out, ref), form


# ------------------------------------------------------------------------------------------------- encoders in fp8 mode
NODROP = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
# Parity gate of configs[4], as measured on MI355X (gpurun_out/parity.jsonl "fp8 ..."), tolerances <= 2x measured.  e4m3 has 3
# mantissa bits (relative rounding 2^-4), so a GEMM output carries ~4 % noise per operand pair; LoRA, attention, the residual
# stream, every backward GEMM and the loss stay in bf16 / f32.  The gate is defined against this repo's bf16 path (and
# reported against the f32 oracle): embeddings by relative L2 error and cosine, gradients by relative L2 error.
FP8_TOL = {"vit": dict(emb=0.3, cos=0.97, grad=0.9), "dna": dict(emb=0.14, cos=0.99, grad=0.5)}


def _log(rec):
    import json
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity.jsonl", "a") as f:
        f.write(json.dumps(rec) + "\n")


@pytest.mark.parametrize("which", ["vit", "dna"])
def test_fp8_encoder_tracks_bf16_encoder(ops, which):
    from bioscanclip.hip.engine import set_precision
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    from oracle import refcpu, synth
    depth = 6
    if which == "vit":
        m, pre = LoRA_ViT_timm(arch.VisionTransformerParams(depth=depth), r=4, num_classes=768), "image_encoder."
    else:
        m = LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=depth, **NODROP)), r=4,
                              num_classes=768)
        pre = "dna_encoder."
    sd = synth.synth_state_dict({pre + k: v for k, v in synth.shapes_of(m).items()}, 17)
    m.load_state_dict({k[len(pre):]: v for k, v in sd.items()})
    m.cuda().train()
    image, dna, _, _ = synth.synth_batch(8, seed=29)
    x = (image if which == "vit" else dna).cuda()
    w = synth.synth_tensor("fp8.cot", (8, 768), seed=5).cuda()
    res = {}
    for prec in ("bf16", "fp8", "bf16"):          # back to bf16 at the end: the switch must rebuild the engine both ways
        set_precision(m, prec)
        for p in m.parameters():
            p.grad = None
        y = m(x)
        (y * w).sum().backward()
        torch.cuda.synchronize()
        assert m._engine.fp8 == (prec == "fp8")
        got = (y.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
        if prec in res:
            assert torch.equal(res[prec][0], got[0])     # bf16 -> fp8 -> bf16 reproduces the bf16 result bit for bit
        res[prec] = got
    y16, g16 = res["bf16"]
    y8, g8 = res["fp8"]
    with torch.no_grad():
        yo = (refcpu.vit_encoder(sd, image) if which == "vit" else refcpu.barcode_bert_encoder(sd, dna))
    cos = torch.nn.functional.cosine_similarity(y8, y16, dim=-1).min().item()
    worst = max(rel_err(g8[k], g16[k]) for k in g16)
    rec = {"test": f"fp8 {which} depth {depth}", "emb_fp8_vs_bf16": rel_err(y8, y16), "min_cosine_fp8_vs_bf16": cos,
           "emb_fp8_vs_f32_oracle": rel_err(y8, yo), "emb_bf16_vs_f32_oracle": rel_err(y16, yo), "worst_grad_fp8_vs_bf16": worst}
    _log(rec)
    tol = FP8_TOL[which]
    assert torch.isfinite(y8).all() and all(torch.isfinite(v).all() for v in g8.values())
    assert rec["emb_fp8_vs_bf16"] < tol["emb"] and cos > tol["cos"] and worst < tol["grad"], rec


# The real gate: against the f32 oracle and against the oracle that rounds to e4m3 exactly where the fp8 engines do
# (oracle/refcpu.py emulate_fp8: activations scale 1 saturated, weights per output row amax -> 448, LoRA / attention / out-projection
# / heads bf16, f32 residual stream).  What e4m3 costs is then a property of the dtype the CPU reproduces, not of the kernels:
#   * HIP vs the fp8-emulating oracle: as close as that oracle is to itself under f32 vs f64 accumulation (x FP8_SELF);
#   * HIP vs f32 oracle <= FP8_VS_EMU x (fp8-emulating oracle vs f32 oracle), and <= 2x the measured value in absolute terms;
#   * gradients likewise (the emulating oracle's backward runs through the same rounded forward, straight-through).
# Measured on the CPU emulation (tools/fp8_floor.py, profiles/r03_e_fp8_floor.log): per-tensor amax activation scales do not
# help (16.2 % vs 16.3 % at scale 1, ViT depth 6: values are O(1), the loss is the 3-bit mantissa, not range), per-token scales 15.3 %.
FP8_SELF, FP8_VS_EMU = 1.6, 1.3
# measured (MI355X, depth 6, B = 4): ViT HIP-vs-emulation 8.3 % (emulation f32- vs f64-accumulated: 8.0 %), HIP-vs-f32 14.8 %
# (emulation-vs-f32 15.4 %), worst gradient 55 % (emulation 50 %); DNA 3.3 % (2.7 %), 6.8 % (7.1 %), 18 % (23 %)
FP8_ABS = {"vit": dict(emb=0.30, grad=0.75), "dna": dict(emb=0.14, grad=0.37)}   # <= 2x measured vs the f32 oracle (ViT gradient: 1.4x)


@pytest.mark.parametrize("which", ["vit", "dna"])
def test_fp8_encoder_matches_the_fp8_emulating_oracle(ops, which):
    from bioscanclip.hip.engine import set_precision
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    from oracle import refcpu, synth
    depth, B = 6, 4
    if which == "vit":
        m, pre = LoRA_ViT_timm(arch.VisionTransformerParams(depth=depth), r=4, num_classes=768), "image_encoder."
    else:
        m = LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=depth, **NODROP)), r=4,
                              num_classes=768)
        pre = "dna_encoder."
    sd = synth.synth_state_dict({pre + k: v for k, v in synth.shapes_of(m).items()}, 17)
    m.load_state_dict({k[len(pre):]: v for k, v in sd.items()})
    m.cuda().train()
    set_precision(m, "fp8")
    image, dna, _, _ = synth.synth_batch(B, seed=29)
    xin = image if which == "vit" else dna
    w = synth.synth_tensor("fp8.cot", (B, 768), seed=5)
    y = m(xin.cuda())
    (y * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert m._engine.fp8
    fn = (lambda s, **kw: refcpu.vit_encoder(s, image.to(next(iter(s.values())).dtype) if kw.get("f64") else image,
                                             **{k: v for k, v in kw.items() if k != "f64"})) if which == "vit" else \
         (lambda s, **kw: refcpu.barcode_bert_encoder(s, dna, **{k: v for k, v in kw.items() if k != "f64"}))

    def run(**kw):
        s = {k: v.clone() for k, v in sd.items()}
        keys = [k for k in s if refcpu.is_trainable_key(k) and s[k].is_floating_point()]
        for k in keys:
            s[k].requires_grad_(True)
        out = fn(s, **kw)
        (out * w).sum().backward()
        return out.detach(), {k: s[k].grad for k in keys}
    y32, g32 = run()
    ye, ge = run(emulate_fp8=True)
    with torch.no_grad():
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        ye64 = fn(sd64, emulate_fp8=True, f64=True).float()
    named = dict(m.named_parameters())
    gh = {k: named[k[len(pre):]].grad.detach().cpu() for k in g32}
    yh = y.detach().cpu()
    rec = {"test": f"fp8 {which} depth {depth} vs fp8-emulating oracle",
           "emb_hip_vs_emu": rel_err(yh, ye), "emu_f32acc_vs_f64acc": rel_err(ye, ye64),
           "emb_hip_vs_f32": rel_err(yh, y32), "emb_emu_vs_f32": rel_err(ye, y32),
           "worst_grad_hip_vs_emu": max(rel_err(gh[k], ge[k]) for k in g32),
           "worst_grad_hip_vs_f32": max(rel_err(gh[k], g32[k]) for k in g32),
           "worst_grad_emu_vs_f32": max(rel_err(ge[k], g32[k]) for k in g32)}
    _log(rec)
    assert torch.isfinite(y).all()
    assert rec["emb_hip_vs_emu"] < FP8_SELF * rec["emu_f32acc_vs_f64acc"], rec
    assert rec["emb_hip_vs_f32"] < FP8_VS_EMU * rec["emb_emu_vs_f32"] and rec["emb_hip_vs_f32"] < FP8_ABS[which]["emb"], rec
    assert rec["worst_grad_hip_vs_f32"] < FP8_VS_EMU * rec["worst_grad_emu_vs_f32"] and rec["worst_grad_hip_vs_f32"] < FP8_ABS[which]["grad"], rec


def test_fp8_training_reduces_the_loss(ops):
    """configs[4] end to end: the golden I+D trajectory setup (B = 8, two batches cycled, AdamW) with fp8 trunks must still
    train -- the loss falls below a quarter of its start within the 10 steps and tracks the reference trajectory loosely."""
    from bioscanclip.hip.engine import set_precision
    from bioscanclip.hip.optim import FusedAdamW
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    from bioscanclip.model.loss_func import ContrastiveLoss
    from bioscanclip.model.simple_clip import SimpleCLIP
    from helpers import load_golden
    from oracle import synth
    g = load_golden("trajectory_id")
    model = SimpleCLIP(LoRA_ViT_timm(arch.vit_base_patch16_224(), r=4, num_classes=768),
                       LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(**NODROP)), r=4, num_classes=768), None)
    model.load_state_dict(synth.synth_state_dict(synth.shapes_of(model), seed=g["weight_seed"]))
    set_precision(model, "fp8")
    model.cuda().train()
    opt = FusedAdamW(model.parameters(), lr=g["lr"])
    crit = ContrastiveLoss(torch.nn.CrossEntropyLoss(), 1 / 0.07)
    losses = []
    for s in range(g["steps"]):
        image, dna, _, label = synth.synth_batch(g["B"], seed=g["batch_seed0"] + s % g["n_batches"])
        opt.zero_grad()
        loss = crit(*model(image.cuda(), dna.cuda(), None), label.cuda())
        loss.backward()
        if s == 0:
            opt.attach(model)
        opt.step()
        losses.append(loss.item())
    _log({"test": "fp8 trajectory", "losses": losses, "ref": g["losses"]})
    assert all(l == l for l in losses) and losses[-1] < 0.25 * losses[0], losses
    assert abs(losses[0] - g["losses"][0]) < 0.1 * g["losses"][0], (losses[0], g["losses"][0])
