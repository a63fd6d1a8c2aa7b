"""Host-side orchestration of the HIP kernels for the three encoders (forward + hand-written backward).

Everything numerical happens in ``libbsclip_hip.so``; this file decides *which* kernel runs on *which* buffer:
  * frozen backbone weights are packed once into bf16 [N,K] operands (plus the transposed copy the dX GEMMs
    need -- HBM is 288 GB, the extra 0.35 GB buys a single NT GEMM kernel for forward and backward);
  * trainable tensors (LoRA A/B, heads) are re-homed into one flat f32 buffer per encoder (``FlatParams``) whose
    layout is exactly what the kernels consume (``lora_a[8,H]``, ``lora_b[2,H,4]`` per layer), with a twin flat
    gradient buffer that ``p.grad`` aliases -- kernels accumulate straight into it and the fused AdamW kernel
    updates the whole buffer in one launch;
  * activations needed by backward live in a per-batch-size workspace sized for HBM (f32 residual stream saved per
    sub-layer, bf16 GEMM operands, attention LSE, LN statistics).
Reference call shapes: SURVEY.md 2.3 (K1-K15), 3.2; semantics App. A.1-A.3.
"""
import itertools
import os
import zlib

import torch

from . import ops
from .lib import (EPI_BF16, EPI_DGELU_BF16, EPI_F32, EPI_GELU_BF16, EPI_GELU_FP8, EPI_PATCH_BF16, EPI_PATCH_F32, EPI_RESID_BF16,
                  EPI_RESID_F32, KPAD)

BF16 = torch.bfloat16
F32 = torch.float32
# Residual-GRADIENT stream dtype of the LoRA-regime backward (the gradient that flows down the skip connections, read and
# written by every LayerNorm backward): "bf16" halves its bytes (LN backward is HBM-bound: 16 -> 10-12 bytes per element);
# the forward and the parameter gradients' f32 accumulation are unchanged.  "f32" restores round 2's behaviour.  Full
# fine-tuning keeps f32 (its LayerNorm parameter gradients read the same stream).
GRAD_STREAM_BF16 = os.environ.get("BSCLIP_GRAD_STREAM", "bf16").lower() != "f32"
# Residual stream dtype of the FORWARD (the tensor every sub-layer adds its output to; saved per sub-layer as the LayerNorm
# backward's input).  "bf16": each sum is formed in f32 in the GEMM epilogue and rounded once on the way to HBM; the
# out-projection / fc2 epilogues move 4 instead of 8 bytes per element, LayerNorm forward reads 2 instead of 4 (BERT: writes no
# f32 copy either), LayerNorm backward reads 2 instead of 4, and the saved stream halves (3.9 -> 2.0 GB for the ViT at B = 256).
# Cost, measured on the CPU oracle at depth 12: the distance to the f32 reference grows by ~10 % (2.5e-2 -> 2.8e-2),
# the gradients' not at all.  "f32" restores round 2's stream.  Full fine-tuning and fp8 trunks keep f32.
RESID_STREAM_BF16 = os.environ.get("BSCLIP_RESID_STREAM", "bf16").lower() != "f32"
# Patch embedding on split-bf16 operands (hi.hi + lo.hi + hi.lo in one K = 2304 GEMM): the first GEMM's operand rounding is the
# one every block amplifies -- exact to ~2^-16 it takes the ViT's distance to the f32 reference at depth 12 from 2.5e-2 to
# 1.7e-2 on the CPU emulation, for +0.8 % of the ViT's forward FLOPs.  "0" restores the plain bf16 GEMM.
PATCH_SPLIT = os.environ.get("BSCLIP_PATCH_SPLIT", "1") != "0"
# Attention-probs dropout (BERT towers, train mode): the forward kernel leaves its keep decisions as bit words, the backward reads them
# instead of re-hashing (csrc/attn_common.h KEEP_WORDS).  "0" = re-hash in the backward (round 4's form; identical masks and gradients).
ATTN_KEEP_BITS = os.environ.get("BSCLIP_ATTN_KEEP_BITS", "1") != "0"
# LoRA gradients: the attention backward leaves dt = dq B_q | dv B_v and dB = (dq | dv)^T t as per-head / per-sequence partial sums taken
# from its accumulators (csrc/attn.hip LoraPart), bsclip_lora_grad_heads reduces them -- instead of bsclip_lora_grad's pass that reads dq
# and dv back from HBM.  "0" = that pass (round 4's form; the same sums up to f32 accumulation order).
ATTN_LORA = os.environ.get("BSCLIP_ATTN_LORA", "1") != "0"


# BSCLIP_PARITY=1 / set_parity_mode(1): f32 residual stream, f32 residual-gradient stream, split-bf16 patch embedding -- at the
# price bench.py reports as `parity_mode_ms_per_step`.  NOT north_star's 1e-3: every trunk GEMM and the attention products still
# take bf16 operands.
# BSCLIP_PARITY=2 / set_parity_mode(2): the EXACT mode (round 4, csrc/exact.hip) -- every GEMM of the forward and of the backward
# on split-bf16 operands (hi + lo, K tripled: exact to ~2^-16 on the bf16 matrix cores), LoRA folded into the weight in f32,
# exact-erf GELU, every attention product on split operands too (round 5, csrc/attn_x3.hip: f32 in / out, f32 softmax arithmetic), f32
# LoRA gradients, f32 streams; the LayerNorm and attention kernels write the next GEMM's split operand themselves.  Embeddings, loss
# and every gradient within 1e-3 of the f32 reference at depth 12 (measured <= 1.6e-4, tests/test_20_encoders_gpu.py).
# LoRA-regime ViT / BERT engines only (fp8 and full fine-tuning keep their own paths, on the f32 streams).
EXACT_FORWARD = False
if os.environ.get("BSCLIP_PARITY", "0") in ("1", "2"):
    GRAD_STREAM_BF16 = RESID_STREAM_BF16 = False
    PATCH_SPLIT = True
    EXACT_FORWARD = os.environ.get("BSCLIP_PARITY") == "2"


def set_parity_mode(on, model=None):
    """Switch at run time: 0 / False = default, 1 / True = f32 streams, 2 = the exact mode (f32 streams + the exact forward and
    backward, csrc/exact.hip).  Engines built
    afterwards pick it up; pass ``model`` to have its engines rebuilt at the next forward.  Returns the previous
    (grad_stream_bf16, resid_stream_bf16) pair (restore with the module attributes; EXACT_FORWARD is reset by passing 0 / 1)."""
    global GRAD_STREAM_BF16, RESID_STREAM_BF16, EXACT_FORWARD
    prev = (GRAD_STREAM_BF16, RESID_STREAM_BF16)
    level = int(on)
    GRAD_STREAM_BF16 = RESID_STREAM_BF16 = level == 0
    EXACT_FORWARD = level == 2
    if model is not None:
        for m in model.modules():
            if getattr(m, "_engine", None) is not None:
                m._engine = None
    return prev


# BSCLIP_DETECT_ANOMALY=1 / set_detect_anomaly(True): this build's counterpart of ``torch.autograd.set_detect_anomaly(True)``, which
# the reference loop switches on in every epoch (train_epoch.py:12).  Every encoder forward / backward is followed by a device-side
# count of Inf / NaN values (bsclip_count_nonfinite) in its output / its flat gradient buffer, read back at once; on a hit the
# per-layer activations are probed in execution order and a RuntimeError names the first tensor that holds non-finite values (the
# reference's anomaly mode names the autograd node).  A debugging aid: it synchronises the host after every tower, so the step
# stays eager (train_epoch skips the captured-graph path) -- off by default, as SURVEY App. B-6 decides for timed runs.
DETECT_ANOMALY = os.environ.get("BSCLIP_DETECT_ANOMALY", "0") == "1"


def set_detect_anomaly(on):
    global DETECT_ANOMALY
    prev, DETECT_ANOMALY = DETECT_ANOMALY, bool(on)
    return prev


def count_nonfinite(t):
    """Number of Inf / NaN elements of an f32 / bf16 device tensor (host-synchronising: a debugging aid)."""
    c = torch.zeros(1, dtype=torch.int32, device=t.device)
    ops.count_nonfinite(t.contiguous(), c)
    return int(c.item())


def _anomaly_check(engine, phase, out=None):
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("BSCLIP_DETECT_ANOMALY reads every tower's result back to the host: it cannot run inside a hipGraph "
                           "capture (train_epoch takes the eager path when it is on)")
    who = type(engine).__name__
    if phase == "forward":
        if count_nonfinite(out) == 0:
            return
        for name, t in engine.anomaly_probes():
            if count_nonfinite(t):
                raise RuntimeError(f"{who} forward: non-finite values first appear in {name} (BSCLIP_DETECT_ANOMALY)")
        raise RuntimeError(f"{who} forward: non-finite values in the encoder output (BSCLIP_DETECT_ANOMALY)")
    if count_nonfinite(engine.flat.grad) == 0:
        return
    owner = getattr(engine, "_owner", None)
    names = {id(p): n for n, p in owner.named_parameters()} if owner is not None else {}
    bad = [names.get(id(p), f"trainable tensor #{i}") for i, p in enumerate(engine.flat.params)
           if p.grad is not None and count_nonfinite(p.grad)]
    raise RuntimeError(f"{who} backward: non-finite gradient values in {', '.join(bad[:6])}"
                       f"{' ...' if len(bad) > 6 else ''} (BSCLIP_DETECT_ANOMALY)")


_WS_GEN = itertools.count(1)   # every workspace gets a unique, never reused number (hip/graph.py keys captured graphs on it)


def _bf16(w, dev):
    return w.detach().to(dev, F32).to(BF16).contiguous()


def _f32(w, dev):
    return w.detach().to(dev, F32).contiguous()


def split_plan(M, N, K):
    """(splits, padded reduction length) for a weight gradient dW[N, K] = dY^T X reduced over M rows (bsclip_gemm_splitk_f32): as
    many K ranges as bring the launch to about two rounds of the 256 CUs, each at least four 64-wide K-tiles long.  The output
    is 9 .. 36 tiles of 256 x 256; one workgroup per tile would leave most of the chip idle for the whole reduction."""
    tiles = -(-N // 256) * (K // 256)
    if K % 256 or tiles >= 192 or M < 512:
        return 1, _pad64(M)
    S = max(1, min(512 // tiles, M // 256))
    unit = 64 * S
    return S, -(-M // unit) * unit


def _pad64(n):
    return (n + 63) // 64 * 64


def _rank():
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class FlatParams:
    """Contiguous f32 home for a list of trainable parameters and their gradients."""

    def __init__(self, params, device):
        self.params = list(params)
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.offsets = offs
        self.data = torch.zeros(total, dtype=F32, device=device)
        self.grad = torch.zeros(total, dtype=F32, device=device)
        for p, o in zip(self.params, offs):
            view = self.data[o:o + p.numel()].view(p.shape)
            view.copy_(p.data.to(device, F32))
            p.data = view
        self.bind_grads(force=True)

    def view(self, p_index, shape=None, n=None, grad=False):
        o = self.offsets[p_index]
        buf = self.grad if grad else self.data
        n = self.params[p_index].numel() if n is None else n
        t = buf[o:o + n]
        return t.view(shape) if shape is not None else t

    def valid(self):
        return all(p.data.data_ptr() == self.data.data_ptr() + 4 * o and p.data.device == self.data.device
                   for p, o in zip(self.params, self.offsets))

    def bind_grads(self, force=False):
        """Make every ``p.grad`` alias the flat gradient buffer.  If autograd/optimizer dropped the grads
        (``zero_grad(set_to_none=True)``) the buffer is cleared first, so accumulate-into-grad stays correct."""
        ok = [p.grad is not None and p.grad.data_ptr() == self.grad.data_ptr() + 4 * o
              for p, o in zip(self.params, self.offsets)]
        if all(ok) and not force:
            return
        if not any(ok):
            self.grad.zero_()
        for p, o, good in zip(self.params, self.offsets, ok):
            if not good:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


def _pack_qkv(w_qkv, b_qkv, H, dev):
    """W_aug bf16 [3H, H+KPAD] (LoRA-B columns refreshed every step) and W^T bf16 [H, 3H] for dX."""
    w = w_qkv.detach().to(dev, F32)
    waug = torch.zeros(3 * H, H + KPAD, dtype=BF16, device=dev)
    waug[:, :H] = w.to(BF16)
    return waug, w.t().contiguous().to(BF16), _f32(b_qkv, dev)


def _pack_linear(lin, dev):
    w = lin.weight.detach().to(dev, F32)
    return w.to(BF16).contiguous(), w.t().contiguous().to(BF16), _f32(lin.bias, dev)


class _Layer:
    pass


class EncoderEngineBase:
    """Shared pieces of the ViT and BERT engines: LoRA bookkeeping and the flat trainable buffer."""

    def _setup_lora(self, qv_modules, H, dev, extra_params):
        """qv_modules: per layer either None (no LoRA) or (A_q, A_v, B_q, B_v) nn.Linear modules."""
        params = []
        self._lora_index = []
        for mods in qv_modules:
            if mods is None:
                self._lora_index.append(None)
                continue
            self._lora_index.append(len(params))
            params += [mods[0].weight, mods[1].weight, mods[2].weight, mods[3].weight]
        self._extra_index = len(params)
        params += list(extra_params)
        self._trunk_index = len(params)
        params += list(self._trunk_trainables())   # full fine-tuning (hip/engine_ft.py): every trunk tensor, in its order
        ops.init_tables()
        self.flat = FlatParams(params, dev)
        self._zero_a = torch.zeros(8, H, dtype=F32, device=dev)

    full_ft = False

    def _trunk_trainables(self):
        return []

    def lora_a(self, l, grad=False):
        i = self._lora_index[l]
        if i is None:
            return None if grad else self._zero_a
        return self.flat.view(i, (8, self.H), n=8 * self.H, grad=grad)

    def lora_b(self, l, grad=False):
        i = self._lora_index[l]
        if i is None:
            return None
        return self.flat.view(i + 2, (2, self.H, 4), n=8 * self.H, grad=grad)

    def extra(self, k, grad=False):
        return self.flat.view(self._extra_index + k, self.flat.params[self._extra_index + k].shape, grad=grad)

    def refresh_lora_weights(self):
        """LoRA-B into the K-augmentation columns of every layer's QKV weight: one launch over a table of device addresses (the
        weights and the flat parameter buffer live as long as the engine)."""
        if getattr(self, "_waug_table", None) is None:
            rows = [(lay.waug.data_ptr(), self.lora_b(l)[0].data_ptr(), self.lora_b(l)[1].data_ptr())
                    for l, lay in enumerate(self.layers) if self.lora_b(l) is not None]
            lds = {_lay.waug.stride(0) for _lay in self.layers}
            assert len(lds) == 1
            self._waug_table = (torch.tensor(rows, dtype=torch.int64, device=self.device).reshape(-1, 3).contiguous(), len(rows), lds.pop())
        table, n, ld_w = self._waug_table
        if n:
            ops.waug_set_lora_layers(table, n, ld_w, self.H)
        if self.fp8:  # LoRA-B columns of the bf16 K-augmentation tile, in the fp8 accumulator's units
            for l, lay in enumerate(self.layers):
                b = self.lora_b(l)
                if b is not None:
                    ops.lora_baug_set(lay.baug, self.H, b[0], b[1], lay.s_qkv)

    fp8 = False

    # ------------------------------------------------------------------------------ exact mode (BSCLIP_PARITY=2)
    def exact(self):
        return EXACT_FORWARD and not self.fp8 and not self.full_ft

    def _ex_weight(self, w_f32, lora=None):
        """[hi | hi | lo] rows of a frozen f32 weight (LoRA folded in f32 when given: refreshed by the caller every step)."""
        w = w_f32.detach().to(self.device, F32).contiguous()
        dst = torch.empty(w.shape[0], 3 * w.shape[1], dtype=BF16, device=self.device)
        if lora is None:
            ops.split3_weight(w, dst)
        return w, dst

    def _ex_gemm(self, a_f32, w3, out, epi, scratch, M=None, **kw):
        """out = epilogue(a_f32 @ W^T) with both operands split: a_f32 f32 [M, K] (row stride free) -> [hi | lo | hi] in ``scratch``."""
        M = a_f32.shape[0] if M is None else M
        K = w3.shape[1] // 3
        a3 = ops.split3_rows(a_f32, scratch, M=M, K=K)
        return ops.gemm(a3, w3, out, epi, M=M, K=3 * K, **kw)

    def _ex_gemm3(self, a3, w3, out, epi, M, **kw):
        """The same with the A operand already split: ``a3`` bf16 [>= M, >= 3K] = [hi | lo | hi], written by the producer itself (the
        LayerNorm kernels' y_split3 / dx_split3 outputs, round 5) instead of a stand-alone split3_rows pass over an f32 tensor."""
        K = w3.shape[1] // 3
        return ops.gemm(a3[:M, :3 * K], w3, out, epi, M=M, K=3 * K, **kw)

    def _ex_weight_t(self, w_f32):
        """[hi | hi | lo] rows of the TRANSPOSE of a frozen f32 weight [N, K] -> bf16 [K, 3N]: the B operand of its dX GEMM."""
        w = w_f32.detach().to(self.device, F32).contiguous()
        N, K = w.shape
        assert N % 64 == 0
        return ops.split3_transpose(w, torch.empty(K * 3 * N, dtype=BF16, device=self.device), 1)

    def _ex_dw(self, dy, x, gw, ws, M=None):
        """gw[N, K] += dy[M, N]^T x[M, K], f32 operands split along the reduction (the M rows, zero-padded to a multiple of 64)."""
        M = dy.shape[0] if M is None else M
        a = ops.split3_transpose(dy, ws["t3a"], 0, R=M)
        b = ops.split3_transpose(x, ws["t3b"], 1, R=M)
        ops.gemm(a, b, gw, EPI_RESID_F32, resid=gw)

    def _ex_backward_weights(self):
        """Transposed split weights of the frozen Linears (once) and the buffer of the LoRA-folded QKV transpose (refilled every step)."""
        for lay in self.layers:
            if not hasattr(lay, "w3t"):
                lay.w3t = [self._ex_weight_t(w) for w in lay.src[1:]]
                lay.wqkvT3 = torch.empty(self.H * 9 * self.H, dtype=BF16, device=self.device)

    def _pack_fp8(self, lay, w_qkv, w_fc1, w_fc2, dev):
        """BASELINE configs[4]: the frozen QKV / fc1 / fc2 weights as OCP fp8 e4m3 with one scale per output row (row amax
        -> 448); activations are quantised with scale 1 by their producers (LayerNorm, GELU epilogue), so a GEMM's
        dequantisation vector alpha is the weight-row scale itself.  LoRA, attention, out-projection, heads, loss and the
        whole backward stay in bf16 / f32."""
        lay.w_qkv8, lay.s_qkv = ops.quantize_rows_fp8(_f32(w_qkv, dev))
        lay.w_fc1_8, lay.s_fc1 = ops.quantize_rows_fp8(_f32(w_fc1, dev))
        lay.w_fc2_8, lay.s_fc2 = ops.quantize_rows_fp8(_f32(w_fc2, dev))
        lay.baug = torch.zeros(3 * self.H, KPAD, dtype=BF16, device=dev)


# ======================================================================================================== ViT
class ViTEngine(EncoderEngineBase):
    """LoRA ViT-B/16 forward/backward (reference image_encoder.py:15-109 over timm vit_base_patch16_224)."""

    def __init__(self, module, device, fp8=False):
        vit = module.lora_vit
        self.fp8 = bool(fp8)
        self.device = dev = device
        self.H = H = vit.blocks[0].norm1.weight.numel()
        self.heads = vit.blocks[0].attn.num_heads
        self.S = vit.pos_embed.shape[1]
        assert H == 768 and self.S == 197 and vit.patch_embed.proj.weight.shape[-1] == 16, \
            "HIP ViT engine is built for vit_base_patch16_224"
        self.w_patch = _bf16(vit.patch_embed.proj.weight.reshape(H, -1), dev)
        # split-bf16 patch embedding (PATCH_SPLIT): the weight as [hi | hi | lo] against im2col rows [hi | lo | hi]
        w32 = vit.patch_embed.proj.weight.detach().reshape(H, -1).to(dev, F32)
        w_hi = w32.to(BF16)
        self.w_patch3 = torch.cat([w_hi, w_hi, (w32 - w_hi.to(F32)).to(BF16)], dim=1).contiguous()
        self.b_patch = _f32(vit.patch_embed.proj.bias, dev)
        self.cls = _f32(vit.cls_token.reshape(-1), dev)
        self.pos = _f32(vit.pos_embed.reshape(self.S, H), dev)
        self.layers, qv = [], []
        for blk in vit.blocks:
            lay = _Layer()
            q = blk.attn.qkv
            if hasattr(q, "linear_a_q"):
                base = q.qkv
                qv.append((q.linear_a_q, q.linear_a_v, q.linear_b_q, q.linear_b_v))
            else:
                base = q
                qv.append(None)
            lay.waug, lay.wqkv_t, lay.b_qkv = _pack_qkv(base.weight, base.bias, H, dev)
            lay.src = (base.weight, blk.attn.proj.weight, blk.mlp.fc1.weight, blk.mlp.fc2.weight)   # f32 masters (exact forward)
            lay.ln1 = (_f32(blk.norm1.weight, dev), _f32(blk.norm1.bias, dev))
            lay.ln2 = (_f32(blk.norm2.weight, dev), _f32(blk.norm2.bias, dev))
            lay.w_proj, lay.w_proj_t, lay.b_proj = _pack_linear(blk.attn.proj, dev)
            lay.w_fc1, lay.w_fc1_t, lay.b_fc1 = _pack_linear(blk.mlp.fc1, dev)
            lay.w_fc2, lay.w_fc2_t, lay.b_fc2 = _pack_linear(blk.mlp.fc2, dev)
            if self.fp8:
                self._pack_fp8(lay, base.weight, blk.mlp.fc1.weight, blk.mlp.fc2.weight, dev)
            self.layers.append(lay)
        self.ln_f = (_f32(vit.norm.weight, dev), _f32(vit.norm.bias, dev))
        self.FF = self.layers[0].w_fc1.shape[0]
        self.out_dim = vit.head.weight.shape[0]
        assert self.out_dim % 128 == 0, "head width must be a multiple of 128 for the HIP GEMM"
        self._setup_lora(qv, H, dev, [vit.head.weight, vit.head.bias])
        self.w_head_bf = torch.empty(self.out_dim, H, dtype=BF16, device=dev)
        self.w_head_t = torch.empty(H, self.out_dim, dtype=BF16, device=dev)
        self.ws = None

    # ------------------------------------------------------------------------------------------ workspace
    def _workspace(self, B):
        if self.ws is not None and self.ws["B"] == B:
            return self.ws
        dev, H, S, L, FF = self.device, self.H, self.S, len(self.layers), self.FF
        M = B * S
        z = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=dev)
        ws = {"B": B, "M": M, "gen": next(_WS_GEN)}
        ws["cols"] = z(B * 196, 3 * H if (PATCH_SPLIT and not self.full_ft) else H)
        rb = ws["resid_bf16"] = RESID_STREAM_BF16 and not self.full_ft and not self.fp8
        if self.exact():
            assert not rb and not GRAD_STREAM_BF16, "the exact forward runs on the f32 streams"
            # per layer, all f32: LN1 output, q | k | v, attention output, fc1 pre-activation -- what the exact backward reads
            ws["y32s"], ws["ctx32s"] = [z(M, H, dt=F32) for _ in range(L)], [z(M, H, dt=F32) for _ in range(L)]
            ws["qkv32s"] = [z(M, 3 * H, dt=F32) for _ in range(L)]
            ws["z32s"] = [z(M if l < L - 1 else B, FF, dt=F32) for l in range(L)]
            ws["y32"], ws["cls32"] = z(M, H, dt=F32), z(B, H, dt=F32)
            ws["a3"] = z(M, 3 * FF)                                   # [hi | lo | hi] rows of the current GEMM's A operand
            ws["whead3"] = z(self.out_dim, 3 * H)
            Bq = _pad64(B)
            ws["g32"], ws["dh32"], ws["dctx32"], ws["dqkv32"] = z(M, FF, dt=F32), z(M, H, dt=F32), z(M, H, dt=F32), z(M, 3 * H, dt=F32)
            ws["dcls32"] = z(B, H, dt=F32)
            ws["t3a"], ws["t3b"] = z(self.out_dim * 3 * Bq), z(H * 3 * Bq)         # dW operands of the head
            ws["wheadT3"] = z(H * 3 * self.out_dim)
        ws["x"] = [z(M, H, dt=BF16 if rb else F32) for _ in range(2 * L + 1)]   # residual stream after every sub-layer
        ws["h1"] = [z(M, H + KPAD) for _ in range(L)]                 # LN1 output + LoRA t (QKV operand)
        ws["st1"] = [z(M, 2, dt=F32) for _ in range(L)]
        ws["st2"] = [z(M, 2, dt=F32) for _ in range(L)]
        ws["qkv"] = [z(M, 3 * H) for _ in range(L)]
        ws["ctx"] = [z(M, H) for _ in range(L)]
        ws["lse"] = [z(B, self.heads, S, dt=F32) for _ in range(L)]
        ws["z"] = [z(M, FF, dt=torch.uint8) for _ in range(L)]        # gelu'(fc1 pre-activation), 8-bit codes
        ws["h2"] = z(M, H)
        ws["act"] = z(M, FF)
        if self.full_ft:  # inputs of fc1 / fc2 per block: operands of their weight-gradient GEMMs
            ws["h2s"] = [z(M, H) for _ in range(L)]
            ws["acts"] = [z(M, FF) for _ in range(L)]
        if self.fp8:  # fp8 GEMM operands: LN1 output per block (lora_grad reads it again) + its bf16 t block, LN2 / GELU shared
            ws["h1_8"] = [z(M, H, dt=ops.FP8) for _ in range(L)]
            ws["t"] = [z(M, KPAD) for _ in range(L)]
            ws["h2_8"], ws["act8"] = z(M, H, dt=ops.FP8), z(M, FF, dt=ops.FP8)
        ws["clsn"] = z(B, H)
        ws["st_f"] = z(B, 2, dt=F32)
        # backward temporaries
        ws["grad_bf16"] = GRAD_STREAM_BF16 and not self.full_ft
        ws["dx"] = None if ws["grad_bf16"] else torch.zeros(M, H, dtype=F32, device=dev)
        ws["dxb"] = torch.zeros(M, H, dtype=BF16, device=dev)
        ws["dz"] = z(M, FF)
        ws["dh"] = z(M, H)
        ws["dctx"] = z(M, H)
        ws["dqkv"] = z(M, 3 * H)
        ws["dt"] = z(M, 8, dt=F32)
        ws["dtp"], ws["dbp"] = z(self.heads, 2, M, 4, dt=F32), z(B * self.heads, 2, 4, 64, dt=F32)   # LoRA partial sums (ATTN_LORA)
        Bp = _pad64(B)
        ws["dout_bf"] = torch.zeros(B, self.out_dim, dtype=BF16, device=dev)
        ws["dout_t"] = torch.zeros(self.out_dim, Bp, dtype=BF16, device=dev)
        ws["clsn_t"] = torch.zeros(H, Bp, dtype=BF16, device=dev)
        ws["dclsn"] = z(B, H)
        ws["dz_c"], ws["dh_c"], ws["st_c"] = z(B, FF), z(B, H), z(B, 2, dt=F32)   # last-block token-0 path
        ws["h2_c"], ws["act_c"], ws["z_c"] = z(B, H), z(B, FF), z(B, FF, dt=torch.uint8)
        self.ws = ws
        return ws

    def anomaly_probes(self):
        """(name, tensor) of the forward's saved activations in execution order (BSCLIP_DETECT_ANOMALY).  The last block's
        sub-layers exist on the token-0 rows only."""
        ws, L, B, S, H = self.ws, len(self.layers), self.ws["B"], self.S, self.H
        qkv, ctx = ("qkv32s", "ctx32s") if self.exact() else ("qkv", "ctx")
        yield "the patch embedding + position table", ws["x"][0]
        for l in range(L):
            yield f"blocks.{l}.attn.qkv output", ws[qkv][l]
            if l < L - 1:
                yield f"blocks.{l}.attn output", ws[ctx][l]
            tok0 = (lambda t: t.view(B, S * H)[:, :H]) if l == L - 1 else (lambda t: t)
            yield f"blocks.{l} residual stream after attention", tok0(ws["x"][2 * l + 1])
            yield f"blocks.{l} residual stream after the MLP", tok0(ws["x"][2 * l + 2])

    # -------------------------------------------------------------------------------------------- forward
    def _forward_exact(self, image, ws):
        """BSCLIP_PARITY=2: the same block sequence with every Linear and every attention product on split-bf16 operands (f32 in / out), exact-erf GELU; what
        ``_backward_exact`` reads -- per layer the f32 LN1 output, q | k | v, attention output and fc1 pre-activation, the
        LayerNorm statistics, lse, the f32 residual stream -- stays resident."""
        B, H, S, M, L, FF = image.shape[0], self.H, self.S, ws["M"], len(self.layers), self.FF
        scale = 64 ** -0.5
        x, a3 = ws["x"], ws["a3"]
        for lay in self.layers:
            if not hasattr(lay, "w3"):
                lay.qkv32, lay.wqkv3 = self._ex_weight(lay.src[0], lora=True)
                lay.w3 = [self._ex_weight(w)[1] for w in lay.src[1:]]   # proj, fc1, fc2
        ops.im2col_patch16(image, ws["cols"])
        ops.gemm(ws["cols"], self.w_patch3, x[0], EPI_PATCH_F32, bias=self.b_patch, resid=self.pos)
        ops.vit_cls_rows(x[0], self.cls, self.pos, B, S, H)
        tok0 = lambda t, w: t.view(B, S * w)[:, :w]
        for l, lay in enumerate(self.layers):
            has = self._lora_index[l] is not None
            y32, qkv32, ctx32, z32 = ws["y32s"][l], ws["qkv32s"][l], ws["ctx32s"][l], ws["z32s"][l]
            # the LayerNorm writes the QKV GEMM's split operand itself (and the f32 copy the LoRA gradients read in the backward)
            ops.layernorm_fwd(x[2 * l], lay.ln1[0], lay.ln1[1], 1e-6, y_f32=y32, y_split3=a3, stats=ws["st1"][l])
            ops.split3_weight(lay.qkv32, lay.wqkv3, lora_a=self.lora_a(l) if has else None, lora_b=self.lora_b(l) if has else None)
            self._ex_gemm3(a3, lay.wqkv3, qkv32, EPI_F32, M, bias=lay.b_qkv)
            ops.attn_fwd_f32(qkv32, B, S, self.heads, scale, ctx32, ws["lse"][l], ctx_split3=a3 if l < L - 1 else None)
            if l == L - 1:   # token-0 rows only, as the default path (and as its backward expects)
                self._ex_gemm(tok0(ctx32, H), lay.w3[0], tok0(x[2 * l + 1], H), EPI_RESID_F32, a3, bias=lay.b_proj,
                              resid=tok0(x[2 * l], H))
                ops.layernorm_fwd(tok0(x[2 * l + 1], H), lay.ln2[0], lay.ln2[1], 1e-6, y_bf16=ws["h2_c"], y_f32=ws["cls32"],
                                  stats=ws["st_c"])
                self._ex_gemm(ws["cls32"], lay.w3[1], z32, EPI_F32, a3, bias=lay.b_fc1)
                g3 = ops.gelu_split3(z32, a3, M=B)
                ops.gemm(g3, lay.w3[2], tok0(x[2 * l + 2], H), EPI_RESID_F32, bias=lay.b_fc2, resid=tok0(x[2 * l + 1], H), M=B)
                continue
            self._ex_gemm3(a3, lay.w3[0], x[2 * l + 1], EPI_RESID_F32, M, bias=lay.b_proj, resid=x[2 * l])   # a3: written by the attention kernel
            ops.layernorm_fwd(x[2 * l + 1], lay.ln2[0], lay.ln2[1], 1e-6, y_split3=a3, stats=ws["st2"][l])
            self._ex_gemm3(a3, lay.w3[1], z32, EPI_F32, M, bias=lay.b_fc1)
            g3 = ops.gelu_split3(z32, a3)
            ops.gemm(g3, lay.w3[2], x[2 * l + 2], EPI_RESID_F32, bias=lay.b_fc2, resid=x[2 * l + 1])
        ops.layernorm_fwd(tok0(x[-1], H), self.ln_f[0], self.ln_f[1], 1e-6, y_bf16=ws["clsn"], y_f32=ws["cls32"], stats=ws["st_f"])
        ops.split3_weight(self.extra(0), ws["whead3"])
        out = torch.empty(B, self.out_dim, dtype=F32, device=self.device)
        self._ex_gemm(ws["cls32"], ws["whead3"], out, EPI_F32, a3, bias=self.extra(1))
        return out

    def forward(self, image):
        B = image.shape[0]
        ws = self._workspace(B)
        H, S, M = self.H, self.S, ws["M"]
        scale = 64 ** -0.5
        self.refresh_lora_weights()
        ops.cast_f32_bf16(self.extra(0), self.w_head_bf)
        if self.exact():
            return self._forward_exact(image, ws)
        x = ws["x"]
        EPI_R, EPI_P = (EPI_RESID_BF16, EPI_PATCH_BF16) if ws["resid_bf16"] else (EPI_RESID_F32, EPI_PATCH_F32)
        ops.im2col_patch16(image, ws["cols"])
        ops.gemm(ws["cols"], self.w_patch3 if ws["cols"].shape[1] == 3 * H else self.w_patch, x[0], EPI_P, bias=self.b_patch,
                 resid=self.pos)
        ops.vit_cls_rows(x[0], self.cls, self.pos, B, S, H)
        L = len(self.layers)
        for l, lay in enumerate(self.layers):
            if self.fp8:
                ops.layernorm_fwd_fp8(x[2 * l], lay.ln1[0], lay.ln1[1], 1e-6, ws["h1_8"][l], t_aug=ws["t"][l],
                                      lora_a=self.lora_a(l), stats=ws["st1"][l])
                ops.gemm_fp8(ws["h1_8"][l], lay.w_qkv8, ws["qkv"][l], lay.s_qkv, lay.b_qkv, EPI_BF16, a_aug=ws["t"][l],
                             b_aug=lay.baug)
            else:
                ops.layernorm_fwd(x[2 * l], lay.ln1[0], lay.ln1[1], 1e-6, y_bf16=ws["h1"][l], lora_a=self.lora_a(l),
                                  stats=ws["st1"][l])
                ops.gemm(ws["h1"][l], lay.waug, ws["qkv"][l], EPI_BF16, bias=lay.b_qkv)
            if l == L - 1:
                # The head reads token 0 of the last block only (timm global_pool='token'), and within a block a token's
                # output depends on the other tokens through K and V alone: attention for query 0, then proj / LN2 / MLP on
                # the B token-0 rows (row stride S*H) instead of all B*197.  Same values, 1/197 of the GEMM work.
                tok0 = lambda t, w: t.view(B, S * w)[:, :w]
                ops.attn_fwd(ws["qkv"][l], B, S, self.heads, scale, ws["ctx"][l], ws["lse"][l], q_rows=1)
                ops.gemm(tok0(ws["ctx"][l], H), lay.w_proj, tok0(x[2 * l + 1], H), EPI_R, bias=lay.b_proj,
                         resid=tok0(x[2 * l], H))
                ops.layernorm_fwd(tok0(x[2 * l + 1], H), lay.ln2[0], lay.ln2[1], 1e-6, y_bf16=ws["h2_c"],
                                  stats=ws["st_c"])
                ops.gemm(ws["h2_c"], lay.w_fc1, ws["act_c"], EPI_GELU_BF16, bias=lay.b_fc1, aux=ws["z_c"])
                ops.gemm(ws["act_c"], lay.w_fc2, tok0(x[2 * l + 2], H), EPI_R, bias=lay.b_fc2,
                         resid=tok0(x[2 * l + 1], H))
                continue
            ops.attn_fwd(ws["qkv"][l], B, S, self.heads, scale, ws["ctx"][l], ws["lse"][l])
            ops.gemm(ws["ctx"][l], lay.w_proj, x[2 * l + 1], EPI_R, bias=lay.b_proj, resid=x[2 * l])
            if self.fp8:
                ops.layernorm_fwd_fp8(x[2 * l + 1], lay.ln2[0], lay.ln2[1], 1e-6, ws["h2_8"], stats=ws["st2"][l])
                ops.gemm_fp8(ws["h2_8"], lay.w_fc1_8, ws["act8"], lay.s_fc1, lay.b_fc1, EPI_GELU_FP8, aux=ws["z"][l])
                ops.gemm_fp8(ws["act8"], lay.w_fc2_8, x[2 * l + 2], lay.s_fc2, lay.b_fc2, EPI_RESID_F32, resid=x[2 * l + 1])
                continue
            h2, act = (ws["h2s"][l], ws["acts"][l]) if self.full_ft else (ws["h2"], ws["act"])   # dW needs them per layer
            ops.layernorm_fwd(x[2 * l + 1], lay.ln2[0], lay.ln2[1], 1e-6, y_bf16=h2, stats=ws["st2"][l])
            ops.gemm(h2, lay.w_fc1, act, EPI_GELU_BF16, bias=lay.b_fc1, aux=ws["z"][l])
            ops.gemm(act, lay.w_fc2, x[2 * l + 2], EPI_R, bias=lay.b_fc2, resid=x[2 * l + 1])
        x_cls = x[-1].view(B, S * H)[:, :H]  # token 0 of every image (row stride S*H)
        ops.layernorm_fwd(x_cls, self.ln_f[0], self.ln_f[1], 1e-6, y_bf16=ws["clsn"], stats=ws["st_f"])
        out = torch.empty(B, self.out_dim, dtype=F32, device=self.device)
        ops.gemm(ws["clsn"], self.w_head_bf, out, EPI_F32, bias=self.extra(1))
        return out

    # ------------------------------------------------------------------------------------------- backward
    def backward(self, dout):
        if self.exact():
            return self._backward_exact(dout)
        ws = self.ws
        B, M, H, S = ws["B"], ws["M"], self.H, self.S
        scale = 64 ** -0.5
        self.flat.bind_grads()
        x = ws["x"]
        dx, dxb = ws["dx"], ws["dxb"]
        # head: dW = dout^T clsn, db = colsum(dout), dclsn = dout W
        ops.cast_f32_bf16(dout, ws["dout_bf"])
        ops.transpose_bf16(ws["dout_bf"], B, self.out_dim, ws["dout_t"])
        ops.transpose_bf16(ws["clsn"], B, H, ws["clsn_t"])
        gw = self.extra(0, grad=True)
        ops.gemm(ws["dout_t"], ws["clsn_t"], gw, EPI_RESID_F32, resid=gw)
        ops.colsum(dout, B, self.out_dim, self.extra(1, grad=True))
        ops.transpose_bf16(self.w_head_bf, self.out_dim, H, self.w_head_t)
        ops.gemm(ws["dout_bf"], self.w_head_t, ws["dclsn"], EPI_BF16)
        # final norm on token-0 rows only: every other row of the residual gradient is zero
        # bf16 gradient stream (no dropout in the ViT): the bf16 operand buffer IS the residual gradient, read and rewritten
        # in place by each LayerNorm backward (same lane, same elements); the f32 copy does not exist
        g16 = ws["grad_bf16"]
        R = dxb if g16 else dx          # the residual-gradient stream
        if not g16:
            dx.zero_()
        dxb.zero_()
        ops.layernorm_bwd(x[-1].view(B, S * H)[:, :H], ws["st_f"], self.ln_f[0], 0, g_gemm=ws["dclsn"],
                          dx_f32=None if g16 else dx.view(B, S * H)[:, :H], dx_bf16=dxb.view(B, S * H)[:, :H])
        L = len(self.layers)
        for l in range(L - 1, -1, -1):
            lay = self.layers[l]
            if l == L - 1:
                # Only token 0 of the last block feeds the head, so the residual gradient entering this block is zero
                # on the other 196 rows of every image: its MLP backward, LN2 backward and proj backward run on the B
                # token-0 rows only (row stride S*H), 1/197 of the work.
                FF = self.FF
                dxb_c = dxb.view(B, S * H)[:, :H]
                ops.gemm(dxb_c, lay.w_fc2_t, ws["dz_c"], EPI_DGELU_BF16, aux=ws["z_c"])
                ops.gemm(ws["dz_c"], lay.w_fc1_t, ws["dh_c"], EPI_BF16)
                dx_c = R.view(B, S * H)[:, :H]
                ops.layernorm_bwd(x[2 * l + 1].view(B, S * H)[:, :H], ws["st_c"], lay.ln2[0], 0, g_resid=dx_c,
                                  g_gemm=ws["dh_c"], dx_f32=None if g16 else dx_c, dx_bf16=dxb_c)
                ws["dctx"].zero_()
                ops.gemm(dxb_c, lay.w_proj_t, ws["dctx"].view(B, S * H)[:, :H], EPI_BF16)
            else:
                ops.gemm(dxb, lay.w_fc2_t, ws["dz"], EPI_DGELU_BF16, aux=ws["z"][l])
                ops.gemm(ws["dz"], lay.w_fc1_t, ws["dh"], EPI_BF16)
                ops.layernorm_bwd(x[2 * l + 1], ws["st2"][l], lay.ln2[0], 0, g_resid=R, g_gemm=ws["dh"],
                                  dx_f32=None if g16 else dx, dx_bf16=dxb)
                ops.gemm(dxb, lay.w_proj_t, ws["dctx"], EPI_BF16)
            lb = self.lora_b(l)
            part = lb is not None and ATTN_LORA and not self.fp8    # dt / dB partial sums out of the attention backward
            ops.attn_bwd(ws["qkv"][l], ws["dctx"], ws["lse"][l], B, S, self.heads, scale, ws["dqkv"],
                         q_rows=1 if l == L - 1 else 0, lora=(ws["h1"][l][:, H:], lb, ws["dtp"], ws["dbp"]) if part else None)
            if lb is not None:
                gb = self.lora_b(l, grad=True)
                if self.fp8:
                    ops.lora_grad_fp8(ws["dqkv"], ws["h1_8"][l], ws["t"][l], M, H, lb, ws["dt"], self.lora_a(l, grad=True),
                                      gb[0], gb[1])
                elif part:
                    ops.lora_grad_heads(ws["h1"][l], M, H, B, ws["dtp"], ws["dbp"], ws["dt"], self.lora_a(l, grad=True), gb[0], gb[1])
                else:
                    ops.lora_grad(ws["dqkv"], ws["h1"][l], M, H, lb, ws["dt"], self.lora_a(l, grad=True), gb[0], gb[1])
            if l > 0:  # nothing trainable sits below block 0 (patch-embed, cls, pos are frozen)
                ops.gemm(ws["dqkv"], lay.wqkv_t, ws["dh"], EPI_BF16)
                ops.layernorm_bwd(x[2 * l], ws["st1"][l], lay.ln1[0], 0, g_resid=R, g_gemm=ws["dh"],
                                  dt=ws["dt"] if lb is not None else None,
                                  lora_a=self.lora_a(l) if lb is not None else None, dx_f32=None if g16 else dx, dx_bf16=dxb)


    def _backward_exact(self, dout):
        """BSCLIP_PARITY=2: the backward of ``_forward_exact`` with every gradient in f32 -- dX GEMMs on split operands against the
        transposed split weights (LoRA folded: W + B A), exact gelu' from the f32 pre-activation, the attention backward on split operands (f32 in / out), f32 LoRA
        gradients, the head's dW on operands split along the batch."""
        ws = self.ws
        B, M, H, S, L = ws["B"], ws["M"], self.H, self.S, len(self.layers)
        scale = 64 ** -0.5
        self.flat.bind_grads()
        self._ex_backward_weights()
        x, dx, a3 = ws["x"], ws["dx"], ws["a3"]
        tok0 = lambda t, w: t.view(B, S * w)[:, :w]
        dout = dout.contiguous()
        self._ex_dw(dout, ws["cls32"], self.extra(0, grad=True), ws)            # cls32 = the final LayerNorm's f32 output
        ops.colsum(dout, B, self.out_dim, self.extra(1, grad=True))
        self._ex_gemm(dout, ops.split3_transpose(self.extra(0), ws["wheadT3"], 1), ws["dcls32"], EPI_F32, a3)
        dx.zero_()
        dx_c = tok0(dx, H)
        ops.layernorm_bwd(tok0(x[-1], H), ws["st_f"], self.ln_f[0], 0, g_gemm=ws["dcls32"], dx_f32=dx_c)
        dx3 = False     # a3 holds the split of dx (written by the LayerNorm backward that produced dx: no split3_rows pass)
        for l in range(L - 1, -1, -1):
            lay = self.layers[l]
            if l == L - 1:    # token-0 rows only (see backward)
                self._ex_gemm(dx_c, lay.w3t[2], ws["g32"], EPI_F32, a3, M=B)
                g3 = ops.dgelu_split3(ws["g32"], ws["z32s"][l], dst=a3, M=B)
                ops.gemm(g3, lay.w3t[1], ws["dh32"], EPI_F32, M=B)
                ops.layernorm_bwd(tok0(x[2 * l + 1], H), ws["st_c"], lay.ln2[0], 0, g_resid=dx_c, g_gemm=ws["dh32"], dx_f32=dx_c, M=B)
                ws["dctx32"].zero_()
                self._ex_gemm(dx_c, lay.w3t[0], tok0(ws["dctx32"], H), EPI_F32, a3, M=B)
            else:
                if dx3:
                    self._ex_gemm3(a3, lay.w3t[2], ws["g32"], EPI_F32, M)
                else:
                    self._ex_gemm(dx, lay.w3t[2], ws["g32"], EPI_F32, a3)
                g3 = ops.dgelu_split3(ws["g32"], ws["z32s"][l], dst=a3)
                ops.gemm(g3, lay.w3t[1], ws["dh32"], EPI_F32)
                ops.layernorm_bwd(x[2 * l + 1], ws["st2"][l], lay.ln2[0], 0, g_resid=dx, g_gemm=ws["dh32"], dx_f32=dx, dx_split3=a3)
                self._ex_gemm3(a3, lay.w3t[0], ws["dctx32"], EPI_F32, M)
            ops.attn_bwd_f32(ws["qkv32s"][l], ws["dctx32"], ws["ctx32s"][l], ws["lse"][l], B, S, self.heads, scale, ws["dqkv32"],
                             dqkv_split3=a3 if l > 0 else None)      # the QKV dX GEMM's operand, written split by the attention kernel
            has = self._lora_index[l] is not None
            if has:
                ops.lora_grad_f32(ws["dqkv32"], ws["y32s"][l], M, H, self.lora_a(l), self.lora_b(l), self.lora_a(l, grad=True),
                                  self.lora_b(l, grad=True))
            if l > 0:  # nothing trainable sits below block 0
                wt = ops.split3_transpose(lay.qkv32, lay.wqkvT3, 1, lora_a=self.lora_a(l) if has else None,
                                          lora_b=self.lora_b(l) if has else None)
                self._ex_gemm3(a3, wt, ws["dh32"], EPI_F32, M)
                ops.layernorm_bwd(x[2 * l], ws["st1"][l], lay.ln1[0], 0, g_resid=dx, g_gemm=ws["dh32"], dx_f32=dx, dx_split3=a3)
                dx3 = True

# ======================================================================================================= BERT
class BertEngine(EncoderEngineBase):
    """HF-BERT trunk with LoRA on query/value (reference dna_encoder.py:40-105, language_encoder.py:24-89).

    ``head`` selects what follows the trunk:
      'mlm_softmax_mean' : cls.predictions.transform -> decoder Linear -> softmax(-1) -> mean over tokens (DNA)
      'mean_proj'        : mean over tokens -> proj Linear (text)
    """

    def __init__(self, bert, head, head_modules, device, fp8=False):
        self.fp8 = bool(fp8)
        self.device = dev = device
        cfg = getattr(bert, "config", None)
        emb = bert.embeddings
        self.H = H = emb.word_embeddings.weight.shape[1]
        assert H in (768, 512), "HIP BERT engine supports hidden sizes 768 and 512"
        self.heads = H // 64
        nh = getattr(cfg, "num_attention_heads", self.heads)
        assert nh == self.heads, "HIP attention kernel is built for head_dim 64"
        self.eps = float(getattr(cfg, "layer_norm_eps", 1e-12))
        self.p_hidden = float(getattr(cfg, "hidden_dropout_prob", 0.0))
        self.p_attn = float(getattr(cfg, "attention_probs_dropout_prob", 0.0))
        self.word, self.posw, self.typew = (_f32(emb.word_embeddings.weight, dev),
                                            _f32(emb.position_embeddings.weight, dev),
                                            _f32(emb.token_type_embeddings.weight, dev))
        self.ln_e = (_f32(emb.LayerNorm.weight, dev), _f32(emb.LayerNorm.bias, dev))
        self.layers, qv = [], []
        for layer in bert.encoder.layer:
            lay = _Layer()
            sa = layer.attention.self
            q, k, v = sa.query, sa.key, sa.value
            if hasattr(q, "w_a"):
                qv.append((q.w_a, v.w_a, q.w_b, v.w_b))
                qb, vb = q.w, v.w
            else:
                qv.append(None)
                qb, vb = q, v
            w = torch.cat([qb.weight.detach(), k.weight.detach(), vb.weight.detach()], 0)
            b = torch.cat([qb.bias.detach(), k.bias.detach(), vb.bias.detach()], 0)
            lay.waug, lay.wqkv_t, lay.b_qkv = _pack_qkv(w, b, H, dev)
            lay.src = (w, layer.attention.output.dense.weight, layer.intermediate.dense.weight, layer.output.dense.weight)   # f32 masters
            lay.w_o, lay.w_o_t, lay.b_o = _pack_linear(layer.attention.output.dense, dev)
            lay.ln_a = (_f32(layer.attention.output.LayerNorm.weight, dev), _f32(layer.attention.output.LayerNorm.bias, dev))
            lay.w_fc1, lay.w_fc1_t, lay.b_fc1 = _pack_linear(layer.intermediate.dense, dev)
            lay.w_fc2, lay.w_fc2_t, lay.b_fc2 = _pack_linear(layer.output.dense, dev)
            lay.ln_b = (_f32(layer.output.LayerNorm.weight, dev), _f32(layer.output.LayerNorm.bias, dev))
            if self.fp8:
                self._pack_fp8(lay, w, layer.intermediate.dense.weight, layer.output.dense.weight, dev)
            self.layers.append(lay)
        self.FF = self.layers[0].w_fc1.shape[0]
        self.head = head
        if head == "mlm_softmax_mean":
            tr, dec = head_modules
            self.w_tr, self.w_tr_t, self.b_tr = _pack_linear(tr.dense, dev)
            self.src_tr = tr.dense.weight
            self.ln_t = (_f32(tr.LayerNorm.weight, dev), _f32(tr.LayerNorm.bias, dev))
            self.eps_t = float(tr.LayerNorm.eps)
            trainable = [dec.weight, dec.bias]
            self.out_dim, self.head_in = dec.weight.shape
            assert self.out_dim == 768, "softmax-mean head kernel is built for 768 classes"
        else:
            (proj,) = head_modules
            trainable = [proj.weight, proj.bias]
            self.out_dim, self.head_in = proj.weight.shape
        assert self.out_dim % 128 == 0 and self.head_in % 128 == 0
        self._setup_lora(qv, H, dev, trainable)
        self.w_head_bf = torch.empty(self.out_dim, self.head_in, dtype=BF16, device=dev)
        self.w_head_t = torch.empty(self.head_in, self.out_dim, dtype=BF16, device=dev)
        self.ws = None

    def _workspace(self, B, S):
        if self.ws is not None and self.ws["B"] == B and self.ws["S"] == S:
            return self.ws
        dev, H, L, FF = self.device, self.H, len(self.layers), self.FF
        M = B * S
        z = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=dev)
        ws = {"B": B, "S": S, "M": M, "gen": next(_WS_GEN)}
        ws["emb"] = z(M, H, dt=F32)
        ws["yb"] = [z(M, H + KPAD) for _ in range(L + 1)]   # LN outputs feeding each layer's QKV GEMM (+ LoRA t)
        rb = ws["resid_bf16"] = RESID_STREAM_BF16 and not self.full_ft and not self.fp8
        sdt = BF16 if rb else F32
        ws["y"] = z(M, H, dt=F32)                           # f32 copy of the current layer input (residual; bf16 stream: head input only)
        ws["ym"] = None if rb else z(M, H, dt=F32)
        ws["ymb"] = z(M, H)
        ws["qkv"] = [z(M, 3 * H) for _ in range(L)]
        ws["ctx"] = [z(M, H) for _ in range(L)]
        ws["lse"] = [z(B, self.heads, S, dt=F32) for _ in range(L)]
        # attention-probs dropout: the forward leaves its keep decisions (1 bit per probability, 32 B per query row) for the backward,
        # which otherwise re-hashes every element (BSCLIP_ATTN_KEEP_BITS=0: the re-hashing form, same masks); 64 B per query row
        ws["kbits"] = ([torch.zeros(B * self.heads * S * ops.KEEP_WORDS, dtype=torch.int32, device=dev) for _ in range(L)]
                       if self.p_attn > 0.0 and ATTN_KEEP_BITS and not self.exact() else None)
        ws["s1"] = [z(M, H, dt=sdt) for _ in range(L)]      # pre-LN sums (LN backward inputs)
        ws["s2"] = [z(M, H, dt=sdt) for _ in range(L)]
        ws["sta"] = [z(M, 2, dt=F32) for _ in range(L)]
        ws["stb"] = [z(M, 2, dt=F32) for _ in range(L)]
        ws["z"] = [z(M, FF, dt=torch.uint8) for _ in range(L)]       # gelu'(intermediate pre-activation), 8-bit codes
        ws["act"] = z(M, FF)
        if self.full_ft:
            ws["ymbs"] = [z(M, H) for _ in range(L)]
            ws["acts"] = [z(M, FF) for _ in range(L)]
            ws["st_e"] = z(M, 2, dt=F32)
        if self.fp8:  # fp8 GEMM operands (the last LN output stays bf16: it feeds the head)
            ws["yb8"] = [z(M, H, dt=ops.FP8) for _ in range(L)]
            ws["t"] = [z(M, KPAD) for _ in range(L)]
            ws["ymb8"], ws["act8"] = z(M, H, dt=ops.FP8), z(M, FF, dt=ops.FP8)
        if self.exact():
            assert not rb and not GRAD_STREAM_BF16, "the exact forward runs on the f32 streams"
            # per layer, all f32: the layer input (ys[l]; ys[L] = the last hidden state), q | k | v, attention output, intermediate
            # pre-activation -- what the exact backward reads
            ws["ys"] = [ws["y"]] + [z(M, H, dt=F32) for _ in range(L)]
            ws["ctx32s"], ws["qkv32s"] = [z(M, H, dt=F32) for _ in range(L)], [z(M, 3 * H, dt=F32) for _ in range(L)]
            ws["z32s"] = [z(M, FF, dt=F32) for _ in range(L)]
            ws["a3"] = z(M, 3 * FF)
            ws["whead3"] = z(self.out_dim, 3 * self.head_in)
            ws["t32"], ws["mp32"] = z(M, H, dt=F32), z(B, H, dt=F32)
            ws["g32"], ws["dh32"], ws["dctx32"], ws["dqkv32"] = z(M, FF, dt=F32), z(M, H, dt=F32), z(M, H, dt=F32), z(M, 3 * H, dt=F32)
            ws["dop32"] = z(M, H, dt=F32)                       # a dX GEMM's operand when it carries a dropout mask
            ws["wheadT3"] = z(self.head_in * 3 * self.out_dim)
            rows = _pad64(M if self.head == "mlm_softmax_mean" else B)   # the head's dW reduces over tokens (MLM decoder) or sequences
            ws["t3a"], ws["t3b"] = z(self.out_dim * 3 * rows), z(self.head_in * 3 * rows)
            if self.head == "mlm_softmax_mean":
                ws["tz32"], ws["tn32"], ws["dlog32"] = z(M, H, dt=F32), z(M, H, dt=F32), z(M, self.out_dim, dt=F32)
        ws["key_bias"] = None
        ws["kb_buf"] = z(B, S, dt=F32)
        # backward temporaries
        ws["grad_bf16"] = GRAD_STREAM_BF16 and not self.full_ft
        gdt = BF16 if ws["grad_bf16"] else F32       # residual-gradient stream (see GRAD_STREAM_BF16)
        ws["ds"] = z(M, H, dt=gdt)
        ws["dsb"] = z(M, H)
        ws["ds1"] = z(M, H, dt=gdt)
        ws["dz"] = z(M, FF)
        ws["dh"] = z(M, H)
        ws["dctx"] = z(M, H)
        ws["dqkv"] = z(M, 3 * H)
        ws["dt"] = z(M, 8, dt=F32)
        ws["dtp"], ws["dbp"] = z(self.heads, 2, M, 4, dt=F32), z(B * self.heads, 2, 4, 64, dt=F32)   # LoRA partial sums (ATTN_LORA)
        if self.head == "mlm_softmax_mean":
            ws["head_splits"], Mp = split_plan(M, self.out_dim, H)
            if ws["head_splits"] > 1:
                ws["head_partial"] = z(ws["head_splits"] * self.out_dim * H, dt=F32)
            ws["tz"] = z(M, H, dt=torch.uint8)   # gelu'(transform pre-activation), 8-bit codes
            ws["tg"] = z(M, H)            # gelu(transform)
            ws["tn"] = z(M, H)            # LN(gelu(.)) = decoder input
            ws["st_t"] = z(M, 2, dt=F32)
            ws["logits"] = z(M, self.out_dim, dt=F32)
            ws["sm"] = z(M, 2, dt=F32)
            ws["dlog"] = z(M, self.out_dim)
            ws["dlog_t"] = torch.zeros(self.out_dim, Mp, dtype=BF16, device=dev)
            ws["tn_t"] = torch.zeros(H, Mp, dtype=BF16, device=dev)
            ws["dtn"] = z(M, H)
            ws["dtg"] = z(M, H)
        else:
            Bp = _pad64(B)
            ws["mp"] = z(B, H)
            ws["dout_bf"] = torch.zeros(B, self.out_dim, dtype=BF16, device=dev)
            ws["dout_t"] = torch.zeros(self.out_dim, Bp, dtype=BF16, device=dev)
            ws["mp_t"] = torch.zeros(H, Bp, dtype=BF16, device=dev)
            ws["dmp"] = z(B, H, dt=F32)
            ws["dyl"] = z(M, H, dt=F32)
        self.ws = ws
        return ws

    def _decoder_grads(self, ws, gw, gb):
        """dW_dec += dlogits^T tn (reduced over all B*S tokens: split-K, the [768, 768] output alone is 9 tiles) and db_dec += column
        sums of dlogits, which fall out of the transpose that builds the GEMM operand."""
        M, H = ws["M"], self.H
        ops.transpose_colsum_bf16(ws["dlog"], M, self.out_dim, ws["dlog_t"], gb)
        ops.transpose_bf16(ws["tn"], M, H, ws["tn_t"])
        if ws["head_splits"] > 1:
            ops.gemm_splitk_f32(ws["dlog_t"], ws["tn_t"], gw, ws["head_splits"], ws["head_partial"], K=ws["dlog_t"].shape[1])
        else:
            ops.gemm(ws["dlog_t"], ws["tn_t"], gw, EPI_RESID_F32, resid=gw)

    def _drop(self, ws, p, layer, site):
        """(p, seed) of one dropout site, or None when dropout is off (eval mode / p = 0).  The seed names the site; what
        makes the masks differ from step to step is the engine's device step word, mixed in when the kernel runs."""
        if not ws["train"] or p <= 0.0:
            return None
        return (p, (ws["drop_base"] + 0x9E3779B1 * (layer + 2) + 0x85EBCA6B * site) & 0xFFFFFFFF)

    def _begin_dropout(self, ws, advance):
        """Point this thread's dropout launches at the engine's step word (``bsclip_set_dropout_step``) and, in forward,
        advance it by one (a captured launch: a replayed hipGraph draws fresh masks, its backward re-reads the same value)."""
        if not (ws["train"] and (self.p_hidden > 0.0 or self.p_attn > 0.0)):
            ops.set_dropout_step(None)
            return
        if getattr(self, "_step_word", None) is None:
            self._step_word = torch.zeros(1, dtype=torch.int32, device=self.device)
        if advance:
            ops.counter_add(self._step_word, 1)
        ops.set_dropout_step(self._step_word)

    def anomaly_probes(self):
        """(name, tensor) of the forward's saved activations in execution order (BSCLIP_DETECT_ANOMALY)."""
        ws = self.ws
        qkv = "qkv32s" if self.exact() else "qkv"
        yield "the embedding sum (word + position + token type)", ws["emb"]
        for l in range(len(self.layers)):
            yield f"encoder.layer.{l}.attention.self q / k / v", ws[qkv][l]
            yield f"encoder.layer.{l}.attention.output (pre-LayerNorm sum)", ws["s1"][l]
            yield f"encoder.layer.{l}.output (pre-LayerNorm sum)", ws["s2"][l]
        if self.head == "mlm_softmax_mean":
            yield "cls.predictions.decoder logits", ws["logits"]

    def _forward_exact(self, ws, key_bias):
        """BSCLIP_PARITY=2 (see ViTEngine._forward_exact): the layers after the embedding LayerNorm (which has just written the f32
        layer input ys[0]) with every Linear and every attention product on split-bf16 operands (f32 in / out), exact-erf GELU; dropout sites and seeds as in
        the default path; what ``_backward_exact`` reads stays resident in f32."""
        B, S, M, H, L, FF = ws["B"], ws["S"], ws["M"], self.H, len(self.layers), self.FF
        a3, ys = ws["a3"], ws["ys"]
        for lay in self.layers:
            if not hasattr(lay, "w3"):
                lay.qkv32, lay.wqkv3 = self._ex_weight(lay.src[0], lora=True)
                lay.w3 = [self._ex_weight(w)[1] for w in lay.src[1:]]   # attention output, intermediate, output
        for l, lay in enumerate(self.layers):
            has = self._lora_index[l] is not None
            qkv32, ctx32, z32 = ws["qkv32s"][l], ws["ctx32s"][l], ws["z32s"][l]
            ops.split3_weight(lay.qkv32, lay.wqkv3, lora_a=self.lora_a(l) if has else None, lora_b=self.lora_b(l) if has else None)
            self._ex_gemm3(a3, lay.wqkv3, qkv32, EPI_F32, M, bias=lay.b_qkv)     # a3 = split of ys[l], written by the LayerNorm before
            ops.attn_fwd_f32(qkv32, B, S, self.heads, 0.125, ctx32, ws["lse"][l], key_bias=key_bias,
                             dropout=self._drop(ws, self.p_attn, l, 1), ctx_split3=a3)
            self._ex_gemm3(a3, lay.w3[0], ws["s1"][l], EPI_RESID_F32, M, bias=lay.b_o, resid=ys[l],
                           dropout=self._drop(ws, self.p_hidden, l, 2))
            ops.layernorm_fwd(ws["s1"][l], lay.ln_a[0], lay.ln_a[1], self.eps, y_f32=ws["ym"], y_split3=a3, stats=ws["sta"][l])
            self._ex_gemm3(a3, lay.w3[1], z32, EPI_F32, M, bias=lay.b_fc1)
            g3 = ops.gelu_split3(z32, a3)
            ops.gemm(g3, lay.w3[2], ws["s2"][l], EPI_RESID_F32, bias=lay.b_fc2, resid=ws["ym"],
                     dropout=self._drop(ws, self.p_hidden, l, 3))
            ops.layernorm_fwd(ws["s2"][l], lay.ln_b[0], lay.ln_b[1], self.eps, y_f32=ys[l + 1], y_split3=a3, stats=ws["stb"][l])
        out = torch.empty(B, self.out_dim, dtype=F32, device=self.device)
        ops.split3_weight(self.extra(0), ws["whead3"])
        if self.head == "mlm_softmax_mean":
            if not hasattr(self, "w_tr3"):
                self.w_tr3 = self._ex_weight(self.src_tr)[1]
            self._ex_gemm(ys[L], self.w_tr3, ws["tz32"], EPI_F32, a3, bias=self.b_tr)
            ops.gelu_split3(ws["tz32"], None, g32=ws["t32"])                                # the GELU feeds a LayerNorm: f32 out
            ops.layernorm_fwd(ws["t32"], self.ln_t[0], self.ln_t[1], self.eps_t, y_bf16=ws["tn"], y_f32=ws["tn32"], stats=ws["st_t"])
            self._ex_gemm(ws["tn32"], ws["whead3"], ws["logits"], EPI_F32, a3, bias=self.extra(1))
            ops.softmax_meanpool_fwd(ws["logits"], B, S, out, ws["sm"])
        else:
            ops.meanpool_tokens_f32(ys[L], B, S, ws["mp32"])
            self._ex_gemm(ws["mp32"], ws["whead3"], out, EPI_F32, a3, bias=self.extra(1))
        ops.set_dropout_step(None)
        return out

    def _backward_exact(self, dout):
        """BSCLIP_PARITY=2: the backward of ``_forward_exact`` in f32 (see ViTEngine._backward_exact); post-LN residuals and the
        dropout sites as in ``backward``."""
        ws = self.ws
        B, S, M, H, L = ws["B"], ws["S"], ws["M"], self.H, len(self.layers)
        self.flat.bind_grads()
        self._begin_dropout(ws, advance=False)
        self._ex_backward_weights()
        a3, ys = ws["a3"], ws["ys"]
        gw, gb = self.extra(0, grad=True), self.extra(1, grad=True)
        dout = dout.contiguous()
        wt_head = ops.split3_transpose(self.extra(0), ws["wheadT3"], 1)
        if self.head == "mlm_softmax_mean":
            if not hasattr(self, "w_tr3t"):
                self.w_tr3t = self._ex_weight_t(self.src_tr)
            ops.softmax_meanpool_bwd_f32(ws["logits"], ws["sm"], dout, B, S, ws["dlog32"])
            self._ex_dw(ws["dlog32"], ws["tn32"], gw, ws)
            ops.colsum(ws["dlog32"], M, self.out_dim, gb)
            self._ex_gemm(ws["dlog32"], wt_head, ws["dh32"], EPI_F32, a3)                   # d tn
            ops.layernorm_bwd(ws["t32"], ws["st_t"], self.ln_t[0], 0, g_gemm=ws["dh32"], dx_f32=ws["dop32"])
            g3 = ops.dgelu_split3(ws["dop32"], ws["tz32"], dst=a3)
            ops.gemm(g3, self.w_tr3t, ws["dh32"], EPI_F32)                                   # d (last hidden state)
            g_resid, g_gemm = None, ws["dh32"]
        else:
            self._ex_dw(dout, ws["mp32"], gw, ws)
            ops.colsum(dout, B, self.out_dim, gb)
            self._ex_gemm(dout, wt_head, ws["dmp"], EPI_F32, a3)
            ops.meanpool_tokens_bwd(ws["dmp"], B, S, ws["dyl"])
            g_resid, g_gemm = ws["dyl"], None
        for l in range(L - 1, -1, -1):
            lay = self.layers[l]
            drop_b, drop_a = self._drop(ws, self.p_hidden, l, 3), self._drop(ws, self.p_hidden, l, 2)
            # a dX GEMM's operand carries the mask its Linear's forward output was dropped with; without dropout it IS the stream
            # (round 5: the LayerNorm backward writes that operand split, [hi | lo | hi], straight into the GEMM's A buffer)
            ops.layernorm_bwd(ws["s2"][l], ws["stb"][l], lay.ln_b[0], 1, g_resid=g_resid, g_gemm=g_gemm, dx_f32=ws["ds"],
                              dx_split3=a3, dropout=drop_b)
            self._ex_gemm3(a3, lay.w3t[2], ws["g32"], EPI_F32, M)
            g3 = ops.dgelu_split3(ws["g32"], ws["z32s"][l], dst=a3)
            ops.gemm(g3, lay.w3t[1], ws["dh32"], EPI_F32)
            ops.layernorm_bwd(ws["s1"][l], ws["sta"][l], lay.ln_a[0], 1, g_resid=ws["ds"], g_gemm=ws["dh32"], dx_f32=ws["ds1"],
                              dx_split3=a3, dropout=drop_a)
            self._ex_gemm3(a3, lay.w3t[0], ws["dctx32"], EPI_F32, M)
            ops.attn_bwd_f32(ws["qkv32s"][l], ws["dctx32"], ws["ctx32s"][l], ws["lse"][l], B, S, self.heads, 0.125, ws["dqkv32"],
                             key_bias=ws["key_bias"], dropout=self._drop(ws, self.p_attn, l, 1), dqkv_split3=a3 if l > 0 else None)
            has = self._lora_index[l] is not None
            if has:
                ops.lora_grad_f32(ws["dqkv32"], ys[l], M, H, self.lora_a(l), self.lora_b(l), self.lora_a(l, grad=True),
                                  self.lora_b(l, grad=True))
            if l > 0:  # embeddings are frozen: nothing to do below layer 0
                wt = ops.split3_transpose(lay.qkv32, lay.wqkvT3, 1, lora_a=self.lora_a(l) if has else None,
                                          lora_b=self.lora_b(l) if has else None)
                self._ex_gemm3(a3, wt, ws["dh32"], EPI_F32, M)
                g_resid, g_gemm = ws["ds1"], ws["dh32"]
        ops.set_dropout_step(None)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None):
        B, S = input_ids.shape
        ws = self._workspace(B, S)
        H, M, L = self.H, ws["M"], len(self.layers)
        scale = 0.125
        # HF BERT dropout (hidden 0.1 / attention-probs 0.1) is active in train mode (reference train_epoch.py:20);
        # masks are functions of (seed, element index), regenerated in backward from the seeds kept here
        ws["train"] = bool(getattr(self, "training", False))
        # the rank is mixed in: ranks seeded alike (bench.py, train_cl.py) must not draw the same masks for their shards
        # ... and the engine (its head kind): the DNA and text towers share layer / site numbers and step values
        ws["drop_base"] = (torch.initial_seed() * 0x2545F491 + _rank() * 0x632BE5AB + zlib.crc32(self.head.encode())) & 0xFFFFFFFF
        self._begin_dropout(ws, advance=True)
        self.refresh_lora_weights()
        ops.cast_f32_bf16(self.extra(0), self.w_head_bf)
        key_bias = None
        if attention_mask is not None:
            # HF extended mask (App. A.3): (1 - m) * finfo.min added to the scores of padded keys.
            key_bias = ws["kb_buf"]
            ops.mask_to_bias(attention_mask.to(torch.int64).contiguous(), key_bias)
        ws["key_bias"] = key_bias
        ops.bert_embed(input_ids.contiguous(), None if token_type_ids is None else token_type_ids.contiguous(),
                       self.word, self.posw, self.typew, ws["emb"])
        f8 = self.fp8
        rb = ws["resid_bf16"]   # bf16 residual stream: the LN's bf16 operand IS the residual, no f32 copy is written
        EPI_R = EPI_RESID_BF16 if rb else EPI_RESID_F32
        if f8:
            ops.layernorm_fwd_fp8(ws["emb"], self.ln_e[0], self.ln_e[1], self.eps, ws["yb8"][0], t_aug=ws["t"][0], y_f32=ws["y"],
                                  lora_a=self.lora_a(0), dropout=self._drop(ws, self.p_hidden, -1, 0))
        else:
            ops.layernorm_fwd(ws["emb"], self.ln_e[0], self.ln_e[1], self.eps, y_bf16=ws["yb"][0], y_f32=None if rb else ws["y"],
                              lora_a=self.lora_a(0), dropout=self._drop(ws, self.p_hidden, -1, 0),
                              stats=ws["st_e"] if self.full_ft else None,
                              y_split3=ws["a3"] if self.exact() else None)     # exact mode: layer 0's QKV operand, already split
        ws["ids"], ws["type_ids"] = input_ids, token_type_ids
        if self.exact():
            return self._forward_exact(ws, key_bias)
        for l, lay in enumerate(self.layers):
            if f8:
                ops.gemm_fp8(ws["yb8"][l], lay.w_qkv8, ws["qkv"][l], lay.s_qkv, lay.b_qkv, EPI_BF16, a_aug=ws["t"][l],
                             b_aug=lay.baug)
            else:
                ops.gemm(ws["yb"][l], lay.waug, ws["qkv"][l], EPI_BF16, bias=lay.b_qkv)
            drop_p = self._drop(ws, self.p_attn, l, 1)
            ops.attn_fwd(ws["qkv"][l], B, S, self.heads, scale, ws["ctx"][l], ws["lse"][l], key_bias=key_bias, dropout=drop_p,
                         keep_bits=ws["kbits"][l] if drop_p is not None and ws["kbits"] is not None else None)
            ops.gemm(ws["ctx"][l], lay.w_o, ws["s1"][l], EPI_R, bias=lay.b_o, resid=ws["yb"][l] if rb else ws["y"],
                     dropout=self._drop(ws, self.p_hidden, l, 2))
            if f8:
                ops.layernorm_fwd_fp8(ws["s1"][l], lay.ln_a[0], lay.ln_a[1], self.eps, ws["ymb8"], y_f32=ws["ym"],
                                      stats=ws["sta"][l])
                ops.gemm_fp8(ws["ymb8"], lay.w_fc1_8, ws["act8"], lay.s_fc1, lay.b_fc1, EPI_GELU_FP8, aux=ws["z"][l])
                ops.gemm_fp8(ws["act8"], lay.w_fc2_8, ws["s2"][l], lay.s_fc2, lay.b_fc2, EPI_RESID_F32, resid=ws["ym"],
                             dropout=self._drop(ws, self.p_hidden, l, 3))
            else:
                ymb, act = (ws["ymbs"][l], ws["acts"][l]) if self.full_ft else (ws["ymb"], ws["act"])
                ops.layernorm_fwd(ws["s1"][l], lay.ln_a[0], lay.ln_a[1], self.eps, y_bf16=ymb, y_f32=ws["ym"],
                                  stats=ws["sta"][l])
                ops.gemm(ymb, lay.w_fc1, act, EPI_GELU_BF16, bias=lay.b_fc1, aux=ws["z"][l])
                ops.gemm(act, lay.w_fc2, ws["s2"][l], EPI_R, bias=lay.b_fc2, resid=ymb if rb else ws["ym"],
                         dropout=self._drop(ws, self.p_hidden, l, 3))
            nxt = self.lora_a(l + 1) if l + 1 < L else self._zero_a
            if f8 and l + 1 < L:
                ops.layernorm_fwd_fp8(ws["s2"][l], lay.ln_b[0], lay.ln_b[1], self.eps, ws["yb8"][l + 1], t_aug=ws["t"][l + 1],
                                      y_f32=ws["y"], lora_a=nxt, stats=ws["stb"][l])
            else:
                # (bf16 stream: the f32 copy is written for the last layer only, where the mean-pool head reads it)
                ops.layernorm_fwd(ws["s2"][l], lay.ln_b[0], lay.ln_b[1], self.eps, y_bf16=ws["yb"][l + 1],
                                  y_f32=None if rb and not (l + 1 == L and self.head != "mlm_softmax_mean") else ws["y"],
                                  lora_a=nxt, stats=ws["stb"][l])
        out = torch.empty(B, self.out_dim, dtype=F32, device=self.device)
        if self.head == "mlm_softmax_mean":
            ops.gemm(ws["yb"][L], self.w_tr, ws["tg"], EPI_GELU_BF16, bias=self.b_tr, aux=ws["tz"], K=H)
            ops.layernorm_fwd(ws["tg"], self.ln_t[0], self.ln_t[1], self.eps_t, y_bf16=ws["tn"], stats=ws["st_t"])
            ops.gemm(ws["tn"], self.w_head_bf, ws["logits"], EPI_F32, bias=self.extra(1))
            ops.softmax_meanpool_fwd(ws["logits"], B, S, out, ws["sm"])
        else:
            ops.meanpool_tokens_fwd(ws["y"], B, S, ws["mp"])
            ops.gemm(ws["mp"], self.w_head_bf, out, EPI_F32, bias=self.extra(1))
        ops.set_dropout_step(None)   # the pointer is per thread: do not leak it into the caller's own launches
        return out

    def backward(self, dout):
        if self.exact():
            return self._backward_exact(dout)
        ws = self.ws
        B, S, M, H, L = ws["B"], ws["S"], ws["M"], self.H, len(self.layers)
        scale = 0.125
        self.flat.bind_grads()
        self._begin_dropout(ws, advance=False)   # backward runs on autograd's thread: the pointer is per thread
        gw, gb = self.extra(0, grad=True), self.extra(1, grad=True)
        ops.transpose_bf16(self.w_head_bf, self.out_dim, self.head_in, self.w_head_t)
        if self.head == "mlm_softmax_mean":
            ops.softmax_meanpool_bwd(ws["logits"], ws["sm"], dout, B, S, ws["dlog"])
            self._decoder_grads(ws, gw, gb)
            ops.gemm(ws["dlog"], self.w_head_t, ws["dtn"], EPI_BF16)                   # d tn
            ops.layernorm_bwd(ws["tg"], ws["st_t"], self.ln_t[0], 0, g_gemm=ws["dtn"], dx_bf16=ws["dtg"])
            ops.dgelu_mul(ws["dtg"], ws["tz"], M, H, ws["dtg"])
            ops.gemm(ws["dtg"], self.w_tr_t, ws["dh"], EPI_BF16)                       # d (last hidden state)
            g_resid, g_gemm = None, ws["dh"]
        else:
            ops.cast_f32_bf16(dout, ws["dout_bf"])
            ops.transpose_bf16(ws["dout_bf"], B, self.out_dim, ws["dout_t"])
            ops.transpose_bf16(ws["mp"], B, H, ws["mp_t"])
            ops.gemm(ws["dout_t"], ws["mp_t"], gw, EPI_RESID_F32, resid=gw)            # dW_proj += dout^T mean
            ops.colsum(dout, B, self.out_dim, gb)
            ops.gemm(ws["dout_bf"], self.w_head_t, ws["dmp"], EPI_F32)
            ops.meanpool_tokens_bwd(ws["dmp"], B, S, ws["dyl"])
            g_resid, g_gemm = ws["dyl"], None
        dt_in, a_in = None, None
        g16 = ws["grad_bf16"]
        for l in range(L - 1, -1, -1):
            lay = self.layers[l]
            # dsb is the operand of fc2's dX GEMM: it carries the mask fc2's forward output was dropped with.  With the bf16
            # gradient stream and no dropout the operand IS the residual gradient: one output, rewritten in place next time
            drop_b, drop_a = self._drop(ws, self.p_hidden, l, 3), self._drop(ws, self.p_hidden, l, 2)
            one_b, one_a = g16 and drop_b is None, g16 and drop_a is None
            ops.layernorm_bwd(ws["s2"][l], ws["stb"][l], lay.ln_b[0], 1, g_resid=g_resid, g_gemm=g_gemm, dt=dt_in,
                              lora_a=a_in, dx_f32=None if one_b else ws["ds"], dx_bf16=ws["dsb"], dropout=drop_b)
            ops.gemm(ws["dsb"], lay.w_fc2_t, ws["dz"], EPI_DGELU_BF16, aux=ws["z"][l])
            ops.gemm(ws["dz"], lay.w_fc1_t, ws["dh"], EPI_BF16)
            ops.layernorm_bwd(ws["s1"][l], ws["sta"][l], lay.ln_a[0], 1, g_resid=ws["dsb"] if one_b else ws["ds"], g_gemm=ws["dh"],
                              dx_f32=None if one_a else ws["ds1"], dx_bf16=ws["dsb"], dropout=drop_a)
            ops.gemm(ws["dsb"], lay.w_o_t, ws["dctx"], EPI_BF16)
            drop_p = self._drop(ws, self.p_attn, l, 1)
            lb = self.lora_b(l)
            kb = ws["kbits"][l] if drop_p is not None and ws["kbits"] is not None else None
            part = lb is not None and ATTN_LORA and not self.fp8 and (drop_p is None or kb is not None)
            ops.attn_bwd(ws["qkv"][l], ws["dctx"], ws["lse"][l], B, S, self.heads, scale, ws["dqkv"],
                         key_bias=ws["key_bias"], dropout=drop_p, keep_bits=kb,
                         lora=(ws["yb"][l][:, H:], lb, ws["dtp"], ws["dbp"]) if part else None)
            if lb is not None:
                gbb = self.lora_b(l, grad=True)
                if self.fp8:
                    ops.lora_grad_fp8(ws["dqkv"], ws["yb8"][l], ws["t"][l], M, H, lb, ws["dt"], self.lora_a(l, grad=True),
                                      gbb[0], gbb[1])
                elif part:
                    ops.lora_grad_heads(ws["yb"][l], M, H, B, ws["dtp"], ws["dbp"], ws["dt"], self.lora_a(l, grad=True), gbb[0], gbb[1])
                else:
                    ops.lora_grad(ws["dqkv"], ws["yb"][l], M, H, lb, ws["dt"], self.lora_a(l, grad=True), gbb[0], gbb[1])
            if l > 0:  # embeddings are frozen: nothing to do below layer 0
                ops.gemm(ws["dqkv"], lay.wqkv_t, ws["dh"], EPI_BF16)
                g_resid, g_gemm = (ws["dsb"] if one_a else ws["ds1"]), ws["dh"]
                dt_in, a_in = (ws["dt"], self.lora_a(l)) if lb is not None else (None, None)
        ops.set_dropout_step(None)


# ====================================================================================== autograd integration
_PARENT_STREAM = []  # stack: the stream a tower's side stream was forked from (set by SimpleCLIP.forward)


class forked_from:
    """``with forked_from(stream):`` -- encoders run inside (on a side stream) make ``stream`` wait for their BACKWARD.

    Autograd replays a node on the stream of its forward and, when backward() returns, only joins the streams on which it
    accumulated leaf gradients itself.  The encoder nodes write their parameter gradients from inside the kernels (they
    return ``None`` for the parameters), so autograd would not join the tower streams: the optimizer on the main stream
    could read the flat gradient buffer while the last blocks' backward kernels are still running."""

    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        _PARENT_STREAM.append(self.stream)

    def __exit__(self, *exc):
        _PARENT_STREAM.pop()


class _EncoderFn(torch.autograd.Function):
    """One autograd node per encoder: forward and backward are whole kernel sequences.  Trainable parameters are
    passed only so autograd records the node; their gradients are accumulated in place by the kernels (into the
    flat buffer ``p.grad`` aliases), hence ``None`` is returned for them."""

    @staticmethod
    def forward(ctx, engine, fwd_args, *params):
        ctx.engine = engine
        ctx.parent = _PARENT_STREAM[-1] if _PARENT_STREAM else None
        ctx.generation = engine._fwd_generation
        out = engine.forward(*fwd_args)
        if DETECT_ANOMALY:
            _anomaly_check(engine, "forward", out)
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.generation != ctx.engine._fwd_generation:
            raise RuntimeError("backward through an encoder forward whose saved activations were overwritten by a later forward "
                               "of the same encoder: the HIP engines hold one workspace per encoder -- run backward before the "
                               "next training-mode forward (the reference loop does: train_epoch.py:28-42)")
        ctx.engine.backward(dout.contiguous())
        if DETECT_ANOMALY:
            _anomaly_check(ctx.engine, "backward")
        from .dist import start_allreduce
        start_allreduce(ctx.engine.flat, type(ctx.engine).__name__)  # global-batch step: this tower's gradients are complete on this stream
        here = torch.cuda.current_stream()
        # gradients are complete before anything queued afterwards on the stream the tower was forked from -- and, outside a
        # stream capture, on the default stream (inside a capture that wait would pull the default stream into the graph and
        # leave it unjoined)
        parents = {ctx.parent} if torch.cuda.is_current_stream_capturing() else {ctx.parent, torch.cuda.default_stream(here.device)}
        for parent in parents:
            if parent is not None and parent != here:
                parent.wait_stream(here)
        return (None, None) + (None,) * len(ctx.engine.flat.params)


def _frozen_signature(module, eng):
    """(storage address, in-place version) of every tensor the engine packed into its own bf16 layouts.  An in-place
    ``load_state_dict`` / ``copy_`` after the engine was built bumps the version, ``.to()`` moves the storage."""
    mine = {id(p) for p in eng.flat.params}
    return tuple((t.data_ptr(), t._version) for t in list(module.parameters()) + list(module.buffers())
                 if id(t) not in mine)


def wants_fp8(module):
    """``module.hip_precision = "fp8"`` (set_precision) selects the fp8 frozen-trunk GEMMs (BASELINE configs[4])."""
    return getattr(module, "hip_precision", "bf16") == "fp8"


def set_precision(model, precision):
    """Select "bf16" (default) or "fp8" frozen-trunk GEMMs for the ViT and BarcodeBERT encoders under ``model``; engines
    are rebuilt on the next forward (the text tower stays bf16: its GEMMs are latency-sized)."""
    if precision not in ("bf16", "fp8"):
        raise ValueError(f"precision must be 'bf16' or 'fp8', not {precision!r}")
    for m in model.modules():
        if hasattr(m, "lora_vit") or hasattr(m, "lora_barcode_bert"):
            m.hip_precision = precision
            m._engine = None


def wants_full_ft(module):
    """``module.hip_full_ft`` (set by load_clip_model for ``disable_lora: true``): every parameter is trained."""
    return bool(getattr(module, "hip_full_ft", False))


def _engine_for(module, build):
    eng = getattr(module, "_engine", None)
    if eng is not None and (eng.fp8 != wants_fp8(module) or eng.full_ft != wants_full_ft(module)):
        eng = None
    if eng is not None and eng.flat.valid() and _frozen_signature(module, eng) != eng._frozen_sig:
        eng = None  # frozen weights were overwritten (checkpoint loaded after the first forward): repack
    if eng is None or not eng.flat.valid():
        if not torch.cuda.is_available():
            raise RuntimeError("bioscanclip needs a ROCm GPU: all arithmetic runs in libbsclip_hip.so "
                               "(there is no CPU/torch fallback)")
        eng = build()
        eng._frozen_sig = _frozen_signature(module, eng)
        module._engine = eng
    return eng


def run_encoder(module, build, fwd_args):
    eng = _engine_for(module, build)
    eng.training = module.training
    eng._owner = module
    # the engine keeps ONE set of saved activations (its workspace): any later forward of the same encoder, with or without
    # autograd, overwrites them -- _EncoderFn.backward checks that it still owns them
    eng._fwd_generation = getattr(eng, "_fwd_generation", 0) + 1
    if torch.is_grad_enabled() and any(p.requires_grad for p in eng.flat.params):
        return _EncoderFn.apply(eng, fwd_args, *eng.flat.params)
    out = eng.forward(*fwd_args)
    if DETECT_ANOMALY:
        _anomaly_check(eng, "forward", out)
    return out
