"""Full fine-tuning engines: ``disable_lora: true`` of the reference (bioscanclip/model/simple_clip.py:125-203,
config/model_config/full_fine_tuning/**) -- every parameter of the towers is trained (:199-201), the BERT encoders carry no
LoRA branch and the ViT keeps LoRA on all blocks (the reference's own quirk: an empty ``lora_layer`` list is falsy in
image_encoder.py:56-59 and means "all layers").

The forward pass is the LoRA-regime one (same kernels; the bf16 operand copies of the weights are re-packed from the f32
masters every step, and the inputs of fc1 / fc2 are kept per layer).  The backward pass adds, next to every dX GEMM, the
gradients the LoRA regime never needs:
  * Linear weights: dW = dY^T X on the same MFMA GEMM with both operands transposed (as the trainable heads always did),
    accumulated in f32 into the flat gradient buffer; biases by ``bsclip_colsum``;
  * LayerNorm gains / biases (``bsclip_ln_param_grad``), BertEmbeddings tables (``bsclip_embed_grad``), ViT patch filters
    (``bsclip_gather_cast_rows`` + the dW GEMM), cls token and position table (ordered column sums);
  * the dX chain continues through layer 0 into the embeddings.
All trainable tensors live in one flat f32 buffer per engine; q / k / v weights (and biases) of a BERT layer are adjacent in
it, so the fused [3H, H] operand and its gradient are plain views.
"""
import torch

from . import engine, ops
from .engine import BF16, F32, BertEngine, ViTEngine, split_plan
from .lib import EPI_BF16, EPI_DGELU_BF16, EPI_F32, EPI_RESID_F32, KPAD


class _FTMixin:
    full_ft = True
    _partial = None

    # ------------------------------------------------------------------------------------------------ flat-buffer access
    def tp(self, name, shape=None, n=None, grad=False):
        """View of trunk tensor ``name`` (or of ``n`` elements starting at it) in the flat data / gradient buffer."""
        i = self._trunk_index + self._t[name]
        if shape is None and n is None:
            shape = self.flat.params[i].shape
        return self.flat.view(i, shape, n=n, grad=grad)

    def _register(self, named):
        self._t = {name: i for i, (name, _) in enumerate(named)}
        return [p for _, p in named]

    # --------------------------------------------------------------------------------------------- weight re-packing
    def _repack(self):
        """bf16 operand copies (and their transposes for the dX GEMMs) from the f32 masters; once per forward."""
        for w32, dst, dst_t, aug in self._packs:
            if aug is None:
                ops.cast_f32_bf16(w32, dst)
                if dst_t is not None:
                    ops.transpose_bf16(dst, dst.shape[0], dst.shape[1], dst_t)
            else:  # fused QKV: the forward operand is the [3H, H + KPAD] K-augmented matrix
                ops.cast_f32_bf16(w32, self._wtmp[: w32.shape[0]])
                ops.transpose_bf16(self._wtmp[: w32.shape[0]], w32.shape[0], w32.shape[1], dst_t)
                ops.transpose_bf16(dst_t, w32.shape[1], w32.shape[0], aug)

    def refresh_lora_weights(self):
        self._repack()
        super().refresh_lora_weights()

    # ------------------------------------------------------------------------------------------------ weight gradients
    def _tbuf(self, tag, rows, M, Mp):
        """Zero-padded transposed operand [rows, Mp] of a weight-gradient GEMM over M tokens.  Keyed by M as well: the columns
        [M, Mp) must stay zero, and two GEMMs with different token counts (the ViT blocks' B * 197 and the patch embedding's
        B * 196 round to the same Mp) would otherwise leave each other's last columns behind."""
        key = (tag, rows, M, Mp)
        b = self._tbufs.get(key)
        if b is None:
            b = self._tbufs[key] = torch.zeros(rows, Mp, dtype=BF16, device=self.device)
        return b

    def _dw(self, dY, X, M, N, K, wname, bname, w_n=None):
        """grad(W[N,K]) += dY[:M,:N]^T X[:M,:K];  grad(b[N]) += column sums of dY."""
        S, Mp = split_plan(M, N, K)
        tA, tB = self._tbuf("A", N, M, Mp), self._tbuf("B", K, M, Mp)
        if bname is not None:   # the bias gradient falls out of the transpose's tiles
            ops.transpose_colsum_bf16(dY, M, N, tA, self.tp(bname, n=N if w_n is not None else None, grad=True))
        else:
            ops.transpose_bf16(dY, M, N, tA)
        ops.transpose_bf16(X, M, K, tB)
        gW = self.tp(wname, (N, K), n=w_n, grad=True)
        if S > 1:
            if self._partial is None:
                self._partial = torch.empty(512 * 65536, dtype=F32, device=self.device)   # splits * tiles <= 512 slabs of 256 x 256
            ops.gemm_splitk_f32(tA, tB, gW, S, self._partial, K=Mp)
        else:
            ops.gemm(tA, tB, gW, EPI_RESID_F32, resid=gW, K=Mp)


# ============================================================================================================== ViT
class ViTEngineFT(_FTMixin, ViTEngine):
    def __init__(self, module, device):
        self._module = module
        self._tbufs = {}
        super().__init__(module, device, fp8=False)
        vit = module.lora_vit
        H = self.H
        # live f32 views (the parameters were re-homed into the flat buffer by the base constructor)
        self.cls = vit.cls_token.data.view(-1)
        self.pos = vit.pos_embed.data.view(self.S, H)
        self.b_patch = vit.patch_embed.proj.bias.data
        self.ln_f = (vit.norm.weight.data, vit.norm.bias.data)
        self._wtmp = torch.empty(3 * H, H, dtype=BF16, device=device)
        self._packs = [(vit.patch_embed.proj.weight.data.view(H, -1), self.w_patch, None, None)]
        for blk, lay in zip(vit.blocks, self.layers):
            q = blk.attn.qkv
            base = q.qkv if hasattr(q, "linear_a_q") else q
            lay.b_qkv = base.bias.data
            lay.ln1 = (blk.norm1.weight.data, blk.norm1.bias.data)
            lay.ln2 = (blk.norm2.weight.data, blk.norm2.bias.data)
            lay.b_proj, lay.b_fc1, lay.b_fc2 = blk.attn.proj.bias.data, blk.mlp.fc1.bias.data, blk.mlp.fc2.bias.data
            self._packs += [(base.weight.data, None, lay.wqkv_t, lay.waug[:, :H]),
                            (blk.attn.proj.weight.data, lay.w_proj, lay.w_proj_t, None),
                            (blk.mlp.fc1.weight.data, lay.w_fc1, lay.w_fc1_t, None),
                            (blk.mlp.fc2.weight.data, lay.w_fc2, lay.w_fc2_t, None)]

    def _trunk_trainables(self):
        vit = self._module.lora_vit
        named = [("cls", vit.cls_token), ("pos", vit.pos_embed), ("patch.w", vit.patch_embed.proj.weight),
                 ("patch.b", vit.patch_embed.proj.bias)]
        for i, blk in enumerate(vit.blocks):
            q = blk.attn.qkv
            base = q.qkv if hasattr(q, "linear_a_q") else q
            named += [(f"{i}.n1.w", blk.norm1.weight), (f"{i}.n1.b", blk.norm1.bias), (f"{i}.qkv.w", base.weight),
                      (f"{i}.qkv.b", base.bias), (f"{i}.proj.w", blk.attn.proj.weight), (f"{i}.proj.b", blk.attn.proj.bias),
                      (f"{i}.n2.w", blk.norm2.weight), (f"{i}.n2.b", blk.norm2.bias), (f"{i}.fc1.w", blk.mlp.fc1.weight),
                      (f"{i}.fc1.b", blk.mlp.fc1.bias), (f"{i}.fc2.w", blk.mlp.fc2.weight), (f"{i}.fc2.b", blk.mlp.fc2.bias)]
        named += [("norm.w", vit.norm.weight), ("norm.b", vit.norm.bias)]
        return self._register(named)

    def _workspace(self, B):
        fresh = self.ws is None or self.ws["B"] != B
        ws = super()._workspace(B)
        if fresh:
            ws["dyp"] = torch.empty(B * 196, self.H, dtype=BF16, device=self.device)   # patch rows of d x0, bf16
        return ws

    def backward(self, dout):
        ws = self.ws
        B, M, H, S, FF = ws["B"], ws["M"], self.H, self.S, self.FF
        scale = 64 ** -0.5
        self.flat.bind_grads()
        x = ws["x"]
        dx, dxb = ws["dx"], ws["dxb"]
        tok0 = lambda t, w: t.view(B, S * w)[:, :w]
        # ---- head (trainable in both regimes) ----
        ops.cast_f32_bf16(dout, ws["dout_bf"])
        ops.transpose_bf16(ws["dout_bf"], B, self.out_dim, ws["dout_t"])
        ops.transpose_bf16(ws["clsn"], B, H, ws["clsn_t"])
        gw = self.extra(0, grad=True)
        ops.gemm(ws["dout_t"], ws["clsn_t"], gw, EPI_RESID_F32, resid=gw)
        ops.colsum(dout, B, self.out_dim, self.extra(1, grad=True))
        ops.transpose_bf16(self.w_head_bf, self.out_dim, H, self.w_head_t)
        ops.gemm(ws["dout_bf"], self.w_head_t, ws["dclsn"], EPI_BF16)
        # ---- final norm on the token-0 rows ----
        dx.zero_()
        dxb.zero_()
        ops.ln_param_grad(tok0(x[-1], H), ws["st_f"], 0, self.tp("norm.w", grad=True), self.tp("norm.b", grad=True),
                          g_gemm=ws["dclsn"])
        ops.layernorm_bwd(tok0(x[-1], H), ws["st_f"], self.ln_f[0], 0, g_gemm=ws["dclsn"], dx_f32=tok0(dx, H), dx_bf16=tok0(dxb, H))
        L = len(self.layers)
        for l in range(L - 1, -1, -1):
            lay = self.layers[l]
            n2w, n2b = self.tp(f"{l}.n2.w", grad=True), self.tp(f"{l}.n2.b", grad=True)
            if l == L - 1:   # only token 0 of the last block carries gradient (see ViTEngine.backward)
                dxb_c, dx_c = tok0(dxb, H), tok0(dx, H)
                self._dw(dxb_c, ws["act_c"], B, H, FF, f"{l}.fc2.w", f"{l}.fc2.b")
                ops.gemm(dxb_c, lay.w_fc2_t, ws["dz_c"], EPI_DGELU_BF16, aux=ws["z_c"])
                self._dw(ws["dz_c"], ws["h2_c"], B, FF, H, f"{l}.fc1.w", f"{l}.fc1.b")
                ops.gemm(ws["dz_c"], lay.w_fc1_t, ws["dh_c"], EPI_BF16)
                ops.ln_param_grad(tok0(x[2 * l + 1], H), ws["st_c"], 0, n2w, n2b, g_gemm=ws["dh_c"])
                ops.layernorm_bwd(tok0(x[2 * l + 1], H), ws["st_c"], lay.ln2[0], 0, g_resid=dx_c, g_gemm=ws["dh_c"], dx_f32=dx_c,
                                  dx_bf16=dxb_c)
                self._dw(dxb_c, tok0(ws["ctx"][l], H), B, H, H, f"{l}.proj.w", f"{l}.proj.b")
                ws["dctx"].zero_()
                ops.gemm(dxb_c, lay.w_proj_t, tok0(ws["dctx"], H), EPI_BF16)
            else:
                self._dw(dxb, ws["acts"][l], M, H, FF, f"{l}.fc2.w", f"{l}.fc2.b")
                ops.gemm(dxb, lay.w_fc2_t, ws["dz"], EPI_DGELU_BF16, aux=ws["z"][l])
                self._dw(ws["dz"], ws["h2s"][l], M, FF, H, f"{l}.fc1.w", f"{l}.fc1.b")
                ops.gemm(ws["dz"], lay.w_fc1_t, ws["dh"], EPI_BF16)
                ops.ln_param_grad(x[2 * l + 1], ws["st2"][l], 0, n2w, n2b, g_gemm=ws["dh"])
                ops.layernorm_bwd(x[2 * l + 1], ws["st2"][l], lay.ln2[0], 0, g_resid=dx, g_gemm=ws["dh"], dx_f32=dx, dx_bf16=dxb)
                self._dw(dxb, ws["ctx"][l], M, H, H, f"{l}.proj.w", f"{l}.proj.b")
                ops.gemm(dxb, lay.w_proj_t, ws["dctx"], EPI_BF16)
            lb = self.lora_b(l)
            part = lb is not None and engine.ATTN_LORA    # the LoRA regime's path: dt / dB partial sums out of the attention backward
            ops.attn_bwd(ws["qkv"][l], ws["dctx"], ws["lse"][l], B, S, self.heads, scale, ws["dqkv"], q_rows=1 if l == L - 1 else 0,
                         lora=(ws["h1"][l][:, H:], lb, ws["dtp"], ws["dbp"]) if part else None)
            self._dw(ws["dqkv"], ws["h1"][l], M, 3 * H, H, f"{l}.qkv.w", f"{l}.qkv.b")
            if lb is not None:
                gb = self.lora_b(l, grad=True)
                if part:
                    ops.lora_grad_heads(ws["h1"][l], M, H, B, ws["dtp"], ws["dbp"], ws["dt"], self.lora_a(l, grad=True), gb[0], gb[1])
                else:
                    ops.lora_grad(ws["dqkv"], ws["h1"][l], M, H, lb, ws["dt"], self.lora_a(l, grad=True), gb[0], gb[1])
            # the chain continues through block 0: patch filters, cls token and position table are trained too
            ops.gemm(ws["dqkv"], lay.wqkv_t, ws["dh"], EPI_BF16)
            dt, la = (ws["dt"], self.lora_a(l)) if lb is not None else (None, None)
            ops.ln_param_grad(x[2 * l], ws["st1"][l], 0, self.tp(f"{l}.n1.w", grad=True), self.tp(f"{l}.n1.b", grad=True),
                              g_gemm=ws["dh"], dt=dt, lora_a=la)
            ops.layernorm_bwd(x[2 * l], ws["st1"][l], lay.ln1[0], 0, g_resid=dx, g_gemm=ws["dh"], dt=dt, lora_a=la, dx_f32=dx,
                              dx_bf16=dxb)
        # ---- d x0 [B, 197, H]: x0[b, 0] = cls + pos[0], x0[b, 1 + p] = patch_p W^T + b + pos[1 + p] ----
        ops.colsum(dx.view(B, S * H), B, S * H, self.tp("pos", grad=True).view(-1))      # ordered sum over the batch
        ops.colsum(tok0(dx, H), B, H, self.tp("cls", grad=True).view(-1))
        ops.gather_cast_rows(dx, B * 196, S, 196, 1, ws["dyp"])
        self._dw(ws["dyp"], ws["cols"], B * 196, H, ws["cols"].shape[1], "patch.w", "patch.b")


# ============================================================================================================= BERT
class BertEngineFT(_FTMixin, BertEngine):
    def __init__(self, module_bert, head, head_modules, device):
        self._module = module_bert
        self._head_modules = head_modules
        self._head_kind = head
        self._tbufs = {}
        super().__init__(module_bert, head, head_modules, device, fp8=False)
        bert, H = module_bert, self.H
        emb = bert.embeddings
        self.word, self.posw = emb.word_embeddings.weight.data, emb.position_embeddings.weight.data
        self.typew = emb.token_type_embeddings.weight.data
        self.ln_e = (emb.LayerNorm.weight.data, emb.LayerNorm.bias.data)
        pid = getattr(emb.word_embeddings, "padding_idx", None)    # HF: padding_idx = config.pad_token_id = 0
        self.pad_id = -1 if pid is None else int(pid)
        self._wtmp = torch.empty(3 * H, H, dtype=BF16, device=device)
        self._packs = []
        for i, (layer, lay) in enumerate(zip(bert.encoder.layer, self.layers)):
            lay.b_qkv = self.tp(f"{i}.q.b", n=3 * H)
            lay.b_o = layer.attention.output.dense.bias.data
            lay.ln_a = (layer.attention.output.LayerNorm.weight.data, layer.attention.output.LayerNorm.bias.data)
            lay.b_fc1, lay.b_fc2 = layer.intermediate.dense.bias.data, layer.output.dense.bias.data
            lay.ln_b = (layer.output.LayerNorm.weight.data, layer.output.LayerNorm.bias.data)
            self._packs += [(self.tp(f"{i}.q.w", (3 * H, H), n=3 * H * H), None, lay.wqkv_t, lay.waug[:, :H]),
                            (layer.attention.output.dense.weight.data, lay.w_o, lay.w_o_t, None),
                            (layer.intermediate.dense.weight.data, lay.w_fc1, lay.w_fc1_t, None),
                            (layer.output.dense.weight.data, lay.w_fc2, lay.w_fc2_t, None)]
        if head == "mlm_softmax_mean":
            tr = head_modules[0]
            self.b_tr = tr.dense.bias.data
            self.ln_t = (tr.LayerNorm.weight.data, tr.LayerNorm.bias.data)
            self._packs.append((tr.dense.weight.data, self.w_tr, self.w_tr_t, None))

    def _trunk_trainables(self):
        bert = self._module
        emb = bert.embeddings
        named = [("word", emb.word_embeddings.weight), ("posw", emb.position_embeddings.weight),
                 ("typew", emb.token_type_embeddings.weight), ("lne.w", emb.LayerNorm.weight), ("lne.b", emb.LayerNorm.bias)]
        for i, layer in enumerate(bert.encoder.layer):
            sa = layer.attention.self
            lin = lambda m: m.w if hasattr(m, "w_a") else m
            q, k, v = lin(sa.query), lin(sa.key), lin(sa.value)
            # q / k / v adjacent: the fused [3H, H] operand, its bias and both gradients are plain views of the flat buffers
            named += [(f"{i}.q.w", q.weight), (f"{i}.k.w", k.weight), (f"{i}.v.w", v.weight), (f"{i}.q.b", q.bias),
                      (f"{i}.k.b", k.bias), (f"{i}.v.b", v.bias), (f"{i}.o.w", layer.attention.output.dense.weight),
                      (f"{i}.o.b", layer.attention.output.dense.bias), (f"{i}.lna.w", layer.attention.output.LayerNorm.weight),
                      (f"{i}.lna.b", layer.attention.output.LayerNorm.bias), (f"{i}.fc1.w", layer.intermediate.dense.weight),
                      (f"{i}.fc1.b", layer.intermediate.dense.bias), (f"{i}.fc2.w", layer.output.dense.weight),
                      (f"{i}.fc2.b", layer.output.dense.bias), (f"{i}.lnb.w", layer.output.LayerNorm.weight),
                      (f"{i}.lnb.b", layer.output.LayerNorm.bias)]
        if self._head_kind == "mlm_softmax_mean":
            tr = self._head_modules[0]
            named += [("tr.w", tr.dense.weight), ("tr.b", tr.dense.bias), ("lnt.w", tr.LayerNorm.weight), ("lnt.b", tr.LayerNorm.bias)]
        return self._register(named)

    def _workspace(self, B, S):
        fresh = self.ws is None or self.ws["B"] != B or self.ws["S"] != S
        ws = super()._workspace(B, S)
        if fresh:
            ws["demb"] = torch.empty(B * S, self.H, dtype=F32, device=self.device)
        return ws

    def backward(self, dout):
        ws = self.ws
        B, S, M, H, L, FF = ws["B"], ws["S"], ws["M"], self.H, len(self.layers), self.FF
        scale = 0.125
        self.flat.bind_grads()
        self._begin_dropout(ws, advance=False)
        gw, gb = self.extra(0, grad=True), self.extra(1, grad=True)
        ops.transpose_bf16(self.w_head_bf, self.out_dim, self.head_in, self.w_head_t)
        if self.head == "mlm_softmax_mean":
            ops.softmax_meanpool_bwd(ws["logits"], ws["sm"], dout, B, S, ws["dlog"])
            self._decoder_grads(ws, gw, gb)
            ops.gemm(ws["dlog"], self.w_head_t, ws["dtn"], EPI_BF16)
            ops.ln_param_grad(ws["tg"], ws["st_t"], 0, self.tp("lnt.w", grad=True), self.tp("lnt.b", grad=True), g_gemm=ws["dtn"])
            ops.layernorm_bwd(ws["tg"], ws["st_t"], self.ln_t[0], 0, g_gemm=ws["dtn"], dx_bf16=ws["dtg"])
            ops.dgelu_mul(ws["dtg"], ws["tz"], M, H, ws["dtg"])
            self._dw(ws["dtg"], ws["yb"][L], M, H, H, "tr.w", "tr.b")
            ops.gemm(ws["dtg"], self.w_tr_t, ws["dh"], EPI_BF16)
            g_resid, g_gemm = None, ws["dh"]
        else:
            ops.cast_f32_bf16(dout, ws["dout_bf"])
            ops.transpose_bf16(ws["dout_bf"], B, self.out_dim, ws["dout_t"])
            ops.transpose_bf16(ws["mp"], B, H, ws["mp_t"])
            ops.gemm(ws["dout_t"], ws["mp_t"], gw, EPI_RESID_F32, resid=gw)
            ops.colsum(dout, B, self.out_dim, gb)
            ops.gemm(ws["dout_bf"], self.w_head_t, ws["dmp"], EPI_F32)
            ops.meanpool_tokens_bwd(ws["dmp"], B, S, ws["dyl"])
            g_resid, g_gemm = ws["dyl"], None
        dt_in, a_in = None, None
        for l in range(L - 1, -1, -1):
            lay = self.layers[l]
            ops.ln_param_grad(ws["s2"][l], ws["stb"][l], 1, self.tp(f"{l}.lnb.w", grad=True), self.tp(f"{l}.lnb.b", grad=True),
                              g_resid=g_resid, g_gemm=g_gemm, dt=dt_in, lora_a=a_in)
            ops.layernorm_bwd(ws["s2"][l], ws["stb"][l], lay.ln_b[0], 1, g_resid=g_resid, g_gemm=g_gemm, dt=dt_in, lora_a=a_in,
                              dx_f32=ws["ds"], dx_bf16=ws["dsb"], dropout=self._drop(ws, self.p_hidden, l, 3))
            self._dw(ws["dsb"], ws["acts"][l], M, H, FF, f"{l}.fc2.w", f"{l}.fc2.b")   # dsb carries fc2's forward dropout mask
            ops.gemm(ws["dsb"], lay.w_fc2_t, ws["dz"], EPI_DGELU_BF16, aux=ws["z"][l])
            self._dw(ws["dz"], ws["ymbs"][l], M, FF, H, f"{l}.fc1.w", f"{l}.fc1.b")
            ops.gemm(ws["dz"], lay.w_fc1_t, ws["dh"], EPI_BF16)
            ops.ln_param_grad(ws["s1"][l], ws["sta"][l], 1, self.tp(f"{l}.lna.w", grad=True), self.tp(f"{l}.lna.b", grad=True),
                              g_resid=ws["ds"], g_gemm=ws["dh"])
            ops.layernorm_bwd(ws["s1"][l], ws["sta"][l], lay.ln_a[0], 1, g_resid=ws["ds"], g_gemm=ws["dh"], dx_f32=ws["ds1"],
                              dx_bf16=ws["dsb"], dropout=self._drop(ws, self.p_hidden, l, 2))
            self._dw(ws["dsb"], ws["ctx"][l], M, H, H, f"{l}.o.w", f"{l}.o.b")
            ops.gemm(ws["dsb"], lay.w_o_t, ws["dctx"], EPI_BF16)
            drop_p = self._drop(ws, self.p_attn, l, 1)   # the forward (BertEngine.forward) left its keep decisions in ws["kbits"]
            lb = self.lora_b(l)
            kb = ws["kbits"][l] if drop_p is not None and ws["kbits"] is not None else None
            part = lb is not None and engine.ATTN_LORA and (drop_p is None or kb is not None)
            ops.attn_bwd(ws["qkv"][l], ws["dctx"], ws["lse"][l], B, S, self.heads, scale, ws["dqkv"], key_bias=ws["key_bias"],
                         dropout=drop_p, keep_bits=kb, lora=(ws["yb"][l][:, H:], lb, ws["dtp"], ws["dbp"]) if part else None)
            self._dw(ws["dqkv"], ws["yb"][l], M, 3 * H, H, f"{l}.q.w", f"{l}.q.b", w_n=3 * H * H)
            if lb is not None:
                gbb = self.lora_b(l, grad=True)
                if part:
                    ops.lora_grad_heads(ws["yb"][l], M, H, B, ws["dtp"], ws["dbp"], ws["dt"], self.lora_a(l, grad=True), gbb[0], gbb[1])
                else:
                    ops.lora_grad(ws["dqkv"], ws["yb"][l], M, H, lb, ws["dt"], self.lora_a(l, grad=True), gbb[0], gbb[1])
            ops.gemm(ws["dqkv"], lay.wqkv_t, ws["dh"], EPI_BF16)
            g_resid, g_gemm = ws["ds1"], ws["dh"]
            dt_in, a_in = (ws["dt"], self.lora_a(l)) if lb is not None else (None, None)
        # ---- embeddings: LayerNorm (its output was dropped in forward with site (-1, 0)) and the three tables ----
        e_drop = self._drop(ws, self.p_hidden, -1, 0)
        ops.ln_param_grad(ws["emb"], ws["st_e"], 1, self.tp("lne.w", grad=True), self.tp("lne.b", grad=True), g_resid=g_resid,
                          g_gemm=g_gemm, dt=dt_in, lora_a=a_in, in_dropout=e_drop)
        ops.layernorm_bwd(ws["emb"], ws["st_e"], self.ln_e[0], 1, g_resid=g_resid, g_gemm=g_gemm, dt=dt_in, lora_a=a_in,
                          dx_f32=ws["demb"], in_dropout=e_drop)
        ids = ws["ids"].contiguous()
        tt = ws["type_ids"]
        ops.embed_grad(ids, None if tt is None else tt.contiguous(), ws["demb"], self.tp("word", grad=True),
                       self.tp("posw", grad=True), self.tp("typew", grad=True), pad_id=self.pad_id)
        ops.set_dropout_step(None)
