"""Entries used by ``LoRA_barcode_bert.forward`` (reference dna_encoder.py:103-105) and ``LoRA_bert.forward``
(language_encoder.py:87-89)."""
import torch

from .engine import BertEngine, run_encoder, wants_fp8, wants_full_ft


def _need_gpu(t, who):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(f"{who}: inputs must live on the GPU (bioscanclip has no CPU compute path)")


def barcode_bert_forward(module, ids):
    _need_gpu(ids, "LoRA_barcode_bert.forward")
    m = module.lora_barcode_bert
    def build():
        heads = (m.cls.predictions.transform, m.cls.predictions.decoder)
        if wants_full_ft(module):
            from .engine_ft import BertEngineFT
            return BertEngineFT(m.bert, "mlm_softmax_mean", heads, ids.device)
        return BertEngine(m.bert, "mlm_softmax_mean", heads, ids.device, fp8=wants_fp8(module))
    # the reference passes input_ids only: token_type 0, no attention mask
    return run_encoder(module, build, (ids.to(torch.int64), None, None))


def bert_text_forward(module, x):
    ids = x["input_ids"]
    _need_gpu(ids, "LoRA_bert.forward")
    def build():
        if wants_full_ft(module):
            from .engine_ft import BertEngineFT
            return BertEngineFT(module.lora_bert, "mean_proj", (module.proj,), ids.device)
        return BertEngine(module.lora_bert, "mean_proj", (module.proj,), ids.device)
    tt = x.get("token_type_ids")
    am = x.get("attention_mask")
    return run_encoder(module, build, (ids.to(torch.int64), None if tt is None else tt.to(torch.int64),
                                       None if am is None else am))
