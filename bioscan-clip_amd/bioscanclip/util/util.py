"""Checkpoint helpers with the reference's names (bioscanclip/util/util.py:72-84).  The rest of the reference's util.py
(result tables, plotting colours, faiss helper) is control plane and not rebuilt."""
import torch


def remove_extra_pre_fix(state_dict):
    """Strip the ``module.`` prefix DataParallel/DDP checkpoints carry (util.py:72-78)."""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}


def load_bert_model(bert_model, path_to_ckpt):
    """util.py:81-84: load a (possibly ``module.``-prefixed) checkpoint into a BERT parameter tree, strictly."""
    state_dict = torch.load(path_to_ckpt, map_location=torch.device("cpu"))
    bert_model.load_state_dict(remove_extra_pre_fix(state_dict))
