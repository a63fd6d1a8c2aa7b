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


def load_checked(module, state_dict, what, allow_missing=(), allow_unexpected=()):
    """``load_state_dict`` that refuses a checkpoint which does not cover the module: the reference loads strictly
    (util.py:81-84); a silently half-loaded trunk would train on random weights.  ``allow_*``: key prefixes that may be
    absent / extra (e.g. a classifier head that is replaced right after, HF's ``position_ids`` buffer)."""
    res = module.load_state_dict(state_dict, strict=False)
    missing = [k for k in res.missing_keys if not any(k.startswith(a) for a in allow_missing)]
    unexpected = [k for k in res.unexpected_keys if not any(k.startswith(a) for a in allow_unexpected)]
    if missing or unexpected:
        raise RuntimeError(f"{what}: checkpoint does not match the parameter tree -- missing {missing[:8]} "
                           f"({len(missing)} keys), unexpected {unexpected[:8]} ({len(unexpected)} keys)")
    return res
