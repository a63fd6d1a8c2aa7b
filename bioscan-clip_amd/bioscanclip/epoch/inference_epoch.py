"""Feature extraction over a dataloader (reference bioscanclip/epoch/inference_epoch.py:8-68).

Same contract as the reference's ``get_feature_and_label``: eval mode, no autograd, every feature row L2-normalised, and a
``(file_name_list, features ndarray [N, D], label_list)`` triple (``(None, None, None)`` when the model has no encoder of
the requested modality).  The encoders run on the HIP engines; features stay in HBM until the loader is exhausted, so the
host sees one copy instead of a python-list append per batch.
"""
import numpy as np
import torch

from bioscanclip.epoch.eval_epoch import convert_label_dict_to_list_of_dict
from bioscanclip.hip import functional as HF

_ENCODER_OF = {"dna": "dna_encoder", "image": "image_encoder", "text": "language_encoder"}


def get_feature_and_label(dataloader, model, device, type_of_feature="dna", for_open_clip=False, multi_gpu=False):
    if type_of_feature not in _ENCODER_OF:
        raise TypeError(f"{type_of_feature} is not a valid input type")
    core = model.module if multi_gpu else model
    encoder = getattr(core, _ENCODER_OF[type_of_feature])
    if encoder is None:
        return None, None, None
    if for_open_clip:
        raise NotImplementedError("open_clip towers are outside the HIP hot path (SURVEY 8: out of scope)")

    features, label_list, file_name_list = [], [], []
    model.eval()
    with torch.no_grad():
        for batch in dataloader:
            processid_batch, image_input, dna_input, input_ids, token_type_ids, attention_mask, label_batch = batch
            if type_of_feature == "dna":
                out = encoder(dna_input.to(device))
            elif type_of_feature == "image":
                out = encoder(image_input.to(device))
            else:
                out = encoder({"input_ids": input_ids.to(device), "token_type_ids": token_type_ids.to(device),
                               "attention_mask": attention_mask.to(device)})
            # the reference normalises only on its single-process branch; under multi_gpu=True it keeps model.module's raw
            # output for dna / image (inference_epoch.py:36-54) -- the text branch always normalises (:58)
            raw = multi_gpu and type_of_feature != "text"
            features.append(out.float() if raw else HF.l2_normalize(out.float()))
            label_list += convert_label_dict_to_list_of_dict(label_batch)
            file_name_list += list(processid_batch)
    if not features:
        return file_name_list, np.zeros((0,)), label_list
    # the reference builds the array from python floats, i.e. float64 holding f32 values
    return file_name_list, torch.cat(features).cpu().numpy().astype(np.float64), label_list
