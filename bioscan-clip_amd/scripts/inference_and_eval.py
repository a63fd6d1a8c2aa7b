"""Retrieval evaluation (reference scripts/inference_and_eval.py): feature extraction on the HIP encoders, exact
inner-product top-k search on the GPU (``bsclip_topk_ip`` instead of faiss ``IndexFlatIP``), micro / macro top-k accuracy.

Mirrored entry points (same names, argument meaning and return shapes as the reference):
    make_prediction            :414-445
    top_k_micro_accuracy       :448-464
    top_k_macro_accuracy       :467-511
    inference_and_print_result :633-715   (accuracy table only; the csv / plotting side is control plane, out of scope)
    get_features_and_label     :734-784
    main                       :786-871   (config -> load_clip_model -> checkpoint unless load_ckpt=false -> features -> table;
                                           splits from the synthetic evaluation loaders, feature cache as .npz)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from bioscanclip.epoch.inference_epoch import get_feature_and_label  # noqa: E402
from bioscanclip.hip import ops  # noqa: E402

All_TYPE_OF_FEATURES_OF_QUERY = [
    "encoded_image_feature",
    "encoded_dna_feature",
    "encoded_language_feature",
    "averaged_feature",
    "concatenated_feature",
]
All_TYPE_OF_FEATURES_OF_KEY = All_TYPE_OF_FEATURES_OF_QUERY + ["all_key_features"]
LEVELS = ["order", "family", "genus", "species"]


def search_topk(query_feature, keys_feature, max_k, device=None):
    """L2-normalise both sides and return (similarities f32 [Q, max_k], indices int64 [Q, max_k]) as numpy arrays.

    The arithmetic of ``normalize`` + ``IndexFlatIP.add`` + ``.search`` (:415-422) on the GPU.  There is no CPU path: without
    the HIP library or a GPU this raises.
    """
    device = torch.device(device if device is not None else "cuda")
    q = torch.as_tensor(np.ascontiguousarray(query_feature, dtype=np.float32)).to(device)
    k = torch.as_tensor(np.ascontiguousarray(keys_feature, dtype=np.float32)).to(device)
    sims, idx = ops.topk_ip(q, k, int(max_k))
    return sims.cpu().numpy(), idx.cpu().numpy()


def make_prediction(query_feature, keys_feature, keys_label, with_similarity=False, with_indices=False, max_k=5):
    similarities, indices = search_topk(query_feature, keys_feature, max_k)
    pred_list = [{level: [keys_label[i][level] for i in key_indices] for level in LEVELS} for key_indices in indices]
    out = [pred_list]
    if with_similarity:
        out.append(similarities)
    if with_indices:
        out.append(indices)
    return out[0] if len(out) == 1 else out


def _hits(pred_list, gt_list, k, level):
    return [gt[level] in pred[level][:k] for pred, gt in zip(pred_list, gt_list)]


def top_k_micro_accuracy(pred_list, gt_list, k_list=None):
    # like the reference, k_list has no default here: None is not iterable (:451)
    total = len(pred_list)
    return {k: {level: sum(_hits(pred_list, gt_list, k, level)) * 1.0 / total for level in LEVELS} for k in k_list}


def top_k_macro_accuracy(pred_list, gt_list, k_list=None):
    if k_list is None:
        k_list = [1, 3, 5]
    macro, per_class = {}, {}
    for k in k_list:
        macro[k], per_class[k] = {}, {}
        for level in LEVELS:
            seen, right = {}, {}
            for hit, gt in zip(_hits(pred_list, gt_list, k, level), gt_list):
                name = gt[level]
                seen[name] = seen.get(name, 0) + 1
                right[name] = right.get(name, 0) + int(hit)
            per_class[k][level] = {name: right[name] * 1.0 / seen[name] for name in seen}
            total = 0
            for name in seen:  # same summation order as the reference (:499-508)
                total = total + right[name] * 1.0 / seen[name]
            macro[k][level] = total / len(seen)
    return macro, per_class


def get_features_and_label(dataloader, model, device, for_key_set=False, for_open_clip=False):
    model.eval()
    _, lang, _ = get_feature_and_label(dataloader, model, device, type_of_feature="text", for_open_clip=for_open_clip)
    _, dna, _ = get_feature_and_label(dataloader, model, device, type_of_feature="dna", for_open_clip=for_open_clip)
    names, image, labels = get_feature_and_label(dataloader, model, device, type_of_feature="image",
                                                 for_open_clip=for_open_clip)
    split = {
        "file_name_list": names,
        "encoded_dna_feature": dna,
        "encoded_image_feature": image,
        "encoded_language_feature": lang,
        "averaged_feature": None,
        "concatenated_feature": None,
        "label_list": labels,
        "all_key_features": None,
        "all_key_features_label": None,
    }
    if dna is not None and image is not None:
        split["averaged_feature"] = np.mean([image, dna], axis=0)
        split["concatenated_feature"] = np.concatenate((image, dna), axis=1)
    if for_key_set and image is not None and dna is not None and lang is not None:
        split["all_key_features"] = np.concatenate((image, dna, lang), axis=0)
        split["all_key_features_label"] = labels + labels + labels
    return split


def print_micro_and_macro_acc(acc_dict, k_list, args=None):
    for q in All_TYPE_OF_FEATURES_OF_QUERY:
        for kf in All_TYPE_OF_FEATURES_OF_KEY:
            cell = acc_dict.get(q, {}).get(kf)
            if not cell:
                continue
            for kind in ("micro_acc", "macro_acc"):
                for k in k_list:
                    nums = [round(cell[s][kind][k][level], 4) for s in ("seen", "unseen") for level in LEVELS]
                    print(f"Query_feature: {q}||Key_feature: {kf}||{kind} top-{k}\t" + "\t".join(map(str, nums)))


def inference_and_print_result(keys_dict, seen_dict, unseen_dict, args=None, small_species_list=None, k_list=None):
    if k_list is None:
        k_list = [1, 3, 5]
    max_k = k_list[-1]
    acc_dict, per_class_acc, pred_dict = {}, {}, {}
    keys_label = keys_dict["label_list"]
    for q in All_TYPE_OF_FEATURES_OF_QUERY:
        if q not in seen_dict:
            continue
        acc_dict[q], per_class_acc[q], pred_dict[q] = {}, {}, {}
        for kf in All_TYPE_OF_FEATURES_OF_KEY:
            if kf not in keys_dict:
                continue
            acc_dict[q][kf], per_class_acc[q][kf], pred_dict[q][kf] = {}, {}, {}
            keys, seen, unseen = keys_dict[kf], seen_dict[q], unseen_dict[q]
            if keys is None:
                continue
            if kf == "all_key_features":
                keys_label = keys_dict["all_key_features_label"]  # sticks for later key types, as in the reference (:666)
            if seen is None or unseen is None or keys.shape[-1] != seen.shape[-1] or keys.shape[-1] != unseen.shape[-1]:
                continue
            preds = {"seen": make_prediction(seen, keys, keys_label, max_k=max_k),
                     "unseen": make_prediction(unseen, keys, keys_label, max_k=max_k)}
            gts = {"seen": seen_dict["label_list"], "unseen": unseen_dict["label_list"]}
            pred_dict[q][kf] = {"curr_seen_pred_list": preds["seen"], "curr_unseen_pred_list": preds["unseen"]}
            for s in ("seen", "unseen"):
                macro, per_class = top_k_macro_accuracy(preds[s], gts[s], k_list=k_list)
                acc_dict[q][kf][s] = {"micro_acc": top_k_micro_accuracy(preds[s], gts[s], k_list=k_list),
                                      "macro_acc": macro}
                per_class_acc[q][kf][s] = per_class
    print_micro_and_macro_acc(acc_dict, k_list, args)
    return acc_dict, per_class_acc, pred_dict


def main(argv=None):
    """Reference entry (scripts/inference_and_eval.py:786-871): config -> ``load_clip_model`` -> checkpoint
    (``model_config.ckpt_path`` unless ``model_config.load_ckpt`` is false, :839-843) -> features of the key / seen / unseen
    splits -> accuracy table.  The HDF5 splits are not available here: the splits come from the synthetic
    evaluation loaders; extracted features are cached as ``.npz`` (h5py is absent) under the reference's directory layout
    (``extracted_embedding/<dataset>/<model_output_name>/``) and reused with ``load_inference=true`` (:797-833)."""
    from bioscanclip.model.simple_clip import load_clip_model
    from bioscanclip.util.config import load_config
    from bioscanclip.util.synthetic import SyntheticEvalLoader
    from bioscanclip.util.util import load_checked, remove_extra_pre_fix
    here = os.path.dirname(os.path.abspath(__file__))
    args = load_config(os.path.join(here, "..", "bioscanclip", "config"), list(sys.argv[1:] if argv is None else argv))
    mc = args.model_config
    if not torch.cuda.is_available():
        raise RuntimeError("inference_and_eval needs a ROCm GPU: the encoders and the top-k search run in libbsclip_hip.so")
    device = torch.device("cuda", 0)
    ies = getattr(args, "inference_and_eval_setting", None)
    k_list = list(getattr(ies, "k_list", [1, 3, 5])) if ies is not None else [1, 3, 5]
    root = str(getattr(args, "project_root_path", "."))
    folder = os.path.join(root, "extracted_embedding", str(getattr(mc, "dataset", "synthetic")), str(mc.model_output_name))
    feats_path = os.path.join(folder, "extracted_feature_from_val_split.npz")
    splits = None
    if getattr(args, "load_inference", False) and os.path.exists(feats_path):
        z = np.load(feats_path, allow_pickle=True)
        splits = [z[k].item() for k in ("keys", "seen", "unseen")]
    if splits is None:
        print("Initialize model...")
        model = load_clip_model(args, device)
        if hasattr(mc, "load_ckpt") and mc.load_ckpt is False:
            pass
        else:
            load_checked(model, remove_extra_pre_fix(torch.load(str(mc.ckpt_path), map_location="cpu")), f"checkpoint {mc.ckpt_path}")
        model.eval()
        with_text = hasattr(mc, "language")
        bs, n = 24, int(getattr(args, "synthetic_eval_batches", 2))   # the reference evaluates at batch 24 (:846)
        mk = lambda seed: SyntheticEvalLoader(bs, n, with_text=with_text, seed=seed)
        splits = [get_features_and_label(mk(4321), model, device, for_key_set=True),
                  get_features_and_label(mk(4322), model, device), get_features_and_label(mk(4323), model, device)]
        if getattr(args, "save_inference", False):
            os.makedirs(folder, exist_ok=True)
            np.savez(feats_path, keys=np.array(splits[0], dtype=object), seen=np.array(splits[1], dtype=object),
                     unseen=np.array(splits[2], dtype=object))
    keys_dict, seen_dict, unseen_dict = splits
    return inference_and_print_result(keys_dict, seen_dict, unseen_dict, args, small_species_list=None, k_list=k_list)


if __name__ == "__main__":
    main()
