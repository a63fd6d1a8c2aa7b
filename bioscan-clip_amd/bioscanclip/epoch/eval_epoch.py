"""Label plumbing shared by the inference / evaluation path (reference bioscanclip/epoch/eval_epoch.py:26-38)."""

LEVELS = ("order", "family", "genus", "species")


def convert_label_dict_to_list_of_dict(label_batch):
    """``{'order': [...], 'family': [...], ...}`` (a collated batch) -> one ``{level: name}`` dict per sample."""
    columns = [label_batch[level] for level in LEVELS]
    return [dict(zip(LEVELS, names)) for names in zip(*columns)]
