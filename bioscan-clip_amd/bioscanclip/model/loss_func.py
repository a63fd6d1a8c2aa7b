"""Contrastive loss -- drop-in for reference ``bioscanclip/model/loss_func.py`` (``construct_label_metrix``,
``ContrastiveLoss``) plus the global-batch form north_star asks for (semantics of ``gather_features`` /
``ClipLoss`` with gather_with_grad=False, local_loss=False: loss_func.py:58-91,117-165).

The arithmetic is the fused HIP InfoNCE (``bsclip_infonce_fwd_bwd``): soft-target CE over every ordered modality
pair, temperature fixed at construction, second ``F.normalize`` applied inside (loss_func.py:43-44).
"""
import torch.nn as nn

from bioscanclip.hip.functional import infonce


def construct_label_metrix(labels):
    """Reference loss_func.py:18-21.  Kept for API compatibility (index compare only, no float math); the HIP
    loss builds the same 0/1 targets on the fly from the label vector."""
    matrix = (labels.unsqueeze(0) == labels.unsqueeze(1)).float()
    return matrix


class ContrastiveLoss(nn.Module):
    def __init__(self, criterion=None, logit_scale=1 / 0.07):
        super(ContrastiveLoss, self).__init__()
        # ``criterion`` is nn.CrossEntropyLoss() in the reference (train_cl.py:190); its soft-target form is what the
        # kernel implements.  Anything else cannot be honoured by the fused kernel.
        if criterion is not None and not isinstance(criterion, nn.CrossEntropyLoss):
            raise TypeError("ContrastiveLoss: the HIP path implements nn.CrossEntropyLoss() with soft targets only")
        self.criterion = criterion
        self.logit_scale = logit_scale

    def forward(self, image_features, dna_features, text_features, label, logit_scale=1 / 0.07):
        # the per-call ``logit_scale`` argument is ignored, as in the reference (loss_func.py:46-47 use self.logit_scale)
        feature_list = [image_features, dna_features, text_features]
        feature_list = [item for item in feature_list if item is not None]
        if len(feature_list) < 2:
            raise ValueError("Too less element for calculating the contrastive loss.")
        return infonce(feature_list, label, self.logit_scale)


class GlobalBatchContrastiveLoss(ContrastiveLoss):
    """Global-batch loss over an all-gathered batch: every rank gathers all ranks' embeddings and
    labels (RCCL all_gather over xGMI, issued per modality from its tower's stream in SimpleCLIP.forward and only
    waited for here), evaluates the full N x N loss
    redundantly and keeps dLoss/dz for its own rows only; the caller all-reduces (SUM) the flat trainable
    gradients.  Equivalent to the single-process loss on the concatenated batch."""

    def __init__(self, criterion=None, logit_scale=1 / 0.07, group=None, overlap=True):
        super().__init__(criterion, logit_scale)
        self.group = group
        if overlap:  # all-gathers from the tower streams, gradient all-reduces from the encoder nodes (hip/dist.py)
            from bioscanclip.hip.dist import enable_overlap
            enable_overlap(group)

    def prefetch_labels(self, label):
        """Start the label all-gather at the top of the step (train_epoch calls this before the encoders run)."""
        from bioscanclip.hip.dist import start_label_gather
        start_label_gather(label)

    def forward(self, image_features, dna_features, text_features, label, logit_scale=1 / 0.07):
        from bioscanclip.hip.dist import gather_features_and_labels
        feats = [f for f in (image_features, dna_features, text_features) if f is not None]
        if len(feats) < 2:
            raise ValueError("Too less element for calculating the contrastive loss.")
        gathered, labels, row0 = gather_features_and_labels(feats, label, self.group)
        return infonce(gathered, labels, self.logit_scale, row0=row0, n_local=feats[0].shape[0])
