"""Loss table (reference: lib/models/losses.py:11-61).  `MILNCELoss` runs on the HIP path and, unlike the
reference's hard-coded `.cuda()` (losses.py:18), works on whatever device the embeddings live on."""
import torch.nn as nn

from .functional import milnce_loss


class MILNCELoss(nn.Module):
    def forward(self, video_embd, text_embd):
        return milnce_loss(video_embd, text_embd)


_LOSSES = {"cross_entropy": nn.CrossEntropyLoss, "bce": nn.BCELoss, "bce_logit": nn.BCEWithLogitsLoss, "milnce": MILNCELoss}


def get_loss_func(loss_name):
    if loss_name not in _LOSSES.keys():
        raise NotImplementedError("Loss {} is not supported".format(loss_name))
    return _LOSSES[loss_name]
