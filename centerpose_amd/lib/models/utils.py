"""Tensor helpers with the reference's names (models/utils.py:9-47).  These are host-side plumbing
kept for API compatibility; the device decode (lib.models.decode) does its gathers in one kernel."""
import torch


def _sigmoid(x):
    return torch.clamp(x.sigmoid_(), min=1e-4, max=1 - 1e-4)


def _gather_feat(feat, ind, mask=None):
    if ind.dim() > 2:
        num_symmetry, dim = ind.size(1), feat.size(2)
        ind = ind.unsqueeze(3).expand(ind.size(0), ind.size(1), ind.size(2), dim)
        ind = ind.reshape(ind.size(0), -1, ind.size(3))
        feat = feat.gather(1, ind).view(ind.size(0), num_symmetry, -1, ind.size(2))
        if mask is not None:
            feat = feat[mask.unsqueeze(3).expand_as(feat)].view(-1, dim)
        return feat
    dim = feat.size(2)
    feat = feat.gather(1, ind.unsqueeze(2).expand(ind.size(0), ind.size(1), dim))
    if mask is not None:
        feat = feat[mask.unsqueeze(2).expand_as(feat)].view(-1, dim)
    return feat


def _transpose_and_gather_feat(feat, ind):
    feat = feat.permute(0, 2, 3, 1).contiguous()
    feat = feat.view(feat.size(0), -1, feat.size(3))
    return _gather_feat(feat, ind)


def flip_tensor(x):
    return torch.flip(x, [3])
