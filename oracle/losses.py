"""Oracle: validity-masked per-sequence losses used on the EyeNet / RefineNet train step.

TEST INFRASTRUCTURE.  Restates /root/reference/src/losses/:
  base_loss_with_validity.py:32-73  per clip b: sum_t(valid*loss) / (#valid if #valid > 1 else 1);
                                    then mean over the B clips
  angular.py:29-38                  pitch/yaw -> unit vector (models/common.py:32-36), cosine
                                    similarity (eps 1e-8), hardtanh(+-(1-1e-8)), acos, degrees
  l1.py / mse.py / euclidean.py     mean|a-b| / mean (a-b)^2 / sqrt(sum (a-b)^2) over non-time dims
  cross_entropy.py:27-35            F.binary_cross_entropy per time step (mean over the map)
The reference loops over b (and t for BCE) in Python; the arithmetic is the same batched.
"""
import math

import torch
from torch.nn import functional as F


def pitchyaw_to_vector(a):  # models/common.py:32-36, last dim = (pitch, yaw)
    s, c = torch.sin(a), torch.cos(a)
    return torch.stack([c[..., 0] * s[..., 1], s[..., 0], c[..., 0] * c[..., 1]], dim=-1)


def masked_sequence_mean(per_step, validity):
    """per_step, validity: B x T.  base_loss_with_validity.py:56-73."""
    v = validity.to(per_step.dtype)
    n = v.sum(dim=1)
    acc = (v * per_step).sum(dim=1)
    acc = torch.where(n > 1, acc / n.clamp(min=1), acc)
    return acc.sum() / float(per_step.shape[0])


def angular_steps(a, b):
    """a, b: B x T x 2 pitch/yaw (rad) -> B x T degrees.  angular.py:33-38."""
    va, vb = pitchyaw_to_vector(a), pitchyaw_to_vector(b)
    sim = F.cosine_similarity(va, vb, dim=-1, eps=1e-8)
    sim = F.hardtanh(sim, min_val=-1 + 1e-8, max_val=1 - 1e-8)
    return torch.acos(sim) * (180. / math.pi)


def _reduce_rest(x):
    return x if x.dim() == 2 else x.flatten(2).mean(dim=2)


def l1_steps(a, b):
    return _reduce_rest(torch.abs(a - b))


def mse_steps(a, b):
    return _reduce_rest((a - b) ** 2)


def euclidean_steps(a, b):
    return torch.sqrt(((a - b) ** 2).flatten(2).sum(dim=2))


def bce_steps(a, b):
    """a, b: B x T x 1 x H x W.  cross_entropy.py:31-35 (log terms clamped at -100 by torch)."""
    return F.binary_cross_entropy(a, b, reduction='none').flatten(2).mean(dim=2)


def angular_loss(pred, gt, validity):
    return masked_sequence_mean(angular_steps(pred, gt), validity)


def l1_loss(pred, gt, validity):
    return masked_sequence_mean(l1_steps(pred, gt), validity)


def mse_loss(pred, gt, validity):
    return masked_sequence_mean(mse_steps(pred, gt), validity)


def bce_loss(pred, gt, validity):
    return masked_sequence_mean(bce_steps(pred, gt), validity)
