"""Validity-masked sequence losses of the train step, batched over clips.  The terms over [B, T, k] predictions
(gaze angles, PoG, pupil size: a few KB) are small device-side tensor expressions, the four EyeNet terms one fused
kernel (EyeLossesFn); the terms over [B, T, 1, 72, 128] heat-maps (35 MB at B=32 x T=30) run on the HIP kernels of
csrc/heatmap_loss.hip (ops.HeatmapLossFn).

Semantics of /root/reference/src/losses/: per clip, sum over valid time steps divided by the number
of valid steps when that number exceeds one, then the mean over clips
(base_loss_with_validity.py:64-73); angular error in degrees between pitch/yaw gaze vectors
(angular.py:33-38, models/common.py:32-36); L1 / MSE reduced over the non-time dims (l1.py, mse.py);
per-step binary cross-entropy of the heat-map (cross_entropy.py:31-35).
The reference iterates clips in Python with a host sync per clip; here one masked reduction does it.
"""
import math

import torch
import torch.nn.functional as F


def _masked_clip_mean(per_step, validity):
    v = validity.to(per_step.dtype)
    count = v.sum(dim=1)
    total = (per_step * v).sum(dim=1)
    denom = torch.where(count > 1, count, torch.ones_like(count))
    return (total / denom).mean()


def gaze_vectors(pitchyaw):
    pitch, yaw = pitchyaw[..., 0], pitchyaw[..., 1]
    cp = torch.cos(pitch)
    return torch.stack([cp * torch.sin(yaw), torch.sin(pitch), cp * torch.cos(yaw)], dim=-1)


def angular_loss(pred, target, validity):
    cos = F.cosine_similarity(gaze_vectors(pred), gaze_vectors(target), dim=-1, eps=1e-8)
    cos = F.hardtanh(cos, min_val=-1 + 1e-8, max_val=1 - 1e-8)
    return _masked_clip_mean(torch.acos(cos) * (180.0 / math.pi), validity)


def _per_step(x):
    return x if x.dim() == 2 else x.flatten(2).mean(dim=2)


def l1_loss(pred, target, validity):
    return _masked_clip_mean(_per_step((pred - target).abs()), validity)


def _heatmap_kernel_ok(pred, target):
    """[B, T, 1, H, W] float maps on the GPU go through the fused kernels (csrc/heatmap_loss.hip); the [B, T, k]
    point-of-gaze terms below stay small tensor expressions."""
    return pred.dim() == 5 and pred.is_cuda and pred.dtype == torch.float32 and target.dtype == torch.float32 \
        and target.shape == pred.shape and not target.requires_grad


def mse_loss(pred, target, validity):
    if _heatmap_kernel_ok(pred, target):
        from .ops import HeatmapLossFn
        return HeatmapLossFn.apply(pred, target, validity, 1)
    return _masked_clip_mean(_per_step((pred - target) ** 2), validity)


def euclidean_loss(pred, target, validity):                 # euclidean.py:27-33
    return _masked_clip_mean(torch.sqrt(((pred - target) ** 2).flatten(2).sum(dim=2)), validity)


def bce_loss(pred, target, validity):
    if _heatmap_kernel_ok(pred, target):
        from .ops import HeatmapLossFn
        return HeatmapLossFn.apply(pred, target, validity, 0)
    return _masked_clip_mean(_per_step(F.binary_cross_entropy(pred, target, reduction='none')), validity)


class EyeLossesFn(torch.autograd.Function):
    """All four EyeNet loss terms, their weighted sum and the gradients w.r.t. the predictions from one kernel
    (kernels.eye_losses); returns terms[5] = ang_l, l1_l, ang_r, l1_r, full."""

    @staticmethod
    def forward(ctx, g_l, g_r, p_l, p_r, tgt, c_ang, c_l1):
        from .kernels import default_kernels
        tg_l, tg_r, vg_l, vg_r, tp_l, tp_r, vp_l, vp_r = tgt
        terms, dg, dp = default_kernels().eye_losses((g_l, g_r), (tg_l, tg_r), (vg_l, vg_r), (p_l, p_r), (tp_l, tp_r),
                                                     (vp_l, vp_r), c_ang, c_l1)
        ctx.coeffs = (c_ang, c_l1)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(dg[0], dg[1], dp[0], dp[1])
        return tuple(terms.unbind(0))

    @staticmethod
    def backward(ctx, g_ang_l, g_l1_l, g_ang_r, g_l1_r, g_full):
        dg_l, dg_r, dp_l, dp_r = ctx.saved_tensors
        c_ang, c_l1 = ctx.coeffs

        def scaled(unit, g_term, coeff):                      # unit * (g_term + coeff * g_full); usually only g_full is set
            w = None
            if g_full is not None:
                w = g_full * coeff
            if g_term is not None:
                w = g_term if w is None else w + g_term
            return unit * w if w is not None else None

        return (scaled(dg_l, g_ang_l, c_ang), scaled(dg_r, g_ang_r, c_ang), scaled(dp_l, g_l1_l, c_l1),
                scaled(dp_r, g_l1_r, c_l1), None, None, None)


def _fused_losses_ok(out, batch):
    from .kernels import default_kernels
    t = out['left_g_initial']
    return (t.is_cuda and t.dtype == torch.float32 and hasattr(default_kernels(), 'eye_losses') and
            t.dim() == 3 and t.shape[1] <= 256 and batch['left_g_tobii'].dtype == torch.float32)


def eyenet_loss_terms(out, batch, config):
    """The terms of eve.py:286-325 that carry weight in eye_net.json, and their weighted sum (:234-265)."""
    if _fused_losses_ok(out, batch):
        tgt = tuple(batch[k] for k in ('left_g_tobii', 'right_g_tobii', 'left_g_tobii_validity', 'right_g_tobii_validity',
                                       'left_p', 'right_p', 'left_p_validity', 'right_p_validity'))
        t = EyeLossesFn.apply(out['left_g_initial'], out['right_g_initial'], out['left_pupil_size'], out['right_pupil_size'],
                              tgt, float(config.loss_coeff_g_ang_initial), float(config.loss_coeff_pupil_size))
        return {'loss_ang_left_g_initial': t[0], 'loss_l1_left_pupil_size': t[1], 'loss_ang_right_g_initial': t[2],
                'loss_l1_right_pupil_size': t[3], 'full_loss': t[4]}
    terms = {}
    for side in ('left', 'right'):
        terms['loss_ang_%s_g_initial' % side] = angular_loss(
            out[side + '_g_initial'], batch[side + '_g_tobii'], batch[side + '_g_tobii_validity'])
        terms['loss_l1_%s_pupil_size' % side] = l1_loss(
            out[side + '_pupil_size'], batch[side + '_p'], batch[side + '_p_validity'])
    terms['full_loss'] = (
        config.loss_coeff_g_ang_initial * (terms['loss_ang_left_g_initial'] + terms['loss_ang_right_g_initial']) +
        config.loss_coeff_pupil_size * (terms['loss_l1_left_pupil_size'] + terms['loss_l1_right_pupil_size']))
    return terms


def refinenet_loss_terms(heatmap_final, heatmap_gt, validity, config):
    terms = {'loss_ce_heatmap_final': bce_loss(heatmap_final, heatmap_gt, validity),
             'loss_mse_heatmap_final': mse_loss(heatmap_final, heatmap_gt, validity)}
    terms['full_loss'] = (config.loss_coeff_heatmap_ce_final * terms['loss_ce_heatmap_final'] +
                          config.loss_coeff_heatmap_mse_final * terms['loss_mse_heatmap_final'])
    return terms
