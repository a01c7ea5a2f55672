"""Oracle: the EVE sequence harness -- everything `EVE.forward` does around the two hot modules.

TEST INFRASTRUCTURE (CPU restatement; never imported by the product path).  Restates
  /root/reference/src/models/common.py:32-218   gaze geometry: pitch/yaw <-> vector, rotations, ray / screen-plane
                                                intersection, combined gaze direction, offset (kappa) augmentation
  common.py:226-243                             Gaussian heat-maps on the 128 x 72 grid (+1e-8)
  common.py:249-287                             time-decayed gaze history maps
  common.py:294-323                             soft-argmax (softmax(100 h), expectation on a [0,1]^2 grid, px, clamp)
  /root/reference/src/models/eve.py:441-543     label synthesis (PoG cm, kappa draw, mean origin / PoG, heat-map
                                                labels x validity, combined gaze label)
  eve.py:69-182, 545-601                        per-frame data flow: EyeNet L/R -> [augment] -> PoG -> heat-map ->
                                                RefineNet -> soft-argmax -> PoG cm / gaze
  eve.py:286-439, 234-265                       the loss / metric set and the weighted full_loss
Every frame (b, t) is independent outside the two recurrent modules, so the geometry is written once for a flat
batch of N = B*T frames; the reference's per-t loop gives the same numbers.
Pinned by tests/golden/eve_harness.npz (the reference's own EVE run on oracle/detweights.eve_batch).
"""
import math

import numpy as np
import torch
from torch.nn import functional as F

from . import losses, sequence


# ---------------------------------------------------------------------------------------------- geometry (flat N)
def pitchyaw_to_vector(a):                       # common.py:32-36
    s, c = torch.sin(a), torch.cos(a)
    return torch.stack([c[:, 0] * s[:, 1], s[:, 0], c[:, 0] * c[:, 1]], dim=1)


def vector_to_pitchyaw(v):                       # common.py:43-55 (3-vector branch)
    v = v.reshape(-1, 3)
    n = v / (torch.norm(v, dim=1, keepdim=True) + 1e-7)
    return torch.stack([torch.asin(n[:, 1]), torch.atan2(n[:, 0], n[:, 2])], dim=1)


def pitchyaw_to_rotation(a):                     # common.py:58-80: R = Ry(yaw) Rx(pitch) in the reference's sign convention
    c, s = torch.cos(a), torch.sin(a)
    one, zero = torch.ones_like(c[:, 0]), torch.zeros_like(c[:, 0])
    m1 = torch.stack([one, zero, zero, zero, c[:, 0], s[:, 0], zero, -s[:, 0], c[:, 0]], dim=1).view(-1, 3, 3)
    m2 = torch.stack([c[:, 1], zero, s[:, 1], zero, one, zero, -s[:, 1], zero, c[:, 1]], dim=1).view(-1, 3, 3)
    return m2 @ m1


def transform_point(T, p):                       # common.py:93-98
    return (T[:, :3, :3] @ p.unsqueeze(-1)).squeeze(-1) + T[:, :3, 3]


def rotate(T, v):                                # common.py:101-106
    return (T[:, :3, :3] @ v.unsqueeze(-1)).squeeze(-1)


def intersect_with_screen_plane(o, d):           # common.py:113-133: plane z = 0
    t = (-o[:, 2]) / (d[:, 2] + 1e-7)
    return (o + t.unsqueeze(1) * d)[:, :2]


def combined_gaze_direction(origin, pog_mm, head_R, camera_transformation):      # common.py:136-154
    p = transform_point(camera_transformation, F.pad(pog_mm, (0, 1)))
    d = (head_R @ (p - origin).unsqueeze(-1)).squeeze(-1)
    return vector_to_pitchyaw(-d)


def to_screen_coordinates(origin, g, R, inv_camera_transformation, pixels_per_millimeter, screen_px):   # common.py:157-187
    d = -pitchyaw_to_vector(g)
    d = (R.transpose(1, 2) @ d.unsqueeze(-1)).squeeze(-1)
    d = rotate(inv_camera_transformation, d)
    o = transform_point(inv_camera_transformation, origin)
    mm = intersect_with_screen_plane(o, d)
    px = torch.stack([torch.clamp(mm[:, 0] * pixels_per_millimeter[:, 0], 0.0, float(screen_px[0])),
                      torch.clamp(mm[:, 1] * pixels_per_millimeter[:, 1], 0.0, float(screen_px[1]))], dim=-1)
    return mm, px


def offset_augmentation(g, head_R, kappa):       # common.py:190-229 (inverse_kappa=False, the only use)
    v = -pitchyaw_to_vector(g)
    v = -(head_R.transpose(1, 2) @ v.unsqueeze(-1)).squeeze(-1)
    k = pitchyaw_to_vector(kappa)
    v = (pitchyaw_to_rotation(vector_to_pitchyaw(v)) @ k.unsqueeze(-1)).squeeze(-1)
    v = -(head_R @ (-v).unsqueeze(-1)).squeeze(-1)
    return vector_to_pitchyaw(v)


# ---------------------------------------------------------------------------------------------- maps
def make_heatmaps(centres_px, sigma, config):    # common.py:236-250, centres N x 2 (px) -> N x 1 x H x W
    w, h = config.gaze_heatmap_size
    xs = torch.arange(w, dtype=centres_px.dtype).view(1, 1, w)          # (float32 in the parity tests; float64 when the
    ys = torch.arange(h, dtype=centres_px.dtype).view(1, h, 1)          #  oracle is evaluated as its own error bar)
    cx = (w / config.actual_screen_size[0]) * centres_px[:, 0].view(-1, 1, 1)
    cy = (h / config.actual_screen_size[1]) * centres_px[:, 1].view(-1, 1, 1)
    alpha = -0.5 / (sigma ** 2)
    return (1e-8 + torch.exp(alpha * ((xs - cx) ** 2 + (ys - cy) ** 2))).unsqueeze(1)


def gaze_history_maps(timestamps, heatmaps, validity, config):
    """common.py:256-297.  timestamps B x T (int64 ns, 0 = padding), heatmaps B x L x 1 x H x W (L <= T frames so far),
    validity B x T.  Map of clip b = sum over the L frames with a non-zero timestamp of
    validity * decay^(ms to the clip's LAST non-zero timestamp of the whole window) * heat-map."""
    B, L = heatmaps.shape[:2]
    out = []
    for b in range(B):
        ts = timestamps[b, :L]
        target = ts[torch.nonzero(ts)][-1]
        acc = torch.zeros_like(heatmaps[b, 0])
        for t in range(L):
            if ts[t] == 0:
                continue
            w = torch.pow(torch.tensor(config.gaze_history_map_decay_per_ms, dtype=heatmaps.dtype),
                          (target - ts[t]).to(heatmaps.dtype) * 1e-6).view(1, 1)
            acc = acc + validity[b, t].to(heatmaps.dtype) * w * heatmaps[b, t]
        out.append(acc)
    return torch.stack(out, dim=0)


def soft_argmax(heatmaps, config):               # common.py:304-333, N x 1 x H x W -> N x 2 (px)
    n, _, h, w = heatmaps.shape
    ref_xs, ref_ys = np.meshgrid(np.linspace(0, 1.0, num=w, endpoint=True), np.linspace(0, 1.0, num=h, endpoint=True),
                                 indexing='xy')
    ref_xs = torch.tensor(ref_xs.reshape(1, h * w).astype(np.float32)).to(heatmaps.dtype)
    ref_ys = torch.tensor(ref_ys.reshape(1, h * w).astype(np.float32)).to(heatmaps.dtype)
    p = F.softmax(1e2 * heatmaps.reshape(n, h * w), dim=-1)
    sw, sh = config.actual_screen_size
    return torch.stack([torch.clamp(sw * torch.sum(ref_xs * p, dim=-1), 0.0, sw),
                        torch.clamp(sh * torch.sum(ref_ys * p, dim=-1), 0.0, sh)], dim=-1)


# ---------------------------------------------------------------------------------------------- labels
def synthesise_labels(batch, config, training, kappa=None):
    """eve.py:441-543 on a dict of B x T x ... tensors; returns the extended dict (the reference mutates its input).
    `kappa` = (left, right) B x 2 arrays replaces the reference's np.random.normal draw (same call order as the
    reference when None: left then right, each (B, 2), scale radians(sigma))."""
    d = dict(batch)
    B, T = d['left_eye_patch'].shape[:2]
    for side in ('left', 'right'):
        d[side + '_PoG_cm_tobii'] = d[side + '_PoG_tobii'] * (0.1 * d['millimeters_per_pixel'])
        d[side + '_PoG_cm_tobii_validity'] = d[side + '_PoG_tobii_validity']
    if training and config.refine_net_do_offset_augmentation:
        std = np.radians(config.refine_net_offset_augmentation_sigma)
        if kappa is None:
            kappa = (np.random.normal(size=(B, 2), loc=0.0, scale=std), np.random.normal(size=(B, 2), loc=0.0, scale=std))
        for side, k in zip(('left', 'right'), kappa):
            d[side + '_kappa_fake'] = torch.tensor(np.repeat(np.asarray(k)[:, None], T, axis=1).astype(np.float32)).to(d['left_o'].dtype)
    d['o'] = 0.5 * (d['left_o'] + d['right_o'])
    d['o_validity'] = d['left_o_validity']
    d['PoG_px_tobii'] = torch.stack([d['left_PoG_tobii'], d['right_PoG_tobii']], dim=-1).mean(dim=-1)
    d['PoG_cm_tobii'] = torch.stack([d['left_PoG_cm_tobii'], d['right_PoG_cm_tobii']], dim=-1).mean(dim=-1)
    valid = d['left_PoG_tobii_validity'].bool() & d['right_PoG_tobii_validity'].bool()
    d['PoG_px_tobii_validity'] = d['PoG_cm_tobii_validity'] = valid
    if config.refine_net_enabled:
        flat = d['PoG_px_tobii'].reshape(B * T, 2)
        for name, sigma in (('initial', config.gaze_heatmap_sigma_initial), ('history', config.gaze_heatmap_sigma_history),
                            ('final', config.gaze_heatmap_sigma_final)):
            m = make_heatmaps(flat, sigma, config).view(B, T, 1, *reversed(config.gaze_heatmap_size))
            d['heatmap_' + name] = m * valid.to(m.dtype).view(B, T, 1, 1, 1)
            d['heatmap_%s_validity' % name] = valid
    d['g'] = combined_gaze_direction(d['o'].reshape(-1, 3), 10.0 * d['PoG_cm_tobii'].reshape(-1, 2),
                                     d['left_R'].reshape(-1, 3, 3), d['camera_transformation'].reshape(-1, 4, 4)).view(B, T, 2)
    d['g_validity'] = valid
    return d


# ---------------------------------------------------------------------------------------------- forward
def _pog_block(inter, d, suffix_in, suffix_out, config):
    """eve.py:545-601 for all frames at once: per-eye PoG, their mean, the combined gaze and the heat-map."""
    B, T = d['left_o'].shape[:2]
    flat = lambda k, *s: d[k].reshape(B * T, *s)
    for side in ('left', 'right'):
        mm, px = to_screen_coordinates(flat(side + '_o', 3), inter[side + '_g_' + suffix_in].reshape(B * T, 2),
                                       flat(side + '_R', 3, 3), flat('inv_camera_transformation', 4, 4),
                                       flat('pixels_per_millimeter', 2), config.actual_screen_size)
        inter['%s_PoG_cm_%s' % (side, suffix_out)] = (0.1 * mm).view(B, T, 2)
        inter['%s_PoG_px_%s' % (side, suffix_out)] = px.view(B, T, 2)
    for unit in ('px', 'cm'):
        inter['PoG_%s_%s' % (unit, suffix_out)] = torch.stack(
            [inter['left_PoG_%s_%s' % (unit, suffix_out)], inter['right_PoG_%s_%s' % (unit, suffix_out)]], dim=-1).mean(dim=-1)
    inter['PoG_mm_' + suffix_out] = 10.0 * inter['PoG_cm_' + suffix_out]
    inter['g_' + suffix_out] = combined_gaze_direction(
        flat('o', 3), inter['PoG_mm_' + suffix_out].reshape(B * T, 2), flat('left_R', 3, 3),
        flat('camera_transformation', 4, 4)).view(B, T, 2)
    if config.refine_net_enabled:
        inter['heatmap_' + suffix_out] = make_heatmaps(
            inter['PoG_px_' + suffix_out].reshape(B * T, 2), config.gaze_heatmap_sigma_initial, config
        ).view(B, T, 1, *reversed(config.gaze_heatmap_size))


def eve_forward(eye_net, refine_net, batch, config, training, kappa=None, create_images=False):
    """The reference's EVE.forward (eve.py:69-284) on a dict of B x T x ... tensors.  Returns (output_dict with every
    loss_* / metric_* scalar and full_loss, intermediate dict of B x T x ... tensors, label dict)."""
    d = synthesise_labels(batch, config, training, kappa)
    B, T = d['left_eye_patch'].shape[:2]
    inter = sequence.eyenet_sequence(eye_net, d)                    # eve.py:105-111 over t
    augment = training and config.refine_net_do_offset_augmentation
    if augment:                                                     # eve.py:114-135
        _pog_block(inter, d, 'initial', 'initial_unaugmented', config)
        for side in ('left', 'right'):
            inter[side + '_g_initial_unaugmented'] = inter[side + '_g_initial']
            inter[side + '_g_initial'] = offset_augmentation(
                inter[side + '_g_initial'].reshape(B * T, 2), d['head_R'].reshape(B * T, 3, 3),
                d[side + '_kappa_fake'].reshape(B * T, 2)).view(B, T, 2)
        _pog_block(inter, d, 'initial', 'initial_augmented', config)
    _pog_block(inter, d, 'initial', 'initial', config)               # eve.py:138-143
    if create_images:
        hist = make_heatmaps(inter['PoG_px_initial'].reshape(B * T, 2), config.gaze_heatmap_sigma_history, config)
        inter['history_initial_last'] = gaze_history_maps(d['timestamps'], hist.view(B, T, 1, *hist.shape[2:]),
                                                          d['PoG_px_tobii_validity'], config)
    if refine_net is not None:                                      # eve.py:146-166
        hf, _ = sequence.refinenet_sequence(refine_net, inter['heatmap_initial'], d.get('screen_frame'))
        inter['heatmap_final'] = hf
        inter['PoG_px_final'] = soft_argmax(hf.reshape(B * T, 1, *hf.shape[3:]), config).view(B, T, 2)
        inter['PoG_cm_final'] = inter['PoG_px_final'] * (0.1 * d['millimeters_per_pixel'])
        inter['g_final'] = combined_gaze_direction(
            d['o'].reshape(-1, 3), (10.0 * inter['PoG_cm_final']).reshape(-1, 2), d['left_R'].reshape(-1, 3, 3),
            d['camera_transformation'].reshape(-1, 4, 4)).view(B, T, 2)
        if create_images:
            inter['refined_gaze_history'] = gaze_history_maps(d['timestamps'], hf, d['PoG_px_tobii_validity'], config)
    out = losses_and_metrics(d, inter, config, augment)
    out['full_loss'] = full_loss(out, config)
    return out, inter, d


def losses_and_metrics(d, inter, config, augment):                   # eve.py:286-439
    out = {}
    ang = lambda p, k: losses.angular_loss(p, d[k], d[k + '_validity'])
    mse = lambda p, k, ref=d: losses.masked_sequence_mean(losses.mse_steps(p, ref[k]), ref[k + '_validity'])
    euc = lambda p, k, ref=d: losses.masked_sequence_mean(losses.euclidean_steps(p, ref[k]), ref[k + '_validity'])
    un = '_unaugmented' if augment else ''
    for side in ('left', 'right'):
        out['loss_ang_%s_g_initial' % side] = ang(inter['%s_g_initial%s' % (side, un)], side + '_g_tobii')
        p = inter['%s_PoG_cm_initial%s' % (side, un)]
        out['loss_mse_%s_PoG_cm_initial' % side] = mse(p, side + '_PoG_cm_tobii')
        out['metric_euc_%s_PoG_cm_initial' % side] = euc(p, side + '_PoG_cm_tobii')
        out['metric_euc_%s_PoG_px_initial' % side] = euc(inter[side + '_PoG_px_initial'], side + '_PoG_tobii')
        out['loss_l1_%s_pupil_size' % side] = losses.l1_loss(inter[side + '_pupil_size'], d[side + '_p'], d[side + '_p_validity'])
    lr = {'right_PoG_cm_initial': inter['right_PoG_cm_initial'],
          'right_PoG_cm_initial_validity': d['left_PoG_tobii_validity'] & d['right_PoG_tobii_validity']}
    out['loss_mse_lr_consistency'] = mse(inter['left_PoG_cm_initial'], 'right_PoG_cm_initial', lr)
    out['metric_euc_lr_consistency'] = euc(inter['left_PoG_cm_initial'], 'right_PoG_cm_initial', lr)
    if 'heatmap_initial' in d:
        out['loss_ce_heatmap_initial'] = losses.bce_loss(inter['heatmap_initial' + un], d['heatmap_initial'], d['heatmap_initial_validity'])
    if 'heatmap_final' in inter:
        out['loss_ce_heatmap_final'] = losses.bce_loss(inter['heatmap_final'], d['heatmap_final'], d['heatmap_final_validity'])
        out['loss_mse_heatmap_final'] = mse(inter['heatmap_final'], 'heatmap_final')
    if config.refine_net_do_offset_augmentation and augment:
        out['metric_euc_PoG_px_initial_unaugmented'] = euc(inter['PoG_px_initial_unaugmented'], 'PoG_px_tobii')
        out['metric_euc_PoG_cm_initial_unaugmented'] = euc(inter['PoG_cm_initial_unaugmented'], 'PoG_cm_tobii')
        out['metric_ang_g_initial_unaugmented'] = ang(inter['g_initial_unaugmented'], 'g')
    for stage in ('initial', 'final'):
        if 'PoG_px_' + stage not in inter:
            continue
        out['loss_mse_PoG_px_' + stage] = mse(inter['PoG_px_' + stage], 'PoG_px_tobii')
        out['metric_euc_PoG_px_' + stage] = euc(inter['PoG_px_' + stage], 'PoG_px_tobii')
        out['loss_mse_PoG_cm_' + stage] = mse(inter['PoG_cm_' + stage], 'PoG_cm_tobii')
        out['metric_euc_PoG_cm_' + stage] = euc(inter['PoG_cm_' + stage], 'PoG_cm_tobii')
        out['metric_ang_g_' + stage] = ang(inter['g_' + stage], 'g')
    return out


def full_loss(out, config):                                          # eve.py:234-265
    total = torch.zeros(())
    total = total + config.loss_coeff_g_ang_initial * (out['loss_ang_left_g_initial'] + out['loss_ang_right_g_initial'])
    if config.loss_coeff_PoG_cm_initial > 0.0:
        total = total + config.loss_coeff_PoG_cm_initial * (out['loss_mse_left_PoG_cm_initial'] + out['loss_mse_right_PoG_cm_initial'])
    total = total + config.loss_coeff_pupil_size * (out['loss_l1_left_pupil_size'] + out['loss_l1_right_pupil_size'])
    if 'loss_mse_PoG_cm_final' in out:
        total = total + config.loss_coeff_PoG_cm_final * out['loss_mse_PoG_cm_final']
    if 'loss_ce_heatmap_initial' in out:
        total = total + config.loss_coeff_heatmap_ce_initial * out['loss_ce_heatmap_initial']
    if 'loss_ce_heatmap_final' in out:
        total = total + config.loss_coeff_heatmap_ce_final * out['loss_ce_heatmap_final']
    if 'loss_mse_heatmap_final' in out:
        total = total + config.loss_coeff_heatmap_mse_final * out['loss_mse_heatmap_final']
    return total
