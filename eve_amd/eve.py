"""EVE: the sequence harness around EyeNet and RefineNet -- drop-in for /root/reference/src/models/eve.py (class EVE).

Same constructor (`EVE(output_predictions=False)`, reads the config singleton), same call
(`model(full_input_dict, create_images=False, current_epoch=None)`; in training mode the argument is a dict holding
ONE data-source dict, eve.py:70-72), same result: a dict with `left/right_pupil_size`, every `loss_*` / `metric_*`
scalar of eve.py:286-439, the weighted `full_loss` (:234-265), the prediction tensors when `output_predictions`
(:194-222) and the visualisation maps when `create_images` (:268-283).  The input dict is extended in place with the
synthesised labels exactly as the reference does (:441-543).

What differs is the schedule.  The reference walks the clip frame by frame and runs ~60 small torch ops + Python-side
per-sample loops per frame; outside the two recurrent networks every frame (b, t) is independent, so here
  * EyeNet and RefineNet run over whole clips (`forward_sequence`: the recurrences are scans inside the modules),
  * gaze geometry (models/common.py:157-229), Gaussian heat-maps (:236-255) and soft-argmax (:304-333) are one HIP
    kernel each over the N = B*T frames (csrc/gaze_geometry.hip, autograd through ops.GazeToPoGFn / MakeHeatmapsFn /
    SoftArgmaxFn),
  * the O(T^2) gaze-history maps (:256-297) are built only for `create_images` (nothing else reads them),
  * the combined L/R gaze angles `g_*` (:136-154) feed metrics only and are returned without a graph.
The masked-loss reductions are device-side tensor ops on [B, T, ...] values (losses.py).
"""
import numpy as np
import torch
import torch.nn as nn

from . import losses, ops
from .config import get_config
from .eye_net import EyeNet
from .kernels import default_kernels


def _mean2(a, b):
    return torch.stack([a, b], dim=-1).mean(dim=-1)


class EVE(nn.Module):
    def __init__(self, output_predictions=False):
        super(EVE, self).__init__()
        config = get_config()
        self.config = config
        self.output_predictions = output_predictions
        if config.eye_net_load_pretrained or (config.refine_net_enabled and config.refine_net_load_pretrained):
            # utils/load_model.py downloads the released weights; there is no network on the hot path's side of the fence
            raise RuntimeError('eve_amd.EVE: *_load_pretrained needs the released checkpoints; load them with '
                               'load_state_dict (same keys as the reference) and set the config flag to False')
        self.eye_net = EyeNet()                                       # eve.py:55-60
        if config.eye_net_frozen:
            for p in self.eye_net.parameters():
                p.requires_grad = False
        self.refine_net = None
        if config.refine_net_enabled:                                 # eve.py:65
            from .refine_net import RefineNet
            self.refine_net = RefineNet()

    # ------------------------------------------------------------------------------------------ kappa draw (eve.py:463-479)
    @staticmethod
    def _draw_kappa(B, T, std):
        """The reference's draw: same amounts, same order, same global numpy RNG (:463-466), repeated over T (:467-479)."""
        draws = {'left': np.random.normal(size=(B, 2), loc=0.0, scale=std),
                 'right': np.random.normal(size=(B, 2), loc=0.0, scale=std)}
        return {side: np.repeat(np.expand_dims(draws[side], axis=1), T, axis=1).astype(np.float32) for side in draws}

    def refresh_static_kappa(self, B, T, device):
        """One host draw per optimiser step into FIXED device buffers (allocated on first use): what lets the whole train
        step be captured into a hipGraph although the reference draws kappa_fake on the host inside forward.  The sequence
        of draws is the eager path's: one call here per train.Trainer.step().
        The host never waits for the GPU under replay, so it may be several steps ahead: the draws go through a RING of
        pinned staging buffers, each guarded by the event of the copy that last read it -- a slot is rewritten only after
        its DMA has run (one pinned buffer per side was torn / reused by later draws).  The static buffers are used by
        calculate_additional_labels only while `_static_kappa_active` is set (Trainer sets it around capture / replay), so
        an eager step on the same model draws afresh."""
        cfg = self.config
        if not (self.training and cfg.refine_net_do_offset_augmentation):
            return
        std = np.radians(cfg.refine_net_offset_augmentation_sigma)
        static = getattr(self, '_static_kappa', None)
        if static is None or static['both'].shape[1:3] != (B, T) or static['both'].device != device:
            both = torch.empty((2, B, T, 2), dtype=torch.float32, device=device)
            static = {'both': both, 'left': both[0], 'right': both[1]}
            cuda = device.type == 'cuda'
            self._static_ring = [{'host': torch.empty((2, B, T, 2), dtype=torch.float32).pin_memory() if cuda
                                  else torch.empty((2, B, T, 2), dtype=torch.float32),
                                  'event': torch.cuda.Event() if cuda else None, 'used': False}
                                 for _ in range(self.STATIC_KAPPA_RING)]
            self._static_ring_pos = 0
            self._static_kappa = static
        slot = self._static_ring[self._static_ring_pos]
        self._static_ring_pos = (self._static_ring_pos + 1) % len(self._static_ring)
        if slot['used'] and slot['event'] is not None:
            slot['event'].synchronize()              # the copy that read this slot STATIC_KAPPA_RING steps ago has run
        kap = self._draw_kappa(B, T, std)
        slot['host'][0].copy_(torch.from_numpy(kap['left']))
        slot['host'][1].copy_(torch.from_numpy(kap['right']))
        static['both'].copy_(slot['host'], non_blocking=True)
        if slot['event'] is not None:
            slot['event'].record()
        slot['used'] = True

    STATIC_KAPPA_RING = 4
    _static_kappa_active = False

    def drop_static_kappa(self):
        self._static_kappa = None
        self._static_kappa_active = False

    # ------------------------------------------------------------------------------------------ labels (eve.py:441-543)
    def calculate_additional_labels(self, d, current_epoch=None):
        cfg = self.config
        sample = next(iter(d.values()))
        B, T = sample.shape[0], sample.shape[1]
        dev = sample.device
        k = default_kernels()
        for side in ('left', 'right'):
            if side + '_PoG_tobii' in d:
                d[side + '_PoG_cm_tobii'] = (d[side + '_PoG_tobii'] * (0.1 * d['millimeters_per_pixel'])).detach()
                d[side + '_PoG_cm_tobii_validity'] = d[side + '_PoG_tobii_validity']
        if self.training and cfg.refine_net_do_offset_augmentation:
            assert isinstance(current_epoch, float)
            std = np.radians(cfg.refine_net_offset_augmentation_sigma)
            static = getattr(self, '_static_kappa', None) if self._static_kappa_active else None
            if static is not None and tuple(static['left'].shape[:2]) == (B, T):
                # hipGraph replay (train.eve_trainer(use_graph=True)): the draw happened on the host BEFORE the replay
                # (refresh_static_kappa) and sits in fixed device buffers the captured kernels read; any other call (an
                # eager step on the same model, another batch shape) draws afresh below
                for side in ('left', 'right'):
                    d[side + '_kappa_fake'] = static[side]
            else:
                for side, kap in self._draw_kappa(B, T, std).items():
                    d[side + '_kappa_fake'] = torch.from_numpy(kap).to(dev)
        if 'left_o' in d:
            d['o'] = _mean2(d['left_o'], d['right_o']).detach()
            d['o_validity'] = d['left_o_validity']
        if 'left_PoG_tobii' in d:
            d['PoG_px_tobii'] = _mean2(d['left_PoG_tobii'], d['right_PoG_tobii']).detach()
            d['PoG_cm_tobii'] = _mean2(d['left_PoG_cm_tobii'], d['right_PoG_cm_tobii']).detach()
            valid = (d['left_PoG_tobii_validity'].bool() & d['right_PoG_tobii_validity'].bool()).detach()
            d['PoG_px_tobii_validity'] = d['PoG_cm_tobii_validity'] = valid
            if cfg.refine_net_enabled:
                w, h = cfg.gaze_heatmap_size
                centres = d['PoG_px_tobii'].reshape(B * T, 2).float()
                for name, sigma in (('initial', cfg.gaze_heatmap_sigma_initial), ('history', cfg.gaze_heatmap_sigma_history),
                                    ('final', cfg.gaze_heatmap_sigma_final)):
                    m = k.make_heatmaps(centres, sigma, (h, w), cfg.actual_screen_size, validity=valid.reshape(B * T))
                    d['heatmap_' + name] = m.view(B, T, 1, h, w)
                    d['heatmap_%s_validity' % name] = valid
        if 'PoG_cm_tobii' in d:
            d['g'] = k.combined_gaze(d['o'].reshape(B * T, 3), (10.0 * d['PoG_cm_tobii']).reshape(B * T, 2),
                                     d['left_R'].reshape(B * T, 3, 3), d['camera_transformation'].reshape(B * T, 4, 4)).view(B, T, 2)
            d['g_validity'] = d['PoG_cm_tobii_validity']

    # ------------------------------------------------------------------------------------------ eve.py:545-601
    def _pog_block(self, d, inter, g_key, out_suffix, kappa_suffix=None, heatmap_history=False):
        """Per-eye PoG from `<side>_g_<g_key>`, their mean, the combined gaze, the heat-map -- for all B*T frames.  With
        `kappa_suffix` the offset augmentation is applied first and the augmented angles are stored back under
        `<side>_g_<kappa_suffix>`."""
        if 'inv_camera_transformation' not in d:                      # GazeCapture / MPIIGaze style inputs
            if kappa_suffix is not None:
                self._augment_only(d, inter, g_key, kappa_suffix)
            return
        cfg = self.config
        B, T = d['inv_camera_transformation'].shape[:2]
        N = B * T
        flat = lambda key, *s: d[key].reshape(N, *s).float()
        for side in ('left', 'right'):
            origin = inter[side + '_o'] if side + '_o' in inter else d[side + '_o']
            rot = inter[side + '_R'] if side + '_R' in inter else d[side + '_R']
            head_R = kappa = None
            if kappa_suffix is not None:
                head_R, kappa = flat('head_R', 3, 3), flat(side + '_kappa_fake', 2)
            g_out, mm, px = ops.GazeToPoGFn.apply(
                inter['%s_g_%s' % (side, g_key)].reshape(N, 2), origin.reshape(N, 3).float(), rot.reshape(N, 3, 3).float(),
                flat('inv_camera_transformation', 4, 4), flat('pixels_per_millimeter', 2), tuple(cfg.actual_screen_size),
                head_R, kappa)
            if kappa_suffix is not None:
                inter['%s_g_%s' % (side, kappa_suffix)] = g_out.view(B, T, 2)
            inter['%s_PoG_cm_%s' % (side, out_suffix)] = (0.1 * mm).view(B, T, 2)
            inter['%s_PoG_px_%s' % (side, out_suffix)] = px.view(B, T, 2)
        for unit in ('px', 'cm'):
            inter['PoG_%s_%s' % (unit, out_suffix)] = _mean2(inter['left_PoG_%s_%s' % (unit, out_suffix)],
                                                            inter['right_PoG_%s_%s' % (unit, out_suffix)])
        inter['PoG_mm_' + out_suffix] = 10.0 * inter['PoG_cm_' + out_suffix]
        inter['g_' + out_suffix] = default_kernels().combined_gaze(
            flat('o', 3), inter['PoG_mm_' + out_suffix].detach().reshape(N, 2), flat('left_R', 3, 3),
            flat('camera_transformation', 4, 4)).view(B, T, 2)
        if cfg.refine_net_enabled:
            w, h = cfg.gaze_heatmap_size
            centres = inter['PoG_px_' + out_suffix].reshape(N, 2)
            inter['heatmap_' + out_suffix] = ops.MakeHeatmapsFn.apply(
                centres, cfg.gaze_heatmap_sigma_initial, (h, w), tuple(cfg.actual_screen_size)).view(B, T, 1, h, w)
            if heatmap_history and 'PoG_px_tobii' in d:
                hist = default_kernels().make_heatmaps(centres.detach(), cfg.gaze_heatmap_sigma_history, (h, w),
                                                       tuple(cfg.actual_screen_size)).view(B, T, 1, h, w)
                inter['history_' + out_suffix] = self._history_maps(d['timestamps'], hist, d['PoG_px_tobii_validity'])

    def _augment_only(self, d, inter, g_key, kappa_suffix):
        """apply_offset_augmentation without camera geometry: the kernel's PoG outputs are computed against an identity
        camera and discarded (only head_R and kappa enter the augmented angles)."""
        B, T = d['head_R'].shape[:2]
        N = B * T
        dev = d['head_R'].device
        eye3 = torch.eye(3, device=dev).expand(N, 3, 3).contiguous()
        eye4 = torch.eye(4, device=dev).expand(N, 4, 4).contiguous()
        origin = torch.tensor([0.0, 0.0, 1.0], device=dev).expand(N, 3).contiguous()
        for side in ('left', 'right'):
            g_out, _, _ = ops.GazeToPoGFn.apply(
                inter['%s_g_%s' % (side, g_key)].reshape(N, 2), origin, eye3, eye4, torch.ones((N, 2), device=dev),
                tuple(self.config.actual_screen_size), d['head_R'].reshape(N, 3, 3).float(),
                d[side + '_kappa_fake'].reshape(N, 2).float())
            inter['%s_g_%s' % (side, kappa_suffix)] = g_out.view(B, T, 2)

    def _history_maps(self, timestamps, heatmaps, validity):
        """common.py:256-297 for the full window: per clip, sum over frames with a non-zero timestamp of
        validity * decay^(ms before the clip's last non-zero timestamp) * heat-map.  Visualisation only."""
        ts = timestamps.to(torch.int64)
        nz = ts != 0
        idx = torch.arange(ts.shape[1], device=ts.device).expand_as(ts)
        last = torch.where(nz, idx, torch.full_like(idx, -1)).max(dim=1).values.clamp(min=0)
        target = ts.gather(1, last.view(-1, 1))
        diff_ms = ((target - ts) * 1e-6).float()
        wgt = torch.pow(torch.tensor(self.config.gaze_history_map_decay_per_ms, device=ts.device), diff_ms)
        wgt = wgt * nz.float() * validity.float()
        return (heatmaps * wgt.view(*wgt.shape, 1, 1, 1)).sum(dim=1)

    # ------------------------------------------------------------------------------------------ eve.py:69-284
    def forward(self, full_input_dict, create_images=False, current_epoch=None):
        cfg = self.config
        if self.training:
            assert len(full_input_dict) == 1
            full_input_dict = next(iter(full_input_dict.values()))
        d = full_input_dict
        self.calculate_additional_labels(d, current_epoch=current_epoch)
        B, T = d['left_eye_patch'].shape[:2]

        inter = dict(self.eye_net.forward_sequence(d))                # eve.py:105-111 for every t
        if self.training and cfg.refine_net_do_offset_augmentation:   # eve.py:114-135
            self._pog_block(d, inter, 'initial', 'initial_unaugmented')
            for side in ('left', 'right'):
                inter[side + '_g_initial_unaugmented'] = inter[side + '_g_initial']
            # augmented angles -> `<side>_g_initial`; the PoG block on them is both 'initial_augmented' and 'initial'
            self._pog_block(d, inter, 'initial_unaugmented', 'initial', kappa_suffix='initial', heatmap_history=create_images)
            for key in list(inter.keys()):
                if key.endswith('_initial') and ('PoG' in key or key in ('g_initial', 'heatmap_initial')):
                    inter[key + '_augmented'] = inter[key]
        else:
            self._pog_block(d, inter, 'initial', 'initial', heatmap_history=create_images)   # eve.py:138-143

        refined_history = None
        if self.refine_net is not None:                               # eve.py:146-166
            hf, states = self.refine_net.forward_sequence(inter['heatmap_initial'], d.get('screen_frame'))
            inter['heatmap_final'] = hf
            for i, st in enumerate(states):
                if not isinstance(st, tuple):
                    inter['refinenet_rnn_states_%d' % i] = st
            h, w = hf.shape[-2:]
            px = ops.SoftArgmaxFn.apply(hf.reshape(B * T, 1, h, w).float(), tuple(cfg.actual_screen_size)).view(B, T, 2)
            inter['PoG_px_final'] = px
            inter['PoG_cm_final'] = px * (0.1 * d['millimeters_per_pixel'])
            inter['g_final'] = default_kernels().combined_gaze(
                d['o'].reshape(B * T, 3).float(), (10.0 * inter['PoG_cm_final']).detach().reshape(B * T, 2),
                d['left_R'].reshape(B * T, 3, 3).float(), d['camera_transformation'].reshape(B * T, 4, 4).float()).view(B, T, 2)
            if create_images and 'PoG_px_tobii' in d:
                refined_history = self._history_maps(d['timestamps'], hf.detach().float(), d['PoG_px_tobii_validity'])

        output_dict = {k_: v for k_, v in inter.items() if k_.startswith('output_')}
        output_dict['left_pupil_size'] = inter['left_pupil_size']
        output_dict['right_pupil_size'] = inter['right_pupil_size']
        if self.output_predictions:                                   # eve.py:194-222
            for key in ('timestamps', 'o', 'left_R', 'head_R', 'millimeters_per_pixel', 'pixels_per_millimeter',
                        'camera_transformation', 'inv_camera_transformation'):
                output_dict[key] = d[key]
            for key in ('g_initial', 'PoG_px_initial', 'PoG_cm_initial'):
                output_dict[key] = inter[key]
            if 'g' in d:
                output_dict['g'] = d['g']
                output_dict['validity'] = d['PoG_px_tobii_validity']
                output_dict['PoG_cm'] = d['PoG_cm_tobii']
                output_dict['PoG_px'] = d['PoG_px_tobii']
            if self.refine_net is not None:
                for key in ('g_final', 'PoG_px_final', 'PoG_cm_final'):
                    output_dict[key] = inter[key]

        self.calculate_losses_and_metrics(d, inter, output_dict)
        output_dict['full_loss'] = self._full_loss(output_dict, d['left_eye_patch'].device)

        if create_images:                                             # eve.py:268-283
            if cfg.load_screen_content:
                output_dict['screen_frame'] = d['screen_frame'][:, -1]
            if 'history_initial' in inter:
                output_dict['initial_gaze_history'] = inter['history_initial']
            if 'heatmap_initial' in inter:
                output_dict['initial_heatmap'] = inter['heatmap_initial'][:, -1]
            if 'heatmap_final' in inter:
                output_dict['final_heatmap'] = inter['heatmap_final'][:, -1]
                output_dict['refined_gaze_history'] = refined_history
            if 'heatmap_final' in d:
                output_dict['gt_heatmap'] = d['heatmap_final'][:, -1]
        return output_dict

    # ------------------------------------------------------------------------------------------ eve.py:286-439
    def calculate_losses_and_metrics(self, d, inter, out):
        cfg = self.config
        augment = self.training and cfg.refine_net_do_offset_augmentation
        un = '_unaugmented' if augment else ''

        # the [B, T, D <= 3] terms are collected and evaluated by ONE kernel launch (ops.VectorTermsFn); heat-map terms, CPU
        # tensors and Euclidean terms that would need a gradient keep their own path
        batched = []
        kinds = {losses.mse_loss: 'mse', losses.euclidean_loss: 'euclidean', losses.l1_loss: 'l1', losses.angular_loss: 'angular'}

        def term(name, fn, interm_key, input_key, ref=None):
            ref = d if ref is None else ref
            if interm_key in inter and input_key in ref:
                pred, tgt, val = inter[interm_key], ref[input_key], ref[input_key + '_validity']
                kind = kinds.get(fn)
                if (kind is not None and pred.is_cuda and pred.dim() <= 3 and pred.dtype == torch.float32 and
                        tgt.dtype == torch.float32 and tuple(tgt.shape) == tuple(pred.shape) and not tgt.requires_grad and
                        (pred.dim() == 2 or pred.shape[2] <= 3) and not (kind == 'euclidean' and pred.requires_grad and
                                                                          torch.is_grad_enabled()) and
                        hasattr(default_kernels(), 'vector_terms')):
                    out[name] = None                         # (keeps the reference's key order)
                    batched.append((name, kind, pred, tgt, val))
                else:
                    out[name] = fn(pred, tgt, val)

        for side in ('left', 'right'):
            term('loss_ang_%s_g_initial' % side, losses.angular_loss, '%s_g_initial%s' % (side, un), side + '_g_tobii')
            term('loss_mse_%s_PoG_cm_initial' % side, losses.mse_loss, '%s_PoG_cm_initial%s' % (side, un), side + '_PoG_cm_tobii')
            term('metric_euc_%s_PoG_cm_initial' % side, losses.euclidean_loss, '%s_PoG_cm_initial%s' % (side, un), side + '_PoG_cm_tobii')
            term('metric_euc_%s_PoG_px_initial' % side, losses.euclidean_loss, side + '_PoG_px_initial', side + '_PoG_tobii')
            term('loss_l1_%s_pupil_size' % side, losses.l1_loss, side + '_pupil_size', side + '_p')
        if 'left_PoG_tobii' in d and 'right_PoG_tobii' in d:         # left-right consistency
            inter['right_PoG_cm_initial_validity'] = d['left_PoG_tobii_validity'] & d['right_PoG_tobii_validity']
            term('loss_mse_lr_consistency', losses.mse_loss, 'left_PoG_cm_initial', 'right_PoG_cm_initial', inter)
            term('metric_euc_lr_consistency', losses.euclidean_loss, 'left_PoG_cm_initial', 'right_PoG_cm_initial', inter)
        term('loss_ce_heatmap_initial', losses.bce_loss, 'heatmap_initial' + un, 'heatmap_initial')
        term('loss_ce_heatmap_final', losses.bce_loss, 'heatmap_final', 'heatmap_final')
        term('loss_mse_heatmap_final', losses.mse_loss, 'heatmap_final', 'heatmap_final')
        if cfg.refine_net_do_offset_augmentation:
            term('metric_euc_PoG_px_initial_unaugmented', losses.euclidean_loss, 'PoG_px_initial_unaugmented', 'PoG_px_tobii')
            term('metric_euc_PoG_cm_initial_unaugmented', losses.euclidean_loss, 'PoG_cm_initial_unaugmented', 'PoG_cm_tobii')
            term('metric_ang_g_initial_unaugmented', losses.angular_loss, 'g_initial_unaugmented', 'g')
        for stage in ('initial', 'final'):
            term('loss_mse_PoG_px_' + stage, losses.mse_loss, 'PoG_px_' + stage, 'PoG_px_tobii')
            term('metric_euc_PoG_px_' + stage, losses.euclidean_loss, 'PoG_px_' + stage, 'PoG_px_tobii')
            term('loss_mse_PoG_cm_' + stage, losses.mse_loss, 'PoG_cm_' + stage, 'PoG_cm_tobii')
            term('metric_euc_PoG_cm_' + stage, losses.euclidean_loss, 'PoG_cm_' + stage, 'PoG_cm_tobii')
            term('metric_ang_g_' + stage, losses.angular_loss, 'g_' + stage, 'g')
        if batched:
            vals = ops.VectorTermsFn.apply(tuple((kind, tgt, val) for _, kind, _, tgt, val in batched), *[b[2] for b in batched])
            for (name, _, _, _, _), v in zip(batched, vals):
                out[name] = v

    def _full_loss(self, out, device):                                # eve.py:234-265
        cfg = self.config
        total = torch.zeros((), device=device)
        if 'loss_ang_left_g_initial' in out:
            total = total + cfg.loss_coeff_g_ang_initial * (out['loss_ang_left_g_initial'] + out['loss_ang_right_g_initial'])
        if 'loss_mse_left_PoG_cm_initial' in out and cfg.loss_coeff_PoG_cm_initial > 0.0:
            total = total + cfg.loss_coeff_PoG_cm_initial * (out['loss_mse_left_PoG_cm_initial'] + out['loss_mse_right_PoG_cm_initial'])
        if 'loss_l1_left_pupil_size' in out:
            total = total + cfg.loss_coeff_pupil_size * (out['loss_l1_left_pupil_size'] + out['loss_l1_right_pupil_size'])
        if 'loss_mse_PoG_cm_final' in out:
            total = total + cfg.loss_coeff_PoG_cm_final * out['loss_mse_PoG_cm_final']
        if 'loss_ce_heatmap_initial' in out:
            total = total + cfg.loss_coeff_heatmap_ce_initial * out['loss_ce_heatmap_initial']
        if 'loss_ce_heatmap_final' in out:
            total = total + cfg.loss_coeff_heatmap_ce_final * out['loss_ce_heatmap_final']
        if 'loss_mse_heatmap_final' in out:
            total = total + cfg.loss_coeff_heatmap_mse_final * out['loss_mse_heatmap_final']
        return total
