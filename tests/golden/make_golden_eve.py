#!/usr/bin/env python
"""Golden fixture for the EVE sequence harness: RUNS THE REFERENCE's own models.eve.EVE in the
build container on the deterministic clip batch of oracle/detweights.eve_batch and stores its outputs.

Run here only (needs /root/reference):
    python tests/golden/make_golden_eve.py

Cases (all B=2, T=4, 25 % invalid labels, deterministic weights, np.random.seed(0) for the kappa draw):
  c3     configs/refine_net.json with refine_net_rnn_type=CGRU, train mode: EyeNet frozen, offset augmentation on,
         losses BCE(heat-map) + 1e-3 MSE(PoG cm)                                      
  joint  both networks trainable, every loss coefficient non-zero (gradients through the gaze geometry, the Gaussian
         heat-maps' consumers, soft-argmax), train mode
  eval   the c3 model in eval mode (no augmentation), create_images=True (gaze history maps)
Only numbers are written (tests/golden/eve_harness.npz); no reference source travels.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import OUT, REF_SRC, grad_summary, import_reference, np_  # noqa: E402
from oracle import detweights  # noqa: E402

B, T = 2, 4
PRED_KEYS = ('g_initial', 'PoG_px_initial', 'PoG_cm_initial', 'g_final', 'PoG_px_final', 'PoG_cm_final',
             'left_pupil_size', 'right_pupil_size')


def run_case(EVE, tag, fix, train, create_images=False, weights=None):
    eve = EVE(output_predictions=True)
    if weights is None:
        detweights.fill_module(eve.eye_net, seed=0)
        detweights.fill_module(eve.refine_net, seed=1)
    else:
        eve.load_state_dict(weights)
    eve.train(train)
    taps = {'heatmap_initial': [], 'left_g_initial': [], 'right_g_initial': []}

    def grab(module, args):
        for k in taps:
            taps[k].append(args[1][k].detach().clone())
    h = eve.refine_net.register_forward_pre_hook(grab)
    batch = detweights.eve_batch(B, T, seed=0, invalid_fraction=0.25)
    np.random.seed(0)
    out = eve({'synthetic': batch} if train else batch, create_images=create_images, current_epoch=0.0)
    h.remove()
    for k, v in out.items():
        if isinstance(v, torch.Tensor) and v.dim() == 0:
            fix['%s_%s' % (tag, k)] = np_(v)
    for k in PRED_KEYS:
        fix['%s_%s' % (tag, k)] = np_(out[k])
    for k, v in taps.items():
        v = torch.stack(v, dim=1)
        fix['%s_%s' % (tag, k)] = np_(v[..., ::4, ::4]) if v.dim() == 5 else np_(v)
    # labels synthesised by the reference (eve.py:441-543) land in the input dict
    for k in ('g', 'PoG_px_tobii', 'PoG_cm_tobii', 'o'):
        fix['%s_label_%s' % (tag, k)] = np_(batch[k])
    fix['%s_label_heatmap_final' % tag] = np_(batch['heatmap_final'][..., ::4, ::4])
    fix['%s_label_validity' % tag] = batch['PoG_px_tobii_validity'].numpy()
    if train:
        fix['%s_kappa_left' % tag] = np_(batch['left_kappa_fake'])
        fix['%s_kappa_right' % tag] = np_(batch['right_kappa_fake'])
        out['full_loss'].backward()
        for net in ('eye_net', 'refine_net'):
            names, norms, heads = grad_summary(getattr(eve, net))
            fix['%s_%s_grad_names' % (tag, net)], fix['%s_%s_grad_norms' % (tag, net)] = names, norms
            fix['%s_%s_grad_heads' % (tag, net)] = heads
    if create_images:
        for k in ('initial_gaze_history', 'refined_gaze_history', 'initial_heatmap', 'final_heatmap', 'gt_heatmap'):
            fix['%s_%s' % (tag, k)] = np_(out[k][..., ::4, ::4])
    print(tag, {k[len(tag) + 1:]: float(v) for k, v in fix.items()
                if k.startswith(tag + '_') and np.ndim(v) == 0 and ('loss' in k or 'metric' in k)})
    return eve


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    config = import_reference()
    from models.eve import EVE
    config.import_json(os.path.join(REF_SRC, 'configs', 'refine_net.json'))
    config.override('refine_net_rnn_type', 'CGRU')
    config.override('eye_net_load_pretrained', False)          # no network; weights come from oracle/detweights
    fix = {'B': B, 'T': T, 'seed': 0, 'invalid_fraction': 0.25}
    eve = run_case(EVE, 'c3', fix, train=True)
    run_case(EVE, 'eval', fix, train=False, create_images=True, weights=eve.state_dict())
    for k, v in (('eye_net_frozen', False), ('loss_coeff_PoG_cm_initial', 0.002), ('loss_coeff_g_ang_initial', 1.0),
                 ('loss_coeff_pupil_size', 1.0), ('loss_coeff_heatmap_ce_initial', 0.0),
                 ('loss_coeff_heatmap_mse_final', 0.5), ('loss_coeff_PoG_cm_final', 0.01)):
        config.override(k, v)
    run_case(EVE, 'joint', fix, train=True)
    np.savez_compressed(os.path.join(OUT, 'eve_harness.npz'), **fix)
    print('eve_harness.npz: %d arrays, %.0f KB' % (len(fix), os.path.getsize(os.path.join(OUT, 'eve_harness.npz')) / 1024))


if __name__ == '__main__':
    main()
