#!/usr/bin/env python
"""FULL float64 gradient tensors of the EVE harness cases, from the reference's own models.eve.EVE.

Run here only (needs /root/reference):
    python tests/golden/make_golden_eve_grads.py

Same two training cases, batch, weights and kappa draw as make_golden_eve.py (`c3`: refine_net.json + CGRU, EyeNet frozen,
offset augmentation on; `joint`: both networks trained, every loss coefficient non-zero), evaluated twice -- in float32 (what
eve_harness.npz holds) and in FLOAT64.  Stored per case and network: every parameter's float64 gradient norm, how far the
reference's OWN float32 evaluation is from its float64 one per parameter (relative L2 of the whole tensor: ReLU / max-pool /
adaptive-pool decisions on float ties re-route gradient in any float32 evaluation), and the complete float64 gradient of
representative parameters.  tests/test_gpu_eve.py holds the float32 HIP path to max(1e-4, 2 x that deviation) on the full
tensors instead of the former 3e-2 on norms.  (The reference builds kappa_fake as a float32 tensor whatever the default dtype,
eve.py:474, and torch.matmul does not promote: for the float64 run the harness hands apply_offset_augmentation the SAME
kappa values widened to float64 -- a cast of an input, every float32 is a float64 -- through a wrapper around the name
models.eve imported; no reference code is changed.)
Only numbers are written (tests/golden/eve_grads_f64.npz)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import OUT, REF_SRC, import_reference  # noqa: E402
from oracle import detweights  # noqa: E402

B, T = 2, 4
FULL = {'refine_net': ('initial.0.weight', 'final.2.weight', 'network.encoder_blocks.0.layers.2.weight',
                       'network.decoder_blocks.0.layers.5.weight'),
        'eye_net': ('cnn_layers.conv1.weight', 'cnn_layers.layer1.0.conv1.weight', 'rnn_cells.0.weight_hh', 'fc_to_gaze.0.weight')}


def run(EVE, dt, batch_args=(B, T, 0, 0.25), np_seed=0):
    torch.set_default_dtype(dt)
    eve = EVE(output_predictions=True)
    detweights.fill_module(eve.eye_net, seed=0)
    detweights.fill_module(eve.refine_net, seed=1)
    eve = eve.to(dt).train()
    b_, t_, seed, invalid = batch_args
    batch = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in detweights.eve_batch(b_, t_, seed=seed, invalid_fraction=invalid).items()}
    np.random.seed(np_seed)
    out = eve({'synthetic': batch}, current_epoch=0.0)
    out['full_loss'].backward()
    grads = {net: {n: (None if p.grad is None else p.grad.detach().double().clone()) for n, p in getattr(eve, net).named_parameters()}
             for net in ('eye_net', 'refine_net')}
    return float(out['full_loss'].detach()), grads


def record(fix, tag, l32, g32, l64, g64):
    fix['%s_full_loss_f32' % tag], fix['%s_full_loss_f64' % tag] = np.float64(l32), np.float64(l64)
    for net in ('eye_net', 'refine_net'):
        live = [n for n, g in g64[net].items() if g is not None]
        fix['%s_%s_dead' % (tag, net)] = np.array([n for n, g in g64[net].items() if g is None] or [''])
        if not live:
            continue
        fix['%s_%s_names' % (tag, net)] = np.array(live)
        fix['%s_%s_norms' % (tag, net)] = np.array([float(g64[net][n].norm()) for n in live], np.float64)
        fix['%s_%s_ref_f32_dev' % (tag, net)] = np.array(
            [float((g32[net][n] - g64[net][n]).norm() / max(float(g64[net][n].norm()), 1e-30)) for n in live], np.float64)
        for n in FULL[net]:
            if n in live:
                fix['%s_%s_grad_%s' % (tag, net, n)] = g64[net][n].float().numpy()
        dev = fix['%s_%s_ref_f32_dev' % (tag, net)][fix['%s_%s_norms' % (tag, net)] > 1e-9 * fix['%s_%s_norms' % (tag, net)].max()]
        print('%s %s: %d parameters, reference float32 vs float64 worst %.2e, median %.2e' % (tag, net, len(live), dev.max(), np.median(dev)))


# the configuration variants of tests/test_gpu_eve.py::test_eve_config_variants_match_oracle (its batch and kappa seed)
VARIANTS = {'clstm': dict(refine_net_rnn_type='CLSTM'), 'crnn': dict(refine_net_rnn_type='CRNN'),
            'noskip': dict(refine_net_rnn_type='CGRU', refine_net_use_skip_connections=False),
            'noaug': dict(refine_net_rnn_type='CGRU', refine_net_do_offset_augmentation=False)}


def main():
    torch.set_num_threads(8)
    config = import_reference()
    import models.eve as ref_eve
    from models.eve import EVE
    orig_aug = ref_eve.apply_offset_augmentation

    def aug_same_dtype(gaze, head_rotation, kappa, *a, **k):
        return orig_aug(gaze, head_rotation, kappa.to(gaze.dtype), *a, **k)
    ref_eve.apply_offset_augmentation = aug_same_dtype
    # ... likewise the label heat-maps the reference synthesises in float32 (common.py:226-243) meet float64 predictions in
    # F.binary_cross_entropy / mse, which refuse mixed dtypes: the TARGET is widened (exactly) to the prediction's dtype
    import torch.nn.functional as F
    orig_bce, orig_mse = F.binary_cross_entropy, F.mse_loss
    F.binary_cross_entropy = lambda a, b, *x, **k: orig_bce(a, b.to(a.dtype), *x, **k)
    F.mse_loss = lambda a, b, *x, **k: orig_mse(a, b.to(a.dtype), *x, **k)
    config.import_json(os.path.join(REF_SRC, 'configs', 'refine_net.json'))
    config.override('eye_net_load_pretrained', False)
    fix = {'B': B, 'T': T}
    # ---- the configuration variants first (EyeNet frozen, refine_net.json's own loss coefficients) ----
    defaults = dict(refine_net_rnn_type='CLSTM', refine_net_use_skip_connections=True, refine_net_do_offset_augmentation=True)
    for tag, over in VARIANTS.items():
        for k, v in dict(defaults, **over).items():
            config.override(k, v)
        args = dict(batch_args=(2, 3, 23, 0.2), np_seed=2)
        l32, g32 = run(EVE, torch.float32, **args)
        l64, g64 = run(EVE, torch.float64, **args)
        record(fix, tag, l32, g32, l64, g64)
    for k, v in dict(defaults, refine_net_rnn_type='CGRU').items():
        config.override(k, v)
    for tag in ('c3', 'joint'):
        if tag == 'joint':
            for k, v in (('eye_net_frozen', False), ('loss_coeff_PoG_cm_initial', 0.002), ('loss_coeff_g_ang_initial', 1.0),
                         ('loss_coeff_pupil_size', 1.0), ('loss_coeff_heatmap_ce_initial', 0.0),
                         ('loss_coeff_heatmap_mse_final', 0.5), ('loss_coeff_PoG_cm_final', 0.01)):
                config.override(k, v)
        l32, g32 = run(EVE, torch.float32)
        l64, g64 = run(EVE, torch.float64)
        record(fix, tag, l32, g32, l64, g64)
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(OUT, 'eve_grads_f64.npz'), **fix)
    print('eve_grads_f64.npz: %.0f KB' % (os.path.getsize(os.path.join(OUT, 'eve_grads_f64.npz')) / 1024))


if __name__ == '__main__':
    main()
