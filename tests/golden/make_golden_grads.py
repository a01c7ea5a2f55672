#!/usr/bin/env python
"""FULL gradient tensors of representative parameters, from the reference's own code evaluated in FLOAT64.

Run here only (needs /root/reference):
    python tests/golden/make_golden_grads.py

Why float64: the reference's float32 CPU evaluation is itself 5e-4 (relative L2) away from the exact gradient on the first
trunk layers (ReLU / max-pool decisions on float ties: measured by this script, printed below), so a float32 fixture cannot
hold a 1e-4 assertion -- its float64 evaluation can.  Same classes, same call sequence as make_golden.py:
  * models.eve.EVE (eye_net.json) train-mode forward + backward on oracle.detweights.eyenet_batch(2, 3, seed 0, 25 % invalid);
    offset augmentation off (its kappa draw is float32 numpy and does not reach any EyeNet loss): full gradients of the stem
    convolution, a layer-1 convolution, the GRU's hidden weights and the gaze head's first layer; of the layer-4 convolution
    the [0:64, 0:64] output / input channel block (the whole tensor is 9.4 MB) plus every parameter's float64 norm;
  * models.refine_net.RefineNet (CGRU) per-step forward over T = 3 + the reference's CrossEntropyLoss, backward: full
    gradients of the conv-GRU's gates_1 / gate_2 filter banks, the first and the last convolution.
Only numbers are written (tests/golden/grads_f64.npz, gradients stored as float32 of the float64 values)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import OUT, REF_SRC, import_reference  # noqa: E402
from oracle import detweights  # noqa: E402

EYE_FULL = ('cnn_layers.conv1.weight', 'cnn_layers.layer1.0.conv1.weight', 'rnn_cells.0.weight_hh', 'fc_to_gaze.0.weight')
EYE_BLOCK = 'cnn_layers.layer4.1.conv2.weight'


def main():
    torch.set_num_threads(8)
    config = import_reference()
    from models.eve import EVE
    from models.refine_net import RefineNet
    fix = {}
    # ------------------------------------------------------------------ EyeNet through EVE
    config.import_json(os.path.join(REF_SRC, 'configs', 'eye_net.json'))
    config.override('refine_net_do_offset_augmentation', False)
    B, T = 2, 3
    batch = detweights.eyenet_batch(B, T, seed=0, invalid_fraction=0.25)
    eve = EVE()
    detweights.fill_module(eve.eye_net, seed=0)
    eve.train()
    grads = {}
    for dt in (torch.float32, torch.float64):
        torch.set_default_dtype(dt)
        eve = eve.to(dt)
        for p in eve.parameters():
            p.grad = None
        np.random.seed(0)
        full = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in batch.items()}
        full['head_R'] = torch.eye(3, dtype=dt).expand(B, T, 3, 3).contiguous()
        out = eve({'synthetic': full}, current_epoch=0.0)
        out['full_loss'].backward()
        grads[dt] = {n: p.grad.detach().double().clone() for n, p in eve.eye_net.named_parameters()}
        fix['eye_full_loss_%s' % ('f64' if dt == torch.float64 else 'f32')] = np.float64(out['full_loss'].detach())
    worst = max((float((grads[torch.float32][n] - g).norm() / g.norm()), n) for n, g in grads[torch.float64].items())
    print('reference float32 vs float64 gradients: worst relative L2 %.2e (%s)' % worst)
    fix['eye_ref_f32_vs_f64_worst'] = np.float64(worst[0])
    g64 = grads[torch.float64]
    fix['eye_names'] = np.array(list(g64))
    fix['eye_norms'] = np.array([float(g.norm()) for g in g64.values()], np.float64)
    # how far the reference's own float32 evaluation is from these, per parameter (relative L2 of the full tensor)
    fix['eye_ref_f32_dev'] = np.array([float((grads[torch.float32][n] - g).norm() / g.norm()) for n, g in g64.items()], np.float64)
    for n in EYE_FULL:
        fix['eye_grad_' + n] = g64[n].float().numpy()
    fix['eye_block_' + EYE_BLOCK] = g64[EYE_BLOCK][:64, :64].float().numpy()
    # ------------------------------------------------------------------ RefineNet (CGRU), per-step contract
    config.override('load_screen_content', True)
    config.override('refine_net_enabled', True)
    config.override('refine_net_rnn_type', 'CGRU')
    from losses.cross_entropy import CrossEntropyLoss
    rb = detweights.refinenet_batch(2, 3, seed=0, invalid_fraction=0.25)
    rgs = {}
    for dt in (torch.float32, torch.float64):
        torch.set_default_dtype(dt)
        net = detweights.fill_module(RefineNet(), seed=1).to(dt)
        outs, prev = [], None
        for t in range(3):
            sub_in = {'screen_frame': rb['screen_frame'][:, t].to(dt)}
            sub_out = {'heatmap_initial': rb['heatmap_initial'][:, t].to(dt)}
            net(sub_in, sub_out, previous_output_dict=prev)
            outs.append(sub_out['heatmap_final'])
            prev = sub_out
        hf = torch.stack(outs, dim=1)
        ref = {'heatmap_final': rb['heatmap_final_gt'].to(dt), 'heatmap_final_validity': rb['validity']}
        ce = CrossEntropyLoss()(hf, 'heatmap_final', ref)
        ce.backward()
        rgs[dt] = {n: p.grad.detach().double() for n, p in net.named_parameters() if p.grad is not None}
    fix['refine_loss_ce'] = np.float64(ce.detach())
    rg = rgs[torch.float64]
    # adaptive max-pool / (leaky-)ReLU decisions on float ties re-route gradient in the float32 evaluation: the encoder side
    # of the reference's OWN float32 run is ~5e-3 away from float64, the bottleneck and decoder ~1e-5
    fix['refine_ref_f32_dev'] = np.array([float((rgs[torch.float32][n] - g).norm() / max(float(g.norm()), 1e-30)) for n, g in rg.items()],
                                         np.float64)
    print('reference float32 vs float64, RefineNet: worst %.2e' % float(fix['refine_ref_f32_dev'][np.array([float(g.norm()) for g in rg.values()]) > 1e-9].max()))
    fix['refine_names'] = np.array(list(rg))
    fix['refine_norms'] = np.array([float(g.norm()) for g in rg.values()], np.float64)
    for n in rg:
        if n.endswith(('gates_1.weight', 'gate_2.weight')) or n in ('initial.0.weight', 'final.2.weight'):
            fix['refine_grad_' + n] = rg[n].float().numpy()
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(OUT, 'grads_f64.npz'), **fix)
    print('grads_f64.npz:', sorted(k for k in fix if 'grad_' in k or 'block_' in k))


if __name__ == '__main__':
    main()
