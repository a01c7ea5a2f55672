#!/usr/bin/env python
"""Sanity check, not a test: optimisation must make progress on a small synthetic data set that can be memorised.
  (1) EyeNet (configs/eye_net.json, bf16, hipGraph replay): angular error of 8 clips x 10 frames falls from ~50 to a few degrees
  (2) the whole EVE pipeline (refine_net.json with CGRU, EyeNet frozen, offset augmentation on): RefineNet learns to move
      the heat-map towards the labelled point of gaze -- BCE and the final PoG error fall
  (3) round 5: what the 16-bit instantiation costs on a TRAINED network -- the weights of (1) / (2) evaluated through the HIP
      path in float32 and in the 16-bit format, on the training clips and on held-out clips: max / rms deviation of the gaze
      angles (rad), the pupil size, the refined heat-map and the final point of gaze (px)
Prints one line every few steps; logs of runs are kept in profiles/rNN_train_sanity.log.   train_sanity.py [steps] [bf16|fp16]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import eve_amd  # noqa: E402
from eve_amd import train  # noqa: E402
from eve_amd import synthetic as detweights  # noqa: E402  (synthetic clips and weights)

def train_eyenet(steps, dt, log=print, clips=8, frames=10):
    cfg = eve_amd.reset_standalone_config()
    cfg.import_json(os.path.join(REPO, 'configs', 'eye_net.json'))
    cfg.import_dict({'base_learning_rate': 0.000125})     # lr = 16 x this = 0.002 (eye_net.json's 0.016 is sized for real data + decay)
    net = eve_amd.EyeNet()
    net.compute_dtype = dt
    net = net.cuda()
    tr = train.eyenet_trainer(net, cfg, use_graph=True)
    batch = {k: v.cuda() for k, v in detweights.eyenet_batch(clips, frames, seed=3).items()}
    log('EyeNet, %d clips x %d frames, %s, lr %.4f' % (clips, frames, str(dt).replace('torch.', ''), cfg.learning_rate))
    for i in range(steps + 1):
        t = tr.step(batch)
        if i % max(1, steps // 8) == 0:
            log('  step %4d  full %.4f  angular L %.3f deg  R %.3f deg  pupil L1 %.4f' % (
                i, float(t['full_loss'].detach()), float(t['loss_ang_left_g_initial'].detach()),
                float(t['loss_ang_right_g_initial'].detach()), float(t['loss_l1_left_pupil_size'].detach())))
    torch.cuda.synchronize()
    del tr
    return net, batch


def eyenet_16bit_vs_fp32(net, dt, batches):
    """The SAME weights through the HIP path in float32 and in `dt`: {name: (max, rms)} of gaze (rad) and pupil deviation."""
    out = {}
    with torch.no_grad():
        for name, batch in batches.items():
            res = {}
            for d in (torch.float32, dt):
                net.compute_dtype = d
                net.invalidate_packs()
                o = net.forward_sequence(batch)
                res[d] = (torch.cat([o['left_g_initial'], o['right_g_initial']]).float(),
                          torch.cat([o['left_pupil_size'], o['right_pupil_size']]).float())
            dg = res[dt][0] - res[torch.float32][0]
            dp = res[dt][1] - res[torch.float32][1]
            out[name] = {'gaze_max_rad': float(dg.abs().max()), 'gaze_rms_rad': float(dg.pow(2).mean().sqrt()),
                         'pupil_max': float(dp.abs().max()), 'pupil_rms': float(dp.pow(2).mean().sqrt()),
                         'gaze_spread_rad': float(res[torch.float32][0].std())}
    net.compute_dtype = dt
    net.invalidate_packs()
    return out


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    dt_name = sys.argv[2] if len(sys.argv) > 2 else 'bf16'          # bf16 | fp16 (static loss scale in train.Trainer)
    dt = {'bf16': torch.bfloat16, 'fp16': torch.float16}[dt_name]
    net, batch = train_eyenet(steps, dt)
    held_out = {k: v.cuda() for k, v in detweights.eyenet_batch(8, 10, seed=41).items()}
    fresh = detweights.fill_module(eve_amd.EyeNet(), seed=0).cuda()     # (the constructor zero-initialises the gaze head's last layer)
    print('%s vs float32 through the HIP path, same weights (gaze in rad; spread = std of the float32 predictions):' % dt_name)
    for tag, n_ in (('untrained (deterministic random) weights', fresh), ('weights after %d %s steps' % (steps, dt_name), net)):
        dev = eyenet_16bit_vs_fp32(n_, dt, {'training clips': batch, 'held-out clips': held_out})
        for name, d in dev.items():
            print('  %-34s %-15s gaze max %.3e rms %.3e (spread %.3e)  pupil max %.3e rms %.3e' % (
                tag, name, d['gaze_max_rad'], d['gaze_rms_rad'], d['gaze_spread_rad'], d['pupil_max'], d['pupil_rms']))
    del fresh

    cfg = eve_amd.reset_standalone_config()
    cfg.import_json(os.path.join(REPO, 'configs', 'refine_net.json'))
    cfg.import_dict({'refine_net_rnn_type': 'CGRU', 'eye_net_load_pretrained': False})
    model = eve_amd.EVE(output_predictions=True)
    model.eye_net.load_state_dict(net.state_dict())          # the EyeNet trained above, now frozen (refine_net.json)
    model.eye_net.compute_dtype = model.refine_net.compute_dtype = dt
    model = model.cuda().train()
    tr = train.eve_trainer(model, cfg)
    batch = {k: v.cuda() for k, v in detweights.eve_batch(8, 10, seed=3).items()}
    np.random.seed(0)
    print('EVE pipeline (EyeNet frozen, RefineNet/CGRU trained), 8 clips x 10 frames, %s, lr %.4f' % (dt_name, cfg.learning_rate))
    for i in range(steps + 1):
        t = tr.step(batch)
        if i % max(1, steps // 8) == 0:
            print('  step %4d  full %.4f  BCE(heat-map) %.4f  PoG error initial %.1f px -> final %.1f px  (%.2f cm)' % (
                i, float(t['full_loss'].detach()), float(t['loss_ce_heatmap_final'].detach()), float(t['metric_euc_PoG_px_initial']),
                float(t['metric_euc_PoG_px_final'].detach()), float(t['metric_euc_PoG_cm_final'].detach())))
    # the trained pipeline, evaluated (eval mode: no augmentation draw) in float32 and in the 16-bit format
    model.eval()
    res = {}
    with torch.no_grad():
        for d in (torch.float32, dt):
            model.eye_net.compute_dtype = model.refine_net.compute_dtype = d
            model.eye_net.invalidate_packs()
            model.refine_net.invalidate_packs()
            o = model(dict(batch))
            res[d] = {k: o[k].float() for k in ('PoG_px_final', 'PoG_px_initial', 'heatmap_final') if k in o}
    for k in res[dt]:
        dd = res[dt][k] - res[torch.float32][k]
        print('  trained pipeline, %s vs float32: %-15s max %.3e rms %.3e' % (dt_name, k, float(dd.abs().max()), float(dd.pow(2).mean().sqrt())))


if __name__ == '__main__':
    main()
