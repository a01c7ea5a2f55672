#!/usr/bin/env python
"""Sanity check, not a test: optimisation must make progress on a small synthetic data set that can be memorised.
  (1) EyeNet (configs/eye_net.json, bf16, hipGraph replay): angular error of 8 clips x 10 frames falls from ~50 to a few degrees
  (2) the whole EVE pipeline (refine_net.json with CGRU, EyeNet frozen, offset augmentation on): RefineNet learns to move
      the heat-map towards the labelled point of gaze -- BCE and the final PoG error fall
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

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
DT_NAME = sys.argv[2] if len(sys.argv) > 2 else 'bf16'          # bf16 | fp16 (static loss scale in train.Trainer)
DT = {'bf16': torch.bfloat16, 'fp16': torch.float16}[DT_NAME]

cfg = eve_amd.reset_standalone_config()
cfg.import_json(os.path.join(REPO, 'configs', 'eye_net.json'))
cfg.import_dict({'base_learning_rate': 0.000125})     # lr = 16 x this = 0.002 (eye_net.json's 0.016 is sized for real data + decay)
net = eve_amd.EyeNet()
net.compute_dtype = DT
net = net.cuda()
tr = train.eyenet_trainer(net, cfg, use_graph=True)
batch = {k: v.cuda() for k, v in detweights.eyenet_batch(8, 10, seed=3).items()}
print('EyeNet, 8 clips x 10 frames, %s, lr %.4f' % (DT_NAME, cfg.learning_rate))
for i in range(steps + 1):
    t = tr.step(batch)
    if i % (steps // 8) == 0:
        print('  step %4d  full %.4f  angular L %.3f deg  R %.3f deg  pupil L1 %.4f' % (
            i, float(t['full_loss'].detach()), float(t['loss_ang_left_g_initial'].detach()),
            float(t['loss_ang_right_g_initial'].detach()), float(t['loss_l1_left_pupil_size'].detach())))

cfg = eve_amd.reset_standalone_config()
cfg.import_json(os.path.join(REPO, 'configs', 'refine_net.json'))
cfg.import_dict({'refine_net_rnn_type': 'CGRU', 'eye_net_load_pretrained': False})
model = eve_amd.EVE()
model.eye_net.load_state_dict(net.state_dict())          # the EyeNet trained above, now frozen (refine_net.json)
model.eye_net.compute_dtype = model.refine_net.compute_dtype = DT
model = model.cuda().train()
tr = train.eve_trainer(model, cfg)
batch = {k: v.cuda() for k, v in detweights.eve_batch(8, 10, seed=3).items()}
np.random.seed(0)
print('EVE pipeline (EyeNet frozen, RefineNet/CGRU trained), 8 clips x 10 frames, %s, lr %.4f' % (DT_NAME, cfg.learning_rate))
for i in range(steps + 1):
    t = tr.step(batch)
    if i % (steps // 8) == 0:
        print('  step %4d  full %.4f  BCE(heat-map) %.4f  PoG error initial %.1f px -> final %.1f px  (%.2f cm)' % (
            i, float(t['full_loss'].detach()), float(t['loss_ce_heatmap_final'].detach()), float(t['metric_euc_PoG_px_initial']),
            float(t['metric_euc_PoG_px_final'].detach()), float(t['metric_euc_PoG_cm_final'].detach())))
