#!/usr/bin/env python
"""Sanity check, not a test: a few dozen EyeNet training steps (hipGraph replay, bf16) on one synthetic batch -- the loss
must fall."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eve_amd
from eve_amd import train
from oracle import detweights
cfg = eve_amd.reset_standalone_config()
cfg.import_json(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs', 'eye_net.json'))
net = eve_amd.EyeNet(); net.compute_dtype = torch.bfloat16
net = net.cuda()
tr = train.eyenet_trainer(net, cfg, use_graph=True)
batch = {k: v.cuda() for k, v in detweights.eyenet_batch(8, 10, seed=3).items()}
for i in range(41):
    t = tr.step(batch)
    if i % 8 == 0: print(i, float(t['full_loss']), float(t['loss_ang_left_g_initial']), float(t['loss_l1_left_pupil_size']))
