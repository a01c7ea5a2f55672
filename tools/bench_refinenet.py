#!/usr/bin/env python
"""RefineNet train step (refine_net.json losses: heat-map BCE + MSE; CGRU bottleneck) on one GPU: ms/step and frames/s.
Not the headline metric (BASELINE configs[1] is EyeNet training) -- a companion measurement for SURVEY 8 rows a5-a10."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eve_amd  # noqa: E402
from eve_amd import train  # noqa: E402
from eve_amd import synthetic as detweights  # noqa: E402  (synthetic clips and weights)

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--seq', type=int, default=30)
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--dtype', default='bf16')
args = ap.parse_args()

cfg = eve_amd.reset_standalone_config()
cfg.import_json(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs', 'refine_net.json'))
cfg.import_dict({'refine_net_rnn_type': 'CGRU'})
net = eve_amd.RefineNet()
net.compute_dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
detweights.fill_module(net, seed=0)
net = net.cuda()
tr = train.refinenet_trainer(net, cfg)
batch = {k: v.cuda() for k, v in detweights.refinenet_batch(args.batch, args.seq, seed=1).items()}
for _ in range(2):
    tr.step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    terms = tr.step(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
print('refinenet train step: %.2f ms, %.0f frames/s (B=%d, T=%d, %s), loss %.5f' % (
    1e3 * dt, args.batch * args.seq / dt, args.batch, args.seq, args.dtype, float(terms['full_loss'])))
