#!/usr/bin/env python
"""Train step of the whole EVE harness on one GPU -- SURVEY.md 8(d) config C3: configs/refine_net.json (EyeNet frozen,
forward only; RefineNet trained; BCE(heat-map) + 1e-3 MSE(PoG cm)) with refine_net_rnn_type=CGRU, through eve_amd.EVE
(label synthesis, offset augmentation, gaze geometry, heat-maps, soft-argmax, 31 losses/metrics).  ms/step and frames/s;
--joint trains both networks.  A companion measurement to bench.py (whose metric is BASELINE configs[1])."""
import argparse
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import eve_amd  # noqa: E402
from eve_amd import train  # noqa: E402
from eve_amd import synthetic as detweights  # noqa: E402  (synthetic clips and weights)

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--seq', type=int, default=30)
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--joint', action='store_true')
args = ap.parse_args()

cfg = eve_amd.reset_standalone_config()
cfg.import_json(os.path.join(REPO, 'configs', 'refine_net.json'))
cfg.import_dict({'refine_net_rnn_type': 'CGRU', 'eye_net_load_pretrained': False})
if args.joint:
    cfg.import_dict({'eye_net_frozen': False, 'loss_coeff_g_ang_initial': 1.0, 'loss_coeff_pupil_size': 1.0})
model = eve_amd.EVE()
dt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
model.eye_net.compute_dtype = model.refine_net.compute_dtype = dt
detweights.fill_module(model.eye_net, seed=0)
detweights.fill_module(model.refine_net, seed=1)
model = model.cuda().train()
tr = train.eve_trainer(model, cfg)
small = detweights.eve_batch(4, args.seq, seed=1)
reps = (args.batch + 3) // 4
batch = {k: torch.cat([v] * reps, dim=0)[:args.batch].contiguous().cuda() for k, v in small.items()}
np.random.seed(0)
for _ in range(2):
    terms = tr.step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    terms = tr.step(batch)
torch.cuda.synchronize()
dtm = (time.perf_counter() - t0) / args.steps
print('EVE %s train step: %.2f ms, %.0f frames/s (B=%d, T=%d, %s), full_loss %.5f, PoG error %.1f px' % (
    'joint' if args.joint else 'C3 (EyeNet frozen)', 1e3 * dtm, args.batch * args.seq / dtm, args.batch, args.seq, args.dtype,
    float(terms['full_loss']), float(terms['metric_euc_PoG_px_final'])))
