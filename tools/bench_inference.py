#!/usr/bin/env python
"""Forward-only (inference.py / eval_codalab.py style) throughput of the whole EVE pipeline through eve_amd.EVE in eval mode:
EyeNet for both eyes, gaze geometry, heat-maps, RefineNet (CGRU), soft-argmax -- with and without the label-dependent
losses / metrics."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import eve_amd  # noqa: E402
from eve_amd import synthetic as detweights  # noqa: E402  (synthetic clips and weights)

B, T, STEPS = 32, 30, 10
cfg = eve_amd.reset_standalone_config()
cfg.import_json(os.path.join(REPO, 'configs', 'refine_net.json'))
cfg.import_dict({'refine_net_rnn_type': 'CGRU', 'eye_net_load_pretrained': False})
model = eve_amd.EVE(output_predictions=True)
model.eye_net.compute_dtype = model.refine_net.compute_dtype = torch.bfloat16
detweights.fill_module(model.eye_net, 0)
detweights.fill_module(model.refine_net, 1)
model = model.cuda().eval()
small = detweights.eve_batch(4, T, seed=1)
full = {k: torch.cat([v] * (B // 4), dim=0).contiguous().cuda() for k, v in small.items()}
label_keys = [k for k in full if 'tobii' in k or k.endswith('_p') or k.endswith('_p_validity')]
for name, batch in (('with labels (losses + metrics)', full), ('inputs only', {k: v for k, v in full.items() if k not in label_keys})):
    with torch.no_grad():
        for _ in range(2):
            out = model(dict(batch))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            out = model(dict(batch))
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    print('EVE eval forward, %-31s: %.2f ms per %d-frame batch, %.0f frames/s; keys out: %d' % (
        name, 1e3 * dt, B * T, B * T / dt, len(out)))
