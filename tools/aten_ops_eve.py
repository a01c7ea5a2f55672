#!/usr/bin/env python
"""Which ATen launches are left in a configs[2] step, and where they come from: one eager eve_trainer step under torch.profiler
with Python stacks; prints the device-kernel-launching aten ops grouped by the innermost eve_amd / bench frame.
   python tools/aten_ops_eve.py [batch]"""
import collections
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import eve_amd  # noqa: E402
from eve_amd import train  # noqa: E402
from eve_amd import synthetic as detweights  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = eve_amd.reset_standalone_config()
cfg.import_json(os.path.join(REPO, 'configs', 'refine_net.json'))
cfg.import_dict({'refine_net_rnn_type': 'CGRU', 'eye_net_load_pretrained': False})
model = eve_amd.EVE()
model.eye_net.compute_dtype = model.refine_net.compute_dtype = torch.bfloat16
detweights.fill_module(model.eye_net, seed=0)
detweights.fill_module(model.refine_net, seed=1)
model = model.cuda().train()
tr = train.eve_trainer(model, cfg)
batch = {k: v.cuda() for k, v in detweights.eve_batch(B, 30, seed=1).items()}
np.random.seed(0)
for _ in range(2):
    tr.step(batch)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
count = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.device_time_total <= 0 or not ev.kernels:
        continue
    where = '?'
    for fr in (ev.stack or []):
        if 'eve_amd/' in fr or 'bench' in fr or 'tools/' in fr:
            where = fr.split('eve_amd/')[-1] if 'eve_amd/' in fr else fr
            break
    count[(where.strip()[:110], ev.name)] += len(ev.kernels)
tot = sum(count.values())
print('aten ops that launched kernels in one eager step: %d launches' % tot)
for (where, name), n in count.most_common(60):
    print('%4d  %-28s %s' % (n, name, where))
