#!/usr/bin/env python
"""List every conv launch of one RefineNet train step with its shape and the kernel symbol the library dispatched."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eve_amd  # noqa: E402
from eve_amd import kernels, train  # noqa: E402
from eve_amd import synthetic as detweights  # noqa: E402  (synthetic clips and weights)

cfg = eve_amd.reset_standalone_config()
cfg.import_json(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs', 'refine_net.json'))
cfg.import_dict({'refine_net_rnn_type': 'CGRU'})
net = eve_amd.RefineNet()
net.compute_dtype = torch.bfloat16
detweights.fill_module(net, seed=0)
net = net.cuda()
tr = train.refinenet_trainer(net, cfg)
batch = {k: v.cuda() for k, v in detweights.refinenet_batch(4, 6, seed=1).items()}
tr.step(batch)
k = kernels.default_kernels()
log = collections.Counter()
for name in ('conv2d_fwd', 'conv2d_dgrad', 'conv2d_wgrad'):
    orig = getattr(k, name)

    def wrap(*a, _orig=orig, _name=name, **kw):
        r = _orig(*a, **kw)
        x = a[0]
        w = a[1] if _name != 'conv2d_wgrad' else None
        desc = (_name, tuple(x.shape[1:]), tuple(w.shape) if w is not None else tuple(a[1].shape[1:]), k.lib.eve_last_kernel().decode())
        log[desc] += 1
        return r
    setattr(k, name, wrap)
tr.step(batch)
for d, n in sorted(log.items(), key=lambda t: (t[0][3], t[0][0])):
    print(n, d)
