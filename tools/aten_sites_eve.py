#!/usr/bin/env python
"""Call sites of the ATen operations left in one eager configs[2] step (TorchDispatchMode + Python stacks, autograd on the calling
thread so that the backward's operations are seen too):  python tools/aten_sites_eve.py [batch]"""
import collections
import os
import sys
import traceback

import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode

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
torch.autograd.set_multithreading_enabled(False)
for _ in range(2):
    tr.step(batch)
torch.cuda.synchronize()
VIEWS = {'view', 'reshape', '_unsafe_view', 'as_strided', 'select', 'slice', 'expand', 'permute', 'transpose', 't', 'unsqueeze',
         'squeeze', 'detach', 'alias', 'empty', 'empty_like', 'empty_strided', 'unbind', 'split', 'narrow', '_local_scalar_dense',
         'item', 'lift_fresh', 'unflatten', 'view_as', 'chunk', 'new_empty', 'split_with_sizes', 'sym_size', 'set_', 'resize_',
         'is_pinned', '_reshape_alias', 'new_empty_strided', 'record_stream', 'unsafe_split', 'sym_numel', 'sym_stride',
         'sym_storage_offset', 'is_same_size'}
count = collections.Counter()


class Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.name().split('::')[-1].split('.')[0]
        if name in VIEWS:
            return out
        where = '?'
        for fr in reversed(traceback.extract_stack()):
            if '/eve_amd/' in fr.filename:
                where = '%s:%d %s' % (fr.filename.split('/eve_amd/')[-1], fr.lineno, fr.name)
                break
        count[(where, name)] += 1
        return out


with Sites():
    tr.step(batch)
torch.cuda.synchronize()
print('ATen operations in one eager step: %d' % sum(count.values()))
for (w, n), c in sorted(count.items(), key=lambda t: -t[1])[:120]:
    print('%4d  %-26s %s' % (c, n, w))
