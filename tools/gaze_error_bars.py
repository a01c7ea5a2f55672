#!/usr/bin/env python
"""Where the float32 HIP path and the float32 CPU oracle sit relative to the oracle's FLOAT64 evaluation, per refine_net.json
variant (B = 2, T = 3, deterministic weights): max |difference| of g_initial / g_final in rad (profiles/r06_notes.md 13).
small_linear 1 = the tail's linear layers on linear_mm_kernel, 0 = through the 1x1-convolution path."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import eve_amd  # noqa: E402
from eve_amd import kernels  # noqa: E402
from oracle import detweights  # noqa: E402
from oracle import eve as oracle_eve  # noqa: E402
from oracle.config import OracleConfig  # noqa: E402
from oracle.eye_net import EyeNet as OracleEyeNet  # noqa: E402
from oracle.refine_net import RefineNet as OracleRefineNet  # noqa: E402

json_path = os.path.join(REPO, 'configs', 'refine_net.json')
VARIANTS = (dict(refine_net_rnn_type='CGRU', refine_net_use_skip_connections=False), dict(refine_net_rnn_type='CLSTM'),
            dict(refine_net_rnn_type='CRNN'), dict(refine_net_rnn_type='CGRU', refine_net_do_offset_augmentation=False))
for over in VARIANTS:
    res = {}
    batch = detweights.eve_batch(2, 3, seed=23, invalid_fraction=0.2)
    for dt in (torch.float32, torch.float64):
        torch.set_default_dtype(dt)
        ocfg = OracleConfig(json_path, eye_net_load_pretrained=False, **over)
        oeye, oref = detweights.fill_module(OracleEyeNet(ocfg), 0).to(dt), detweights.fill_module(OracleRefineNet(ocfg), 1).to(dt)
        b2 = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
        np.random.seed(2)
        with torch.no_grad():
            _, inter, _ = oracle_eve.eve_forward(oeye, oref, dict(b2), ocfg, True)
        res[dt] = {k: inter[k].detach().double() for k in ('g_initial', 'g_final')}
    torch.set_default_dtype(torch.float32)
    for small in (1, 0):
        k = kernels.default_kernels()
        with k.dispatch_override(small_linear=small):
            cfg = eve_amd.reset_standalone_config()
            cfg.import_json(json_path)
            cfg.import_dict(dict(eye_net_load_pretrained=False, **over))
            model = eve_amd.EVE(output_predictions=True)
            model.eye_net.compute_dtype = model.refine_net.compute_dtype = torch.float32
            detweights.fill_module(model.eye_net, 0)
            detweights.fill_module(model.refine_net, 1)
            model = model.cuda().train()
            np.random.seed(2)
            with torch.no_grad():
                got = model({'s': {kk: v.cuda() for kk, v in batch.items()}}, current_epoch=0.0)
        for kk in ('g_initial', 'g_final'):
            g = got[kk].detach().cpu().double()
            print('%-70s small_linear %d %-9s hip-o32 %.3e  hip-o64 %.3e  o32-o64 %.3e' % (
                ','.join('%s=%s' % (a.replace('refine_net_', ''), b) for a, b in over.items()), small, kk,
                float((g - res[torch.float32][kk]).abs().max()), float((g - res[torch.float64][kk]).abs().max()),
                float((res[torch.float32][kk] - res[torch.float64][kk]).abs().max())))
