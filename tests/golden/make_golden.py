#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in the build container.

Run here only (needs /root/reference; the GPU box does not have it):
    python tests/golden/make_golden.py

What it does
  * stubs the logging/IO modules the reference's ``core/__init__.py`` eagerly imports
    (gspread, oauth2client, tensorboardX -- absent in this image, no compute in them),
  * injects ``oracle.resnet_in`` as ``torchvision.models.resnet`` (torchvision 0.6.1 is the
    un-vendored dependency of src/models/eye_net.py:26 -- see oracle/resnet_in.py),
  * imports the reference's own ``models.eye_net.EyeNet``, ``models.refine_net.RefineNet``,
    ``models.common.C*Cell`` and ``models.eve.EVE`` from /root/reference/src,
  * loads the deterministic weights/inputs of oracle/detweights.py into them and stores
    inputs-by-seed + expected outputs as small .npz files.
Only data (numbers) is written; no reference source travels.
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF_SRC = '/root/reference/src'
OUT = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, REPO)

from oracle import detweights, resnet_in  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    class _Dummy(object):
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None

    _stub('gspread')
    _stub('oauth2client')
    _stub('oauth2client.service_account', ServiceAccountCredentials=_Dummy)
    _stub('tensorboardX', SummaryWriter=_Dummy)
    tv = _stub('torchvision')
    tv.utils = _stub('torchvision.utils', make_grid=lambda *a, **k: None)
    tv.models = _stub('torchvision.models')
    tv.models.resnet = _stub('torchvision.models.resnet',
                             ResNet=resnet_in.ResNet, BasicBlock=resnet_in.BasicBlock)
    os.chdir(REF_SRC)
    sys.argv[0] = os.path.join(REF_SRC, 'train.py')
    sys.path.insert(0, REF_SRC)
    from core import DefaultConfig
    return DefaultConfig()


def np_(t):
    return t.detach().cpu().numpy().astype(np.float32)


def grad_summary(module):
    """Per-parameter gradient L2 norm (None -> -1) and the first 8 gradient values."""
    names, norms, heads = [], [], []
    for n, p in module.named_parameters():
        names.append(n)
        if p.grad is None:
            norms.append(-1.0)
            heads.append(np.zeros(8, np.float32))
        else:
            g = p.grad.detach().reshape(-1)
            norms.append(float(g.double().norm()))
            h = np.zeros(8, np.float32)
            k = min(8, g.numel())
            h[:k] = np_(g[:k])
            heads.append(h)
    return np.array(names), np.array(norms, np.float64), np.stack(heads)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    config = import_reference()
    from models.common import CGRUCell, CLSTMCell, CRNNCell
    from models.eye_net import EyeNet
    from models.refine_net import RefineNet
    from models.eve import EVE

    # ------------------------------------------------------------------ EyeNet (eye_net.json)
    config.import_json(os.path.join(REF_SRC, 'configs', 'eye_net.json'))
    B, T = 2, 3
    batch = detweights.eyenet_batch(B, T, seed=0, invalid_fraction=0.25)

    eye_net = detweights.fill_module(EyeNet(), seed=0)
    # (1) config[0]: single-frame forward, batch 2, through the reference EyeNet.forward
    trunk_taps = {}
    handles = []
    for nm in ('conv1', 'maxpool', 'layer1', 'layer2', 'layer3', 'layer4', 'fc'):
        handles.append(getattr(eye_net.cnn_layers, nm).register_forward_hook(
            lambda m, i, o, nm=nm: trunk_taps.__setitem__(nm, o.detach())))
    sub_in = {k: v[:, 0] for k, v in batch.items()}
    out0 = {}
    eye_net(sub_in, out0, side='left')
    eye_net(sub_in, out0, side='right')          # hooks now hold the right-eye taps
    for h in handles:
        h.remove()
    fix = {'B': B, 'T': T, 'seed': 0, 'invalid_fraction': 0.25}
    for k, v in out0.items():
        fix['frame0_' + k] = np_(v)
    for nm, v in trunk_taps.items():
        v = v.double()
        fix['tap_right_%s_sum' % nm] = v.sum().numpy()
        fix['tap_right_%s_sqsum' % nm] = (v * v).sum().numpy()
        fix['tap_right_%s_head' % nm] = v.reshape(v.shape[0], -1)[:, :64].float().numpy()

    # (2) the sequence through the reference per-step contract (state hand-over by dict)
    steps = []
    for t in range(T):
        sub_in = {k: v[:, t] for k, v in batch.items()}
        sub_out = {}
        prev = steps[-1] if steps else None
        eye_net(sub_in, sub_out, side='left', previous_output_dict=prev)
        eye_net(sub_in, sub_out, side='right', previous_output_dict=prev)
        steps.append(sub_out)
    for k in steps[0]:
        fix['seq_' + k] = np_(torch.stack([s[k] for s in steps], dim=1))

    # (3) one full reference EVE train-mode forward + backward (losses, masking, full_loss, grads).
    #     The batch has no camera geometry, the case eve.py:551-553 handles by returning early;
    #     offset augmentation still runs and needs head_R (identity) -- it does not reach the loss.
    eve = EVE()
    eve.eye_net.load_state_dict(eye_net.state_dict())
    eve.train()
    np.random.seed(0)
    full = dict(batch)
    full['head_R'] = torch.eye(3).expand(B, T, 3, 3).contiguous()
    eve_out = eve({'synthetic': full}, current_epoch=0.0)
    for k, v in eve_out.items():
        if isinstance(v, torch.Tensor) and v.dim() == 0:
            fix['eve_' + k] = np_(v)
    eve_out['full_loss'].backward()
    names, norms, heads = grad_summary(eve.eye_net)
    fix['grad_names'], fix['grad_norms'], fix['grad_heads'] = names, norms, heads
    # clip + Adam step exactly as train.py:49-55 / training.py:492-502, then a few updated weights
    opt = torch.optim.Adam(eve.eye_net.parameters(), lr=config.learning_rate,
                           weight_decay=config.weight_decay)
    total_norm = torch.nn.utils.clip_grad_norm_(eve.parameters(), config.gradient_clip_amount)
    opt.step()
    fix['clip_total_norm'] = np_(total_norm)
    sd = eve.eye_net.state_dict()
    for k in ('cnn_layers.conv1.weight', 'cnn_layers.layer4.1.conv2.weight', 'fc_to_gaze.2.weight',
              'rnn_cells.0.weight_hh', 'fc_common.0.bias'):
        fix['updated_' + k] = np_(sd[k].reshape(-1)[:16])
    np.savez_compressed(os.path.join(OUT, 'eyenet.npz'), **fix)
    print('eyenet.npz', {k: float(v) for k, v in fix.items() if k.startswith('eve_')})

    # ------------------------------------------------------------------ conv-RNN cells alone
    cfix = {}
    for kind, cls in (('CGRU', CGRUCell), ('CLSTM', CLSTMCell), ('CRNN', CRNNCell)):
        cell = detweights.fill_module(cls(64, 64), seed=3)
        g = torch.Generator().manual_seed(7)
        x = torch.randn(2, 64, 5, 8, generator=g).requires_grad_()
        h = (0.5 * torch.randn(2, 64, 5, 8, generator=g)).requires_grad_()
        if kind == 'CLSTM':
            c = (0.5 * torch.randn(2, 64, 5, 8, generator=g)).requires_grad_()
            hn, cn = cell(x, (h, c))
            (hn.sum() + 0.5 * (cn * cn).sum()).backward()
            cfix[kind + '_c'], cfix[kind + '_c_new'], cfix[kind + '_dc'] = np_(c), np_(cn), np_(c.grad)
        else:
            hn = cell(x, h)
            (hn * hn).sum().backward()
        cfix[kind + '_x'], cfix[kind + '_h'], cfix[kind + '_h_new'] = np_(x), np_(h), np_(hn)
        cfix[kind + '_dx'], cfix[kind + '_dh'] = np_(x.grad), np_(h.grad)
        h0 = cell(x.detach())                      # previous_states=None -> zero state
        cfix[kind + '_h_new_from_none'] = np_(h0[0] if isinstance(h0, tuple) else h0)
        names, norms, heads = grad_summary(cell)
        cfix[kind + '_grad_names'], cfix[kind + '_grad_norms'] = names, norms
    np.savez_compressed(os.path.join(OUT, 'cells.npz'), **cfix)
    print('cells.npz written')

    # ------------------------------------------------------------------ RefineNet (refine_net.json keys)
    config.override('load_screen_content', True)
    config.override('refine_net_enabled', True)
    from losses.cross_entropy import CrossEntropyLoss
    from losses.mse import MSELoss
    rb = detweights.refinenet_batch(2, 3, seed=0, invalid_fraction=0.25)
    rfix = {'B': 2, 'T': 3, 'seed': 0, 'invalid_fraction': 0.25}
    for kind in ('CGRU', 'CLSTM', 'CRNN'):
        config.override('refine_net_rnn_type', kind)
        net = detweights.fill_module(RefineNet(), seed=1)
        outs, prev, states = [], None, []
        for t in range(3):
            sub_in = {'screen_frame': rb['screen_frame'][:, t]}
            sub_out = {'heatmap_initial': rb['heatmap_initial'][:, t]}
            net(sub_in, sub_out, previous_output_dict=prev)
            outs.append(sub_out['heatmap_final'])
            states.append(sub_out['refinenet_rnn_states_0'])
            prev = sub_out
        hf = torch.stack(outs, dim=1)
        ref = {'heatmap_final': rb['heatmap_final_gt'], 'heatmap_final_validity': rb['validity']}
        ce = CrossEntropyLoss()(hf, 'heatmap_final', ref)
        mse = MSELoss()(hf, 'heatmap_final', ref)
        (1.0 * ce + 0.0 * mse).backward()
        rfix[kind + '_heatmap_final'] = np_(hf) if kind == 'CGRU' else np_(hf[..., ::4, ::4])
        last = states[-1]
        rfix[kind + '_state_last'] = np_(last[0] if isinstance(last, tuple) else last)
        if isinstance(last, tuple):
            rfix[kind + '_cell_last'] = np_(last[1])
        rfix[kind + '_loss_ce'], rfix[kind + '_loss_mse'] = np_(ce), np_(mse)
        names, norms, heads = grad_summary(net)
        rfix[kind + '_grad_names'], rfix[kind + '_grad_norms'], rfix[kind + '_grad_heads'] = \
            names, norms, heads
        print(kind, 'ce', float(ce), 'mse', float(mse),
              'dead grads', int((norms < 0).sum()), 'of', len(norms))
    # 1-channel input variant (load_screen_content False, refine_net.py:183)
    config.override('load_screen_content', False)
    config.override('refine_net_rnn_type', 'CGRU')
    net = detweights.fill_module(RefineNet(), seed=1)
    sub_out = {'heatmap_initial': rb['heatmap_initial'][:, 0]}
    net({}, sub_out)
    rfix['noscreen_heatmap_final'] = np_(sub_out['heatmap_final'][..., ::4, ::4])
    np.savez_compressed(os.path.join(OUT, 'refinenet.npz'), **rfix)
    print('refinenet.npz written')


if __name__ == '__main__':
    main()
