"""Checkpoints in the reference's on-disk layout, so that runs can move between the two code bases.

/root/reference/src/core/checkpoint_manager.py:47-149: a checkpoint is a DIRECTORY `<dir>/checkpoints/%07d.pt/` holding one
file per first-level module of the model -- `eye_net.pt`, `refine_net.pt`, each a dict of that module's
`state_dict()` entries WITH the prefix kept in the keys -- plus `optimizer_<i>.pt`; loading merges every non-optimizer
file and calls a strict `load_state_dict`; only the newest `keep_n` directories are kept.  The drop-in modules keep the
reference's parameter names and torch OIHW float32 shapes, so these files are interchangeable with the reference's (and
with the released `eve_eyenet_*.pt` / `eve_refinenet_*.pt` weights, which use the same keys without the prefix).
Optimizer state: the build's trainer keeps Adam moments in flat buffers (train.FlatParameters); they are written in
torch.optim.Adam's state_dict format (`state` / `param_groups`, parameters numbered in `model.parameters()` order of the
trainable ones), which is what `optimizer_0.pt` holds in the reference.
"""
import glob
import os
import shutil

import torch

SUFFIX = '.pt'


def checkpoint_dir(output_dir, step):
    return os.path.join(output_dir, 'checkpoints', ('%07d' % step) + SUFFIX)


def available(output_dir):
    """[(step, path)] sorted by step -- checkpoint_manager.py:126-132."""
    found = []
    for fn in glob.glob(os.path.join(output_dir, 'checkpoints', '*' + SUFFIX)):
        if os.path.isdir(fn):
            found.append((int(os.path.split(fn)[-1].split('.')[0]), fn))
    return sorted(found)


def save(model, output_dir, step, trainer=None, keep_n=3):
    ofdir = checkpoint_dir(output_dir, step)
    assert not os.path.isdir(ofdir)
    state = {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    os.makedirs(ofdir)
    for prefix in sorted({k.split('.')[0] for k in state}):
        torch.save({k: v for k, v in state.items() if k.startswith(prefix + '.')}, os.path.join(ofdir, prefix + SUFFIX))
    if trainer is not None:
        torch.save(adam_state_dict(trainer), os.path.join(ofdir, 'optimizer_0' + SUFFIX))
    for _, path in available(output_dir)[:-keep_n] if keep_n else []:
        shutil.rmtree(path)
    return ofdir


def load(model, ifdir, trainer=None, map_location='cpu'):
    """Strict load of every module file in `ifdir`; returns the step encoded in the directory name."""
    assert os.path.isdir(ifdir)
    full = {}
    for p in glob.glob(os.path.join(ifdir, '*' + SUFFIX)):
        if os.path.isfile(p) and not os.path.basename(p).startswith('optimizer_'):
            full.update(torch.load(p, map_location=map_location))
    model.load_state_dict(full)
    for m in model.modules():
        if hasattr(m, 'invalidate_packs'):
            m.invalidate_packs()
    opt = os.path.join(ifdir, 'optimizer_0' + SUFFIX)
    if trainer is not None and os.path.isfile(opt):
        load_adam_state_dict(trainer, torch.load(opt, map_location=map_location))
    return int(os.path.split(ifdir.rstrip('/'))[-1][:-len(SUFFIX)])


def load_last(model, output_dir, trainer=None):
    found = available(output_dir)
    return load(model, found[-1][1], trainer) if found else 0


# ---- Adam moments <-> torch.optim.Adam.state_dict() ------------------------------------------------------------------
def _param_view(flat, off, n, p):
    v = flat[off:off + n]
    if p.dim() == 4:                                   # the flat buffers hold conv weights as OHWI
        O, I, KH, KW = p.shape
        return v.view(O, KH, KW, I).permute(0, 3, 1, 2)
    return v.view(tuple(p.shape))


def adam_state_dict(trainer):
    cfg, fp = trainer.config, trainer.fp
    state = {}
    for i, (p, off, n) in enumerate(fp.entries):
        state[i] = {'step': torch.tensor(float(trainer.step_count)),
                    'exp_avg': _param_view(fp.m, off, n, p).detach().cpu().contiguous(),
                    'exp_avg_sq': _param_view(fp.v, off, n, p).detach().cpu().contiguous()}
    group = {'lr': float(cfg.learning_rate), 'betas': (trainer.beta1, trainer.beta2), 'eps': trainer.eps,
             'weight_decay': float(cfg.weight_decay), 'amsgrad': False, 'params': list(range(len(fp.entries)))}
    return {'state': state, 'param_groups': [group]}


def load_adam_state_dict(trainer, sd):
    fp = trainer.fp
    steps = 0
    for i, (p, off, n) in enumerate(fp.entries):
        st = sd['state'].get(i)
        if st is None:
            continue
        _param_view(fp.m, off, n, p).copy_(st['exp_avg'].to(fp.m.device))
        _param_view(fp.v, off, n, p).copy_(st['exp_avg_sq'].to(fp.v.device))
        steps = max(steps, int(float(st['step'])))
    trainer.step_count = steps
    trainer.step_dev.fill_(steps)
