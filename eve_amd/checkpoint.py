"""Checkpoints in the reference's on-disk layout, so that runs can move between the two code bases.

/root/reference/src/core/checkpoint_manager.py:47-149: a checkpoint is a DIRECTORY `<dir>/checkpoints/%07d.pt/` holding one
file per first-level module of the model -- `eye_net.pt`, `refine_net.pt`, each a dict of that module's
`state_dict()` entries WITH the prefix kept in the keys -- plus `optimizer_<i>.pt`; loading merges every non-optimizer
file and calls a strict `load_state_dict`; only the newest `keep_n` directories are kept.  The drop-in modules keep the
reference's parameter names and torch OIHW float32 shapes, so these files are interchangeable with the reference's (and
with the released `eve_eyenet_*.pt` / `eve_refinenet_*.pt` weights, which use the same keys without the prefix).
Optimizer state: the build's trainer keeps Adam moments in flat buffers (train.FlatParameters); they are written in
torch.optim.Adam's state_dict format, numbered by position in `model.parameters()` with the frozen parameters counted
(train.py:49-55 hands Adam every parameter) and state entries only for the trainable ones -- what `optimizer_0.pt`
holds in the reference.
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
        torch.save(adam_state_dict(trainer, model), os.path.join(ofdir, 'optimizer_0' + SUFFIX))
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
        load_adam_state_dict(trainer, torch.load(opt, map_location=map_location), model)
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


def _positions(trainer, model):
    """Index of every trainer parameter in `list(model.parameters())` -- the numbering torch.optim.Adam(model.parameters())
    uses in its state_dict (/root/reference/src/train.py:49-55 builds the optimizer over ALL parameters, frozen ones
    included, so with a frozen EyeNet the RefineNet entries start at 37).  Without a model: 0..len-1 over the trainable ones."""
    if model is None:
        return list(range(len(trainer.fp.entries))), len(trainer.fp.entries)
    index = {id(p): i for i, p in enumerate(model.parameters())}
    pos = []
    for p, _, _ in trainer.fp.entries:
        if id(p) not in index:
            raise ValueError('a trainer parameter is not a parameter of the model the optimizer state is numbered by')
        pos.append(index[id(p)])
    return pos, len(index)


def adam_state_dict(trainer, model=None):
    """torch.optim.Adam(model.parameters()).state_dict() of the trainer's moments: `state` holds an entry per TRAINABLE
    parameter under its position in model.parameters(), `param_groups[0]['params']` lists every position."""
    cfg, fp = trainer.config, trainer.fp
    pos, total = _positions(trainer, model)
    state = {}
    # Adam's `step` = optimiser steps actually TAKEN (the device-resident counter: a step skipped for a non-finite float16
    # gradient does not count), not the number of step() calls
    steps_taken = int(trainer.guard[0]) if hasattr(trainer, 'guard') else int(trainer.step_count)
    for i, (p, off, n) in zip(pos, fp.entries):
        state[i] = {'step': torch.tensor(float(steps_taken)),
                    'exp_avg': _param_view(fp.m, off, n, p).detach().cpu().contiguous(),
                    'exp_avg_sq': _param_view(fp.v, off, n, p).detach().cpu().contiguous()}
    lr = float(trainer.lr) if getattr(trainer, 'lr', None) is not None else float(cfg.learning_rate)
    # 'initial_lr' is what torch's LambdaLR (which the reference always wraps Adam in, training.py:436-442) multiplies its
    # lambda by; without it a resumed reference run would take the decayed, quirk-scaled `lr` as its base
    group = {'lr': lr, 'initial_lr': float(cfg.learning_rate), 'betas': (trainer.beta1, trainer.beta2), 'eps': trainer.eps,
             'weight_decay': float(cfg.weight_decay), 'amsgrad': False, 'maximize': False, 'foreach': None,
             'capturable': False, 'differentiable': False, 'fused': None, 'params': list(range(total))}
    out = {'state': state, 'param_groups': [group]}
    if hasattr(trainer, 'guard'):
        # what torch.optim.Adam has no slot for (the reference is float32 and never skips): the loss scale after back-offs,
        # the skip counters and the number of step() CALLS, which is what drives the LR schedule.  torch's load_state_dict
        # ignores unknown top-level keys, so the file still loads into the reference's optimiser.
        st = trainer.optimizer_state()
        out['eve_amd'] = {'loss_scale': st['loss_scale'], 'steps_skipped': st['steps_skipped'], 'steps_taken': st['steps_taken'],
                          'step_calls': int(trainer.step_count), 'guard': trainer.guard.detach().cpu().clone()}
    return out


def load_adam_state_dict(trainer, sd, model=None):
    """Inverse of adam_state_dict; every entry's shapes are checked BEFORE anything is copied (a state numbered by another
    parameter list must not land in the wrong tensors)."""
    fp = trainer.fp
    pos, total = _positions(trainer, model)
    listed = sd['param_groups'][0]['params'] if sd.get('param_groups') else None
    if listed is not None and len(listed) != total:
        raise ValueError('optimizer state lists %d parameters, the %s has %d' % (
            len(listed), 'model' if model is not None else 'trainer', total))
    todo = []
    for i, (p, off, n) in zip(pos, fp.entries):
        st = sd['state'].get(i)
        if st is None:
            continue
        for key in ('exp_avg', 'exp_avg_sq'):
            if tuple(st[key].shape) != tuple(p.shape):
                raise ValueError('optimizer state %d %s has shape %s, parameter has %s' % (
                    i, key, tuple(st[key].shape), tuple(p.shape)))
        todo.append((st, p, off, n))
    steps = 0
    for st, p, off, n in todo:
        _param_view(fp.m, off, n, p).copy_(st['exp_avg'].to(fp.m.device))
        _param_view(fp.v, off, n, p).copy_(st['exp_avg_sq'].to(fp.v.device))
        steps = max(steps, int(float(st['step'])))
    trainer.step_count = steps
    trainer.guard[0] = steps
    extra = sd.get('eve_amd')
    if extra is not None:
        # resumed float16 runs keep their backed-off loss scale and skip counters; the LR schedule continues from the number
        # of step() calls (skipped steps included), Adam's exponent from the steps taken
        if extra.get('guard') is not None and tuple(extra['guard'].shape) == tuple(trainer.guard.shape):
            trainer.guard.copy_(extra['guard'].to(trainer.guard.device))
            trainer.guard[0] = steps
        trainer.step_count = int(extra.get('step_calls', steps))
