#!/usr/bin/env python
"""Golden vectors of the reference's learning-rate schedule (src/core/training.py:382-418) and of the learning rate
the optimizer ACTUALLY steps with when that function drives torch's LambdaLR the way the training loop does
(:436-442 construct, :576-577 `lr_scheduler.step(current_step + 1)`): LambdaLR multiplies the function's value by the
initial LR, and the function already returns an absolute LR, so the effective LR is target_lr * schedule(step).

    python tests/golden/make_golden_schedule.py        (build container only: imports /root/reference/src/core/training.py)
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))
import make_golden  # noqa: E402  (the reference import recipe: logging / IO stubs)

CASES = {
    # name: (config overrides, epoch_len, steps)
    'none': (dict(batch_size=16, base_learning_rate=0.001, num_warmup_epochs=0.0, lr_decay_strategy='none'), 50, 120),
    'eye_net_json': (dict(batch_size=16, base_learning_rate=0.001, num_warmup_epochs=0.0, lr_decay_strategy='exponential',
                          lr_decay_factor=0.5, lr_decay_epoch_interval=1.0), 40, 130),
    'warmup_exponential': (dict(batch_size=8, base_learning_rate=0.0005, num_warmup_epochs=0.5, lr_decay_strategy='exponential',
                                lr_decay_factor=0.5, lr_decay_epoch_interval=0.5), 60, 200),
    'warmup_cyclic': (dict(batch_size=8, base_learning_rate=0.0005, num_warmup_epochs=0.25, lr_decay_strategy='cyclic',
                           lr_decay_factor=0.7, lr_decay_epoch_interval=0.4), 50, 200),
}


def main():
    config = make_golden.import_reference()
    m = types.ModuleType('coloredlogs')
    m.install = lambda *a, **k: None
    sys.modules['coloredlogs'] = m
    from core import training
    fix = {}
    for name, (over, epoch_len, steps) in CASES.items():
        for k, v in over.items():
            config.override(k, v)
        w = torch.nn.Parameter(torch.zeros(3))
        opt = torch.optim.Adam([w], lr=config.learning_rate)
        opt.target_lr = opt.param_groups[0]['lr']                      # training.py:216-217
        opt.base_lr = opt.target_lr / config.batch_size
        sched_values = [float(training.learning_rate_schedule(opt, epoch_len, lambda v: None, s)) for s in range(steps)]
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            import functools
            sch = torch.optim.lr_scheduler.LambdaLR(
                opt, functools.partial(training.learning_rate_schedule, opt, epoch_len, lambda v: None))
            used = []
            for s in range(steps):
                used.append(float(opt.param_groups[0]['lr']))          # the LR optimizer.step() of training step s uses
                w.grad = torch.ones(3)
                opt.step()
                sch.step(s + 1)                                        # training.py:576-577
        fix[name + '_schedule'] = np.array(sched_values, np.float64)
        fix[name + '_effective'] = np.array(used, np.float64)
        fix[name + '_epoch_len'] = epoch_len
        for k, v in over.items():
            fix['%s_cfg_%s' % (name, k)] = v
        print(name, sched_values[:3], used[:3], used[-1])
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'lr_schedule.npz'), **fix)


if __name__ == '__main__':
    main()
