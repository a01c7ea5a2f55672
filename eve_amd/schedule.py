"""Learning-rate schedule of the reference's training loop.

/root/reference/src/core/training.py:382-418 `learning_rate_schedule(optimizer, epoch_len, log, step)`: linear warm-up
from base_lr = target_lr / batch_size to target_lr = batch_size * base_learning_rate over
`num_warmup_epochs * epoch_len` steps, then `lr_decay_strategy`: 'exponential' (step function target_lr * factor^k every
`lr_decay_epoch_interval` epochs), 'cyclic' (down to base_lr and up to the next, decayed, peak inside every interval) or
anything else (constant target_lr).

The quirk (:436-442, :576-577): the loop hands that function to `torch.optim.lr_scheduler.LambdaLR`, which MULTIPLIES the
function's value by the optimizer's initial LR -- and the function already returns an absolute LR.  The learning rate the
reference's `optimizer.step()` of training step s really uses is therefore

        effective_lr(s) = target_lr * learning_rate_schedule(s)            (e.g. 0.016^2 = 2.56e-4 for eye_net.json)

`effective_learning_rate(..., reference_quirk=True)` reproduces it (a drop-in must train like the reference does, so it
is the default); `reference_quirk=False` gives the schedule as its author evidently meant it.  Both are pinned by
tests/golden/lr_schedule.npz, produced by the reference's own function driving torch's LambdaLR.
"""
import math


def learning_rate_schedule(config, epoch_len, step, target_lr=None, base_lr=None):
    target_lr = float(config.learning_rate) if target_lr is None else float(target_lr)
    base_lr = target_lr / float(config.batch_size) if base_lr is None else float(base_lr)
    warmup = int(epoch_len * config.num_warmup_epochs)
    if step < warmup:
        return (target_lr - base_lr) / float(warmup) * step + base_lr
    epoch = (step - warmup) / float(epoch_len)
    k = int(epoch / config.lr_decay_epoch_interval)
    if config.lr_decay_strategy == 'exponential':
        return target_lr * math.pow(config.lr_decay_factor, k)
    if config.lr_decay_strategy == 'cyclic':
        # every interval goes down from this interval's peak to base_lr, then up to the next (decayed) peak
        peak_a = target_lr * math.pow(config.lr_decay_factor, k)
        peak_b = peak_a * config.lr_decay_factor
        half = 0.5 * config.lr_decay_epoch_interval
        middle = k * config.lr_decay_epoch_interval + half
        slope = -(peak_a - base_lr) / half if epoch < middle else (peak_b - base_lr) / half
        return slope * (epoch - middle) + base_lr
    return target_lr


def effective_learning_rate(config, epoch_len, step, reference_quirk=True):
    """The LR of training step `step` (0-based): what param_groups[0]['lr'] holds when the reference steps."""
    lr = learning_rate_schedule(config, epoch_len, step)
    return float(config.learning_rate) * lr if reference_quirk else lr
