"""Oracle: sequence-level drivers and the train step around the two hot modules.

TEST INFRASTRUCTURE.  Restates, for the terms that carry gradient into the hot
path, what the reference does around EyeNet/RefineNet:
  /root/reference/src/models/eve.py:91-111,172-182   per-time-step loop, state hand-over via the previous
                                                     step's output dict, torch.stack over T
  eve.py:286-325, 234-265                            loss_ang_{left,right}_g_initial, loss_l1_{left,right}_pupil_size,
                                                     weighted by loss_coeff_g_ang_initial / loss_coeff_pupil_size
  eve.py:350-360, 258-263                            loss_ce_heatmap_final / loss_mse_heatmap_final weights
  /root/reference/src/train.py:49-55                 Adam(lr = batch_size * base_lr, weight_decay) (coupled L2)
  /root/reference/src/core/training.py:452,489-502   zero_grad -> backward -> clip_grad_norm_(5.0) -> step
Geometry / heat-map synthesis / soft-argmax are not restated here.
"""
import torch

from . import losses


def eyenet_sequence(eye_net, batch):
    """batch: dict of B x T x ... tensors ({left,right}_eye_patch, {left,right}_h).
    Returns dict of B x T x ... outputs, exactly as eve.py stacks them."""
    T = batch['left_eye_patch'].shape[1]
    steps = []
    for t in range(T):
        sub_in = {k: v[:, t] for k, v in batch.items() if isinstance(v, torch.Tensor)}
        sub_out = {}
        prev = steps[-1] if steps else None
        eye_net(sub_in, sub_out, side='left', previous_output_dict=prev)
        eye_net(sub_in, sub_out, side='right', previous_output_dict=prev)
        steps.append(sub_out)
    return {k: torch.stack([s[k] for s in steps], dim=1)
            for k in steps[0] if isinstance(steps[0][k], torch.Tensor)}


def eyenet_losses(out, batch, config):
    terms = {}
    for side in ('left', 'right'):
        terms['loss_ang_%s_g_initial' % side] = losses.angular_loss(
            out[side + '_g_initial'], batch[side + '_g_tobii'], batch[side + '_g_tobii_validity'])
        terms['loss_l1_%s_pupil_size' % side] = losses.l1_loss(
            out[side + '_pupil_size'], batch[side + '_p'], batch[side + '_p_validity'])
    full = config.loss_coeff_g_ang_initial * (
        terms['loss_ang_left_g_initial'] + terms['loss_ang_right_g_initial'])
    full = full + config.loss_coeff_pupil_size * (
        terms['loss_l1_left_pupil_size'] + terms['loss_l1_right_pupil_size'])
    terms['full_loss'] = full
    return terms


def refinenet_sequence(refine_net, heatmap_initial, screen_frame=None):
    """heatmap_initial: B x T x 1 x H x W; screen_frame: B x T x 3 x H x W or None.
    Returns (heatmap_final B x T x 1 x 72 x 128, list of per-step states)."""
    T = heatmap_initial.shape[1]
    outs, states, prev = [], [], None
    for t in range(T):
        sub_in = {} if screen_frame is None else {'screen_frame': screen_frame[:, t]}
        sub_out = {'heatmap_initial': heatmap_initial[:, t]}
        refine_net(sub_in, sub_out, previous_output_dict=prev)
        outs.append(sub_out['heatmap_final'])
        states.append(sub_out.get('refinenet_rnn_states_0'))
        prev = sub_out
    return torch.stack(outs, dim=1), states


def refinenet_losses(heatmap_final, heatmap_gt, validity, config):
    terms = {
        'loss_ce_heatmap_final': losses.bce_loss(heatmap_final, heatmap_gt, validity),
        'loss_mse_heatmap_final': losses.mse_loss(heatmap_final, heatmap_gt, validity),
    }
    terms['full_loss'] = (config.loss_coeff_heatmap_ce_final * terms['loss_ce_heatmap_final'] +
                          config.loss_coeff_heatmap_mse_final * terms['loss_mse_heatmap_final'])
    return terms


def make_optimizer(params, config):
    return torch.optim.Adam(params, lr=config.learning_rate, weight_decay=config.weight_decay)


def apply_update(parameters, optimizer, config):
    parameters = [p for p in parameters]
    if config.do_gradient_clipping:
        if config.gradient_clip_by == 'norm':
            torch.nn.utils.clip_grad_norm_(parameters, config.gradient_clip_amount)
        else:
            torch.nn.utils.clip_grad_value_(parameters, config.gradient_clip_amount)
    optimizer.step()


def eyenet_train_step(eye_net, optimizer, batch, config):
    optimizer.zero_grad()
    out = eyenet_sequence(eye_net, batch)
    terms = eyenet_losses(out, batch, config)
    terms['full_loss'].backward()
    apply_update(eye_net.parameters(), optimizer, config)
    return out, terms


def lr_schedule(config, epoch_len, step):
    """/root/reference/src/core/training.py:382-418 for one optimizer with target_lr = config.learning_rate and
    base_lr = target_lr / batch_size (:216-217)."""
    target = config.learning_rate
    base = target / config.batch_size
    n_warm = int(epoch_len * config.num_warmup_epochs)
    if step < n_warm:
        return (target - base) / float(n_warm) * step + base
    epoch = (step - n_warm) / float(epoch_len)
    interval = int(epoch / config.lr_decay_epoch_interval)
    if config.lr_decay_strategy == 'exponential':
        return target * config.lr_decay_factor ** interval
    if config.lr_decay_strategy == 'cyclic':
        peak_a = target * config.lr_decay_factor ** interval
        peak_b = peak_a * config.lr_decay_factor
        half = 0.5 * config.lr_decay_epoch_interval
        mid = interval * config.lr_decay_epoch_interval + half
        slope = -(peak_a - base) / half if epoch < mid else (peak_b - base) / half
        return slope * (epoch - mid) + base
    return target


def lr_used_by_step(config, epoch_len, step):
    """LambdaLR(optimizer, lr_schedule) as driven by training.py:436-442,576-577: initial LR times the function value."""
    return config.learning_rate * lr_schedule(config, epoch_len, step)
