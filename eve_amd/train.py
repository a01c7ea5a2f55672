"""Train-step harness around the two drop-in modules: what the reference does per optimiser step in
src/core/training.py:452-502 with the model call of src/models/eve.py, restated for whole-clip batches.

  zero grads -> forward (all T steps, both eyes, one pass) -> masked losses -> backward ->
  [data-parallel gradient all-reduce, overlapped] -> global-norm clip (5.0) -> Adam (coupled L2 decay)

Parameters are re-homed into ONE flat float32 buffer (conv weights physically OHWI, exposed to torch
as OIHW-shaped views, so `state_dict()` is unchanged): the wgrad kernel's output layout, the packed
bf16 copies, the gradient all-reduce buckets and the fused Adam kernel all work on flat memory with no
per-tensor launches and no layout permutes in the step.
"""
import torch

from . import losses
from .kernels import default_kernels
from .parallel import GradSync


class FlatParameters(object):
    """Moves every parameter of `modules` into one flat buffer (and its gradient into another)."""

    def __init__(self, modules, device=None):
        params, seen = [], set()
        for m in modules:
            for p in m.parameters():
                if id(p) not in seen and p.requires_grad:
                    seen.add(id(p))
                    params.append(p)
        if not params:
            raise ValueError('no trainable parameters')
        device = device or params[0].device
        total = sum(p.numel() for p in params)
        total_padded = (total + 3) // 4 * 4
        self.flat = torch.zeros(total_padded, dtype=torch.float32, device=device)
        self.grad = torch.zeros(total_padded, dtype=torch.float32, device=device)
        self.m = torch.zeros(total_padded, dtype=torch.float32, device=device)
        self.v = torch.zeros(total_padded, dtype=torch.float32, device=device)
        self.entries = []
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                if p.dim() == 4:
                    O, I, KH, KW = p.shape
                    view = lambda buf: buf[off:off + n].view(O, KH, KW, I).permute(0, 3, 1, 2)   # noqa: E731
                else:
                    shape = tuple(p.shape)
                    view = lambda buf: buf[off:off + n].view(shape)                                # noqa: E731
                pv = view(self.flat)
                pv.copy_(p.detach().to(device))
                p.data = pv
                p.grad = view(self.grad)
                self.entries.append((p, off, n))
                off += n
        self.numel = total
        self.params = params

    def zero_grad(self):
        self.grad.zero_()
        for p, off, n in self.entries:          # keep .grad pointing into the flat buffer
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                if p.dim() == 4:
                    O, I, KH, KW = p.shape
                    p.grad = self.grad[off:off + n].view(O, KH, KW, I).permute(0, 3, 1, 2)
                else:
                    p.grad = self.grad[off:off + n].view(tuple(p.shape))


class Trainer(object):
    """One optimiser step = `step(batch)`; `loss_fn(batch) -> dict with 'full_loss'` runs the forward."""

    def __init__(self, modules, config, loss_fn, distributed=False, device=None):
        self.modules = list(modules)
        self.config = config
        self.loss_fn = loss_fn
        self.fp = FlatParameters(self.modules, device)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=self.fp.flat.device)
        self.sync = GradSync(self.fp.grad, self.fp.entries) if distributed else None
        self.step_count = 0
        self.beta1, self.beta2, self.eps = 0.9, 0.999, 1e-8      # torch.optim.Adam defaults (train.py:49-55)
        for m in self.modules:
            if hasattr(m, 'invalidate_packs'):
                m.invalidate_packs()

    def step(self, batch):
        k = default_kernels()
        cfg = self.config
        self.fp.zero_grad()
        if self.sync is not None:
            self.sync.start_step()
        terms = self.loss_fn(batch)
        terms['full_loss'].backward()
        gscale = self.sync.finish_step() if self.sync is not None else 1.0
        self.step_count += 1
        clip = cfg.do_gradient_clipping and cfg.gradient_clip_by == 'norm'
        if cfg.do_gradient_clipping and not clip:
            self.fp.grad.mul_(gscale).clamp_(-cfg.gradient_clip_amount, cfg.gradient_clip_amount)
            gscale = 1.0
        if clip:
            self.sumsq.zero_()
            k.sumsq(self.fp.grad, self.sumsq)
        k.adam_step(self.fp.flat, self.fp.grad, self.fp.m, self.fp.v, self.sumsq if clip else None,
                    float(cfg.gradient_clip_amount), gscale, float(cfg.learning_rate), self.beta1, self.beta2,
                    self.eps, float(cfg.weight_decay), self.step_count)
        for m in self.modules:
            if hasattr(m, 'invalidate_packs'):
                m.invalidate_packs()
        return terms


def eyenet_trainer(eye_net, config, distributed=False):
    def loss_fn(batch):
        return losses.eyenet_loss_terms(eye_net.forward_sequence(batch), batch, config)
    return Trainer([eye_net], config, loss_fn, distributed=distributed)


def refinenet_trainer(refine_net, config, distributed=False):
    def loss_fn(batch):
        hf, _ = refine_net.forward_sequence(batch['heatmap_initial'], batch.get('screen_frame'))
        return losses.refinenet_loss_terms(hf, batch['heatmap_final_gt'], batch['validity'], config)
    return Trainer([refine_net], config, loss_fn, distributed=distributed)
