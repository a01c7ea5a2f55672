"""Train-step harness around the two drop-in modules: what the reference does per optimiser step in
src/core/training.py:452-502 with the model call of src/models/eve.py, restated for whole-clip batches.

  zero grads -> forward (all T steps, both eyes, one pass) -> masked losses -> backward ->
  [data-parallel gradient all-reduce, overlapped] -> global-norm clip (5.0) -> Adam (coupled L2 decay)

Parameters are re-homed into ONE flat float32 buffer (conv weights physically OHWI, exposed to torch
as OIHW-shaped views, so `state_dict()` is unchanged): the wgrad kernel's output layout, the packed
bf16 copies, the gradient all-reduce buckets and the fused Adam kernel all work on flat memory with no
per-tensor launches and no layout permutes in the step.
"""
import os

import torch

from . import losses
from .kernels import default_kernels
from .parallel import GradSync


class FlatParameters(object):
    """Moves every parameter of `modules` into one flat buffer (and its gradient into another).  The buffers start with LEAD
    pad floats (zero weights with zero gradients): grad[0] is the data-parallel exchange's POISON word -- inside the last
    gradient bucket, written by a stream gate that timed out, read by the Adam guard (parallel.GradSync.launch_gated).
    LEAD is 64 floats = 256 bytes so that the parameters keep the alignment they would have without it: the convolution filters
    are multiples of 64 floats, and the weight-gradient kernels' float atomics / 16-byte slab stores into `grad` go by cache line
    (with the 16-byte lead this class had for half of round 6, every wave's 256 contiguous bytes straddled three 128-byte lines
    instead of two: wgrad_tr_kernel 0.143 instead of 0.121 ms per layer-2 launch, the step -1.6 % at B = 32 and -5.5 % at B = 8;
    profiles/r06_notes.md section 10)."""

    LEAD = 64

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
        total = self.LEAD + sum(p.numel() for p in params)
        total_padded = (total + 3) // 4 * 4
        self.flat = torch.zeros(total_padded, dtype=torch.float32, device=device)
        self.grad = torch.zeros(total_padded, dtype=torch.float32, device=device)
        self.m = torch.zeros(total_padded, dtype=torch.float32, device=device)
        self.v = torch.zeros(total_padded, dtype=torch.float32, device=device)
        self.entries = []
        off = self.LEAD
        self.poison = self.grad[0:1]
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
                p._eve_flat_grad = True            # ops.Conv2dFn may accumulate wgrad / bias-grad in place
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
    """One optimiser step = `step(batch)`; `loss_fn(batch) -> dict with 'full_loss'` runs the forward.

    use_graph=True captures the step into a hipGraph (torch.cuda.CUDAGraph over the stream our kernels are
    launched on) after two eager warm-up steps and replays it afterwards: the step is ~600 launches, a third of
    them tiny tail/loss kernels whose host-side enqueue would otherwise leave the GPU idle at the start of every
    backward.  Shapes are static (B, T fixed), so a new batch is copied into the captured input buffers.
    With data parallelism the graph holds forward+backward only; every bucket's ready point is captured as a gate-signal
    node and the bucket's all-reduce is issued eagerly on a communication stream behind a gate-wait kernel for that node
    of the running replay, so the collectives overlap the rest of the backward (parallel.GradSync.launch_gated); clip and
    Adam follow.  graph_collectives captures the all-reduces themselves instead (RCCL only)."""

    def __init__(self, modules, config, loss_fn, distributed=False, device=None, use_graph=False, lr_schedule=None,
                 loss_scale=None, graph_collectives=None):
        """lr_schedule: None (constant config.learning_rate) or a callable step -> learning rate, e.g.
        `lambda s: schedule.effective_learning_rate(config, steps_per_epoch, s)` for the reference's behaviour
        (src/core/training.py:382-418,436-442).  The value lives in a device scalar the Adam kernel reads, so it also
        changes under hipGraph replay.
        loss_scale: static factor on the loss before backward, divided out again inside the Adam kernel's gradient scale
        (clip norm included).  None = 1024 when a module computes in float16 (activation gradients of this network reach
        1e-6..1e-8 per element, below float16's normal range; the reference is float32 and has no such step), else 1."""
        self.modules = list(modules)
        self.config = config
        self.loss_fn = loss_fn
        if loss_scale is None:
            loss_scale = 1024.0 if any(getattr(m, 'compute_dtype', None) == torch.float16 for m in self.modules) else 1.0
        self.loss_scale = float(loss_scale)
        self.fp = FlatParameters(self.modules, device)
        dev = self.fp.flat.device
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.sumsq_ws = torch.zeros(1024, dtype=torch.float32, device=dev)     # scratch of the fixed-order reduction
        # device-resident optimiser state (include/eve_hip.h eve_adam_guard): the count of steps actually TAKEN (Adam's
        # bias-correction exponent -- a step the kernel skips for a non-finite float16 gradient does not advance it), the skip
        # counters and the loss scale, which backs off after two consecutive skips.  The overflow check is on whenever a loss
        # scale is in play (float16), whatever the clipping mode; bf16 / fp32 runs keep failing visibly on NaN.
        self.check_overflow = self.loss_scale != 1.0 or any(getattr(m, 'compute_dtype', None) == torch.float16 for m in self.modules)
        self.guard = default_kernels().new_adam_guard(dev, loss_scale=self.loss_scale)
        self.loss_scale_dev = self.guard.view(torch.float32)[4:5]
        self.lr_schedule = lr_schedule
        self.lr = float(config.learning_rate)
        self.lr_dev = torch.full((1,), self.lr, dtype=torch.float32, device=dev)
        self.sync = GradSync(self.fp.grad, self.fp.entries, poison=self.fp.poison) if distributed else None
        self.step_count = 0
        self.beta1, self.beta2, self.eps = 0.9, 0.999, 1e-8      # torch.optim.Adam defaults (train.py:49-55)
        self.use_graph = bool(use_graph)
        # with several ranks the captured graph may also hold the bucket all-reduces (RCCL supports stream capture; gloo
        # does not): EVE_AMD_GRAPH_COLLECTIVES=1 or graph_collectives=True.  Off by default -- replay then runs forward +
        # backward and the collectives / clip / Adam follow eagerly -- until it has run on a multi-GPU node.
        import torch.distributed as dist
        if graph_collectives is None:
            graph_collectives = os.environ.get('EVE_AMD_GRAPH_COLLECTIVES', '0') == '1'
        self.graph_collectives = bool(graph_collectives) and self.sync is not None and dist.is_initialized() and \
            dist.get_backend() == 'nccl'
        self._graph = None
        self._graph_bwd = None
        # gated mode: replay the forward and the backward as two graphs with an event between them (see _capture); the
        # EVE_AMD_SPLIT_REPLAY=0 escape keeps the one-graph form measurable
        self.split_replay = os.environ.get('EVE_AMD_SPLIT_REPLAY', '1') == '1'
        self._gate_checks = 2         # replays after a capture whose gate outcome is read back (a host sync each) ...
        self.gate_check_every = 64    # ... and then every this many steps
        self._gate_skips_seen = 0
        self._static_batch = None
        self._static_terms = None
        self.pre_step = None          # optional host-side hook run at the top of every step(batch), before capture / replay
        self.post_step = None         # ... and after the step's work has been enqueued
        self._invalidate()

    def _invalidate(self):
        for m in self.modules:
            if hasattr(m, 'invalidate_packs'):
                m.invalidate_packs()

    # ---- the two halves of a step ----
    def _forward(self, batch):
        self.fp.zero_grad()
        return self.loss_fn(batch)

    def _backward(self, terms):
        loss = terms['full_loss']
        # The root gradient is a resident tensor instead of autograd's ones_like (a fill launch per step) -- and under a loss
        # scale it IS the scale, read from the device (it may have backed off, and a captured graph must follow it): the same
        # d(loss * scale) without the multiply launch.
        if self.check_overflow and loss.dim() == 0 and loss.dtype == torch.float32:
            loss.backward(gradient=self.loss_scale_dev[0])
        elif self.check_overflow:
            (loss * self.loss_scale_dev[0]).backward()
        elif loss.dim() == 0 and loss.dtype == torch.float32:
            if getattr(self, '_grad_one', None) is None or self._grad_one.device != loss.device:
                self._grad_one = torch.ones((), dtype=torch.float32, device=loss.device)
            loss.backward(gradient=self._grad_one)
        else:
            loss.backward()

    def _forward_backward(self, batch):
        terms = self._forward(batch)
        self._backward(terms)
        return terms

    def _update(self, gscale):
        k = default_kernels()
        cfg = self.config
        by_norm = cfg.do_gradient_clipping and cfg.gradient_clip_by == 'norm'
        if cfg.do_gradient_clipping and not by_norm:
            if self.check_overflow:       # (a clamp would turn an overflowed gradient into a finite, wrong one)
                raise NotImplementedError('clip-by-value together with a loss scale (float16) is not supported: clip by norm')
            self.fp.grad.mul_(gscale).clamp_(-cfg.gradient_clip_amount, cfg.gradient_clip_amount)
            gscale = 1.0
        need_norm = by_norm or self.check_overflow
        if need_norm:
            self.sumsq.zero_()
            k.sumsq(self.fp.grad, self.sumsq, self.sumsq_ws)
        k.adam_step(self.fp.flat, self.fp.grad, self.fp.m, self.fp.v, self.sumsq if need_norm else None,
                    float(cfg.gradient_clip_amount) if by_norm else 0.0, gscale, self.lr, self.beta1, self.beta2,
                    self.eps, float(cfg.weight_decay), 0, guard=self.guard, check_finite=self.check_overflow, lr_dev=self.lr_dev,
                    poison=self.fp.poison if self.sync is not None and self.fp.poison.is_cuda else None)
        self._invalidate()

    def optimizer_state(self):
        """Host view of the device-resident optimiser state (synchronises): steps taken, steps skipped, loss scale."""
        g = self.guard.cpu()
        return {'steps_taken': int(g[0]), 'steps_skipped': int(g[1]), 'loss_scale': float(g.view(torch.float32)[4]),
                'steps_skipped_gate_timeout': int(g[9])}

    def set_lr(self, lr):
        """Learning rate of the next step(s): written to the device scalar the (possibly graph-captured) Adam kernel reads."""
        lr = float(lr)
        if lr != self.lr:
            self.lr = lr
            self.lr_dev.fill_(lr)

    def _apply_schedule(self):
        if self.lr_schedule is not None:
            self.set_lr(self.lr_schedule(self.step_count))

    def _eager_step(self, batch):
        if self.sync is not None:
            self.sync.start_step()
        terms = self._forward_backward(batch)
        gscale = self.sync.finish_step() if self.sync is not None else 1.0
        self._update(gscale)
        return terms

    def _snapshot(self):
        return [t.clone() for t in (self.fp.flat, self.fp.m, self.fp.v, self.guard)]

    def _restore(self, snap):
        for t, s_ in zip((self.fp.flat, self.fp.m, self.fp.v, self.guard), snap):
            t.copy_(s_)
        self._invalidate()

    def _capture(self, batch):
        # the graph reads its inputs from fixed buffers: copies of the first batch, or (static_inputs='alias') the first
        # batch's own tensors -- for a caller that refills the same device buffers every step (a prefetcher writing in
        # place, or a resident synthetic batch) the per-step device-to-device copy of the clips (378 MB at B=32) is then gone
        alias = getattr(self, 'static_inputs', 'copy') == 'alias'
        self._static_batch = {k: (v if alias else v.clone()) if isinstance(v, torch.Tensor) else v for k, v in batch.items()}
        # ONE warm-up stream per device for the life of the process: the kernels' scratch is keyed by stream and never freed, so a
        # fresh stream per capture (several trainers in one process, re-captures) would pin another workspace each time
        dev_index = self.fp.flat.device.index if self.fp.flat.device.index is not None else torch.cuda.current_device()
        side = _WARMUP_STREAMS.get(dev_index)
        if side is None:
            side = _WARMUP_STREAMS[dev_index] = torch.cuda.Stream(device=dev_index)
        side.wait_stream(torch.cuda.current_stream())
        self._gate_checks = 2
        # two eager warm-up passes (allocator, lazy initialisation) off the default stream, as capture requires; they
        # must not train: parameters, moments and the step counter are put back afterwards, so the first replay IS the
        # first optimiser step (one update on the first batch, exactly like the eager path and the reference).
        # The snapshot is taken ON the side stream, after its wait: clone, warm-up and restore are ordered on one stream.
        with torch.cuda.stream(side):
            snap = self._snapshot()
            for _ in range(2):
                if self.sync is not None and not self.graph_collectives:
                    self._forward_backward(self._static_batch)
                    self._collective_and_update()
                else:
                    self._eager_step(self._static_batch)      # (with collectives: also brings the communicator up before capture)
            self._restore(snap)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        k = default_kernels()
        if hasattr(k, 'prepare_graph_workspace'):
            k.prepare_graph_workspace(self.fp.flat.device)     # the captured launches' scratch: from the ordinary pool, not the graph's
        if self.sync is not None and not self.graph_collectives:
            self.sync.prepare_marks()                          # gate words: allocated and zeroed OUTSIDE the capture
            torch.cuda.synchronize()
        # with a process group alive, its watchdog THREAD polls the events of the warm-up collectives (hipEventQuery): under the
        # default 'global' capture mode any such call from another thread while this one captures is an error that takes the
        # process down -- seen as a rare crash of the one-rank RCCL tests (c10d::ProcessGroupNCCL::Watchdog, HIPEvent query).
        # 'thread_local' confines the capture rules to the capturing thread.
        mode = 'thread_local' if self.sync is not None else 'global'
        with torch.cuda.graph(self._graph, capture_error_mode=mode):
            if self.sync is not None and self.graph_collectives:
                # the whole distributed step in the graph: every bucket's all-reduce is captured where its gradient
                # notification fires (RCCL work on its own stream, joined back before the clip), then clip + Adam
                self.sync.start_step()
                self._static_terms = self._forward_backward(self._static_batch)
                self._update(self.sync.finish_step())
            elif self.sync is not None:
                # forward + backward captured, every bucket's ready point as a gate-signal node (csrc/optim.hip).  The FORWARD
                # is a graph of its own (split_replay): an ordinary event between the two replays keeps the communication stream's
                # first gate-wait wave off the device until the backward begins -- any wave resident beside the replay costs
                # stem_fwd_pairs_kernel +40 % and the layer-1 forward convolutions +9 % (profiles/r06_notes.md section 9), and
                # no backward kernel anything
                self.sync.begin_marks()
                self._static_terms = self._forward(self._static_batch)
                if not self.split_replay:
                    self._backward(self._static_terms)
                    self.sync.end_marks()
            else:
                self._static_terms = self._forward_backward(self._static_batch)
                self._update(1.0)

        self._graph_bwd = None
        if self.sync is not None and not self.graph_collectives and self.split_replay:
            self._graph_bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph_bwd, pool=self._graph.pool(), capture_error_mode=mode):
                self._backward(self._static_terms)
                self.sync.end_marks()

    def _collective_and_update(self):
        self.sync.start_step()
        self._update(self.sync.finish_step())

    def _gated_collective_and_update(self):
        # (clip + Adam as a second small graph behind the collectives was measured: no gain over the four eager launches)
        self._update(self.sync.launch_gated())

    def step(self, batch):
        """One optimiser step.  Returns the loss terms; under use_graph these are the graph's static output tensors
        (overwritten by the next step: clone what must be kept)."""
        self._apply_schedule()
        if self.pre_step is not None:
            self.pre_step(batch)
        try:
            if not self.use_graph:
                self.step_count += 1
                return self._eager_step(batch)
            if self._graph is None:
                self._capture(batch)
            else:
                for k, v in batch.items():
                    if isinstance(v, torch.Tensor) and v.data_ptr() != self._static_batch[k].data_ptr():
                        self._static_batch[k].copy_(v, non_blocking=True)
            self._graph.replay()
            if self._graph_bwd is not None:
                self.sync.forward_done()                 # (an event on this stream: the communication stream starts behind it)
                self._graph_bwd.replay()
            if self.sync is not None and not self.graph_collectives:
                self._gated_collective_and_update()
                if self._gate_checks > 0 or (self.gate_check_every and (self.step_count + 1) % self.gate_check_every == 0):
                    # A gate that timed out has poisoned its step on the device: the Adam guard skipped it on EVERY rank (the
                    # poison word rides in the last bucket's all-reduce), nothing wrong was applied.  What is left for the host
                    # is the policy: the first replays after a capture and every gate_check_every-th step read the guard's count
                    # (a host sync); it is the same number on all ranks, so they all fall back to "collectives behind the whole
                    # replay" at the same step -- e.g. on a box whose communication stream shares a hardware queue with the replay.
                    self._gate_checks = max(0, self._gate_checks - 1)
                    skips = int(self.guard[9].item())
                    if skips > self._gate_skips_seen:
                        import sys
                        print('eve_amd: %d step(s) skipped because a gradient-bucket gate timed out; the all-reduces are issued '
                              'behind the whole replay from now on (no overlap with backward)' % (skips - self._gate_skips_seen),
                              file=sys.stderr)
                        self._gate_skips_seen = skips
                        self.sync.disable_gating()
            self.step_count += 1
            return self._static_terms
        finally:
            if self.post_step is not None:
                self.post_step(batch)


_WARMUP_STREAMS = {}      # device index -> the side stream every Trainer's capture warm-up runs on


def _resolve_schedule(config, lr_schedule, steps_per_epoch):
    """The factories' learning-rate argument.  `lr_schedule` (callable step -> LR) wins; else `steps_per_epoch` selects the
    REFERENCE's behaviour -- warm-up / decay per src/core/training.py:382-418 multiplied by the optimiser's initial LR
    as torch's LambdaLR does (:436-442), i.e. schedule.effective_learning_rate; with neither the LR is the constant
    config.learning_rate (what bench.py and the parity tests step with: one schedule-free Adam update per call)."""
    if lr_schedule is not None or steps_per_epoch is None:
        return lr_schedule
    from . import schedule
    return lambda s: schedule.effective_learning_rate(config, steps_per_epoch, s)


def eyenet_trainer(eye_net, config, distributed=False, use_graph=False, lr_schedule=None, steps_per_epoch=None):
    def loss_fn(batch):
        # = losses.eyenet_loss_terms(eye_net.forward_sequence(batch), batch, config), tail + losses as one node when it applies
        return eye_net.loss_terms_sequence(batch, config)
    return Trainer([eye_net], config, loss_fn, distributed=distributed, use_graph=use_graph,
                   lr_schedule=_resolve_schedule(config, lr_schedule, steps_per_epoch))


def refinenet_trainer(refine_net, config, distributed=False, use_graph=False, lr_schedule=None, steps_per_epoch=None):
    def loss_fn(batch):
        hf, _ = refine_net.forward_sequence(batch['heatmap_initial'], batch.get('screen_frame'))
        return losses.refinenet_loss_terms(hf, batch['heatmap_final_gt'], batch['validity'], config)
    return Trainer([refine_net], config, loss_fn, distributed=distributed, use_graph=use_graph,
                   lr_schedule=_resolve_schedule(config, lr_schedule, steps_per_epoch))


def eve_trainer(model, config, distributed=False, current_epoch=0.0, lr_schedule=None, steps_per_epoch=None, use_graph=False):
    """Train step of the whole EVE harness (eve.EVE): forward through both networks and the geometry / heat-map /
    soft-argmax glue, every loss of eve.py:234-265, backward, clip, Adam on whichever network is trainable
    (refine_net.json freezes EyeNet).
    use_graph: the ~1 200 launches of the step replay as one hipGraph.  The reference draws the offset augmentation's
    kappa_fake on the host inside forward (eve.py:463-479); here the SAME draw (numpy RNG, one per step, same order) is made
    by a pre-step hook and copied into fixed device buffers the captured kernels read (eve.EVE.refresh_static_kappa), so
    eager and replayed steps see identical augmentations."""
    modules = [m for m in (model.eye_net, model.refine_net)
               if m is not None and any(p.requires_grad for p in m.parameters())]

    def loss_fn(batch):
        return model({'train': dict(batch)}, current_epoch=current_epoch)
    tr = Trainer(modules, config, loss_fn, distributed=distributed, use_graph=use_graph,
                 lr_schedule=_resolve_schedule(config, lr_schedule, steps_per_epoch))
    if use_graph:
        def pre_step(batch):
            t = batch['left_eye_patch']
            model.refresh_static_kappa(t.shape[0], t.shape[1], t.device)
            model._static_kappa_active = True        # (only inside Trainer.step: an eager call on the model draws afresh)

        def post_step(batch):
            model._static_kappa_active = False
        tr.pre_step, tr.post_step = pre_step, post_step
    return tr
