"""Data parallelism for the clip batch: one process per GPU, gradients summed with RCCL
(torch.distributed backend "nccl" on ROCm) over xGMI, bucketed and overlapped with backward.

The reference is single-device (cuda:0 everywhere); clips are independent units (InstanceNorm is
per-sample, losses are per-clip means), so the only exchange per step is the gradient all-reduce.
Gradients live in ONE flat float buffer (eve_amd/train.py FlatParameters); buckets are contiguous
slices of it, filled back-to-front because backward produces the last layers' gradients first
(layer4 is ~74 % of EyeNet's bytes).  Each bucket's all-reduce is issued from a post-accumulate hook
the moment its last gradient lands, so it rides under the remaining backward kernels; the 1/world
scale is folded into the optimiser kernel's gradient scale instead of a separate pass.
xGMI is point-to-point and ring collectives are per-link bound, so buckets are few and large.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 and os.environ.get('EVE_AMD_FORCE_DIST', '0') != '1':      # (forced: a one-rank group, to exercise the transport)
        return 0, 0, 1
    rank = int(os.environ['RANK'])
    local_rank = int(os.environ.get('LOCAL_RANK', rank))
    if backend is None:
        backend = os.environ.get('EVE_AMD_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if torch.cuda.is_available():
        # EVE_AMD_FORCE_DEVICE lets a 1-GPU box exercise the multi-rank code path (with gloo) on one device
        torch.cuda.set_device(int(os.environ.get('EVE_AMD_FORCE_DEVICE', local_rank)))
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class GradSync(object):
    """Bucketed, backward-overlapped all-reduce(sum) of a flat gradient buffer."""

    default_bucket_elems = None      # None: eve_dispatch_config.bucket_elems (4 Mi floats); tests shrink it on the class

    def __init__(self, flat_grad, entries, bucket_elems=None, group=None, poison=None):
        """entries: list of (param, offset, numel) in flat order (forward order of the network).
        poison: a one-element float view INSIDE flat_grad below the first entry (train.FlatParameters keeps 64 leading pad
        floats) -- it travels with the last bucket's all-reduce; a gate that times out writes +inf there (launch_gated)."""
        self.flat_grad = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if bucket_elems is None:
            bucket_elems = self.default_bucket_elems
        if bucket_elems is None:
            from .kernels import default_kernels, dispatch_flag
            bucket_elems = dispatch_flag(default_kernels(), 'bucket_elems', 4 * 1024 * 1024)
        self.poison = poison
        # walk back-to-front, closing a bucket once it holds >= bucket_elems
        self.buckets = []            # dicts: lo, hi, params, pending
        hi = flat_grad.numel()
        cur = {'lo': hi, 'hi': hi, 'params': []}
        for p, off, n in reversed(entries):
            cur['lo'] = off
            cur['params'].append(p)
            if cur['hi'] - cur['lo'] >= bucket_elems:
                self.buckets.append(cur)
                cur = {'lo': off, 'hi': off, 'params': []}
        if cur['params']:
            cur['lo'] = 0
            self.buckets.append(cur)
        elif self.buckets:
            self.buckets[-1]['lo'] = 0
        self._bucket_of = {}
        for b in self.buckets:
            for p in b['params']:
                self._bucket_of[id(p)] = b
        self._handles = []
        self._hooks = []
        self._armed = False
        self._marking = False        # hipGraph capture: a bucket's "ready" point becomes an event-record node (begin_marks)
        self._gated = []             # buckets in the order their ready points were captured
        self._comm = None            # the stream the gated all-reduces are issued from
        self._flags = None           # one gate word per bucket (device int32), incremented once per replay by the bucket's node
        self._timeouts = None
        self._replays = 0
        self._fwd_event = None       # recorded between the forward graph and the backward graph of a split replay (forward_done)
        self._fwd_recorded = False
        self.overlaps = None         # does the communication stream run beside the replay's stream on this box (prepare_marks)
        self.launch_counts = []
        # a one-rank group has nothing to exchange; EVE_AMD_FORCE_DIST=1 runs the collectives anyway (transport test)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get('EVE_AMD_FORCE_DIST', '0') == '1')
        if self.active:
            for p, _, _ in entries:
                if p.requires_grad:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
                    p._eve_grad_ready = self._on_grad      # gradients written in place never reach AccumulateGrad

    def start_step(self):
        for b in self.buckets:
            b['pending'] = sum(1 for p in b['params'] if p.requires_grad)
            b['launched'] = False
            for p in b['params']:
                p._eve_pending_uses = 0            # ops._note_use counts this step's uses from zero
        self._handles = []
        self._seen = set()
        self._armed = True
        self.launch_counts = [0] * len(self.buckets)     # all-reduces issued per bucket this step (tests assert 1 each)

    # ---- hipGraph replay with the collectives OUTSIDE the graph, still overlapped with backward ----------------------
    # A captured forward + backward cannot call RCCL eagerly from its gradient notifications (they fire at capture time
    # only), and torch on ROCm refuses external event-record nodes ("External events are disallowed in rocm").  So every
    # bucket's ready point is captured as a GATE SIGNAL: a one-thread kernel node behind the weight-gradient kernel that
    # completes the bucket, which releases and increments a device word (csrc/optim.hip: eve_gate_signal).  After each
    # replay is enqueued, launch_gated() walks the buckets in captured order: the communication stream gets a one-wave
    # gate-wait kernel for the bucket's word of THIS replay (value = replay count; bounded poll), then the all-reduce is
    # issued eagerly behind it -- it starts as soon as the replay passes the bucket's last gradient, under the rest of
    # the backward, exactly like the eager mode's notifications.  The RCCL calls themselves stay ordinary eager calls.
    # The gate-wait kernels SPIN (bounded), which has two consequences, both measured in round 5 on a one-rank RCCL group:
    #  * the communication stream must be a NORMAL-priority stream.  As a high-priority stream (to keep it off the replay's
    #    hardware queue) its resident gate kernel made the hardware throttle the normal-priority replay for as long as it
    #    spun: 4.3 -> 10.7 ms per step at B = 8, 11.3 -> 18.0 at B = 32;
    #  * the JOIN stays outside the graph.  Gate-wait nodes INSIDE the replay, waiting for "bucket reduced" words the
    #    communication stream would signal (with clip + Adam captured behind them), ran into their time-outs: RCCL's own
    #    stream shared the replay's hardware queue on the test box, so the spinning node blocked the collective it waited
    #    for.  The same aliasing could in principle hit the communication stream itself on another box, and a gate can also
    #    run into its bound later in a run (a pre-empted queue, a stalled peer).  A gate that gives up therefore POISONS the
    #    step on the device: it writes +inf into `poison`, a pad float at the head of the flat gradient buffer that the LAST
    #    bucket's all-reduce carries, so that after the exchange every rank holds a non-zero value there and the Adam guard
    #    skips the update on every rank alike (eve_adam_step(.., poison); counted in eve_adam_guard.skipped_gate) -- the
    #    all-reduce that ran on a half-written bucket is never applied.  The trainer reads that counter after its first
    #    replays and every `gate_check_every` steps; it is the same number on every rank, so all ranks fall back to
    #    "collectives behind the replay" (disable_gating) at the same step.
    def prepare_marks(self):
        """Allocate the gate words.  Call BEFORE the capture begins: an allocation inside the capture would come from the graph's
        private pool and its zero-fill would become a node of the graph -- every replay would then reset the words it is
        supposed to count up (seen as every gate timing out from the second replay on)."""
        if self.flat_grad.is_cuda:
            if self._comm is None:
                self._comm = self._pick_concurrent_stream(self.flat_grad.device)      # NORMAL priority (see above)
            self._flags = torch.zeros((len(self.buckets) + 1,), dtype=torch.int32, device=self.flat_grad.device)
            self._timeouts = torch.zeros((1,), dtype=torch.int32, device=self.flat_grad.device)
            self._replays = 0

    def _pick_concurrent_stream(self, device, candidates=8):
        """A stream whose kernels run BESIDE the current stream's.  HIP multiplexes streams onto a few hardware queues (4 by
        default, GPU_MAX_HW_QUEUES), handed out in creation order: a communication stream that lands on the replay's queue runs
        its gate-wait kernels only after the whole replay -- correct, but every all-reduce is then exposed (measured in round 6 on
        the one-rank RCCL test process: a kernel on the communication stream finished 6 us after a 42 ms spin on the main stream).
        So each candidate is tried with the gate kernels themselves: the current stream waits (bounded, ~10 ms) for a word only the
        candidate's signal kernel can set; a wait that opens proves the two streams run concurrently.  None does: the last
        candidate is used anyway (`overlaps` says False; the collectives then follow the replay)."""
        from .kernels import default_kernels
        k = default_kernels()
        self.overlaps = False
        if not hasattr(k, 'gate_wait'):
            return torch.cuda.Stream(device=device)
        words = torch.zeros((candidates + 1,), dtype=torch.int32, device=device)
        cand = None
        for i in range(candidates):
            cand = torch.cuda.Stream(device=device)
            torch.cuda.synchronize(device)
            before = int(words[candidates].item())
            k.gate_wait(words, i, 1, words[candidates:candidates + 1], max_polls=5000)       # on the current stream
            with torch.cuda.stream(cand):
                k.gate_signal(words, i)
            torch.cuda.synchronize(device)
            if int(words[candidates].item()) == before:       # the wait opened: the candidate ran while the current stream was busy
                self.overlaps = True
                break
        return cand

    def begin_marks(self):
        assert self._flags is not None or not self.flat_grad.is_cuda, 'prepare_marks() before the capture'
        self.start_step()
        self._marking = True
        self._gated = []

    def forward_done(self):
        """The trainer replays the forward and the backward as two graphs: called between them, on the replay's stream."""
        if self.flat_grad.is_cuda:
            if self._fwd_event is None:
                self._fwd_event = torch.cuda.Event()
            self._fwd_event.record()
            self._fwd_recorded = True

    def end_marks(self):
        self._marking = False
        self._armed = False
        if self.flat_grad.is_cuda and self.overlaps is False:
            # no stream of this process runs beside the replay's (prepare_marks tried): every gate would only open after the whole
            # replay anyway -- issue the collectives behind it without gates
            self._gated = []

    def launch_gated(self):
        """Issue this step's all-reduces behind a replay whose ready points were captured by begin_marks(); buckets that
        never reported (parameters without a gradient) follow the whole replay.  Returns like finish_step()."""
        self.start_step()
        self._armed = False
        if self.active:
            from .kernels import default_kernels
            k = default_kernels()
            main = torch.cuda.current_stream()
            self._replays += 1
            # the bucket that carries the poison word (lo == 0: the last one) goes last, behind every gate that could poison
            last = self.buckets[-1]
            order = [b for b in self._gated if b is not last] + [b for b in self._gated if b is last]
            if self._fwd_recorded:
                self._comm.wait_event(self._fwd_event)        # no gate-wait wave on the device while the forward runs
                self._fwd_recorded = False
            with torch.cuda.stream(self._comm):
                for b in order:
                    if self._flags is not None:
                        k.gate_wait(self._flags, self.buckets.index(b), self._replays, self._timeouts, poison=self.poison)
                    self._launch(b, inline=True)
            rest = [b for b in self.buckets if not b['launched'] and b['hi'] > b['lo']]
            rest = [b for b in rest if b is not last] + [b for b in rest if b is last]
            if rest:
                self._comm.wait_stream(main)
                with torch.cuda.stream(self._comm):
                    for b in rest:
                        self._launch(b, inline=True)
            with torch.cuda.stream(self._comm):
                for h in self._handles:
                    h.wait()                          # the communication stream waits for every collective ...
            main.wait_stream(self._comm)              # ... and the main stream once for it
        self._handles = []
        return 1.0 / self.world

    def disable_gating(self):
        """Fall back to "collectives behind the whole replay": every bucket is launched once the replay has finished (the mode
        of round 4), in self.buckets order -- the same on every rank.  The trainer does this when the Adam guard has counted a
        poisoned step (a gate ran into its time-out: e.g. the communication stream shares a hardware queue with the replay on
        this box, so its spinning gate-wait blocked the very work it waits for); that count is identical on all ranks, so they
        switch at the same step."""
        self._gated = []

    def gate_timeouts(self):
        """Gates that gave up waiting since begin_marks() (host sync; 0 in a healthy run)."""
        return 0 if self._timeouts is None else int(self._timeouts.item())

    def _launch(self, b, inline=False):
        if b['launched'] or b['hi'] <= b['lo']:
            return
        if self._marking:
            # capture: the point itself (a gate signal node), not the collective
            b['launched'] = True
            if self._flags is not None:
                from .kernels import default_kernels
                default_kernels().gate_signal(self._flags, self.buckets.index(b))
            self._gated.append(b)
            return
        b['launched'] = True
        self.launch_counts[self.buckets.index(b)] += 1
        if inline:
            # a synchronous c10d call: the collective is enqueued on the CURRENT stream (the communication stream here) instead
            # of the process group's own stream -- two cross-queue hand-overs fewer per bucket
            dist.all_reduce(self.flat_grad[b['lo']:b['hi']], op=dist.ReduceOp.SUM, group=self.group, async_op=False)
            return
        self._handles.append(dist.all_reduce(self.flat_grad[b['lo']:b['hi']], op=dist.ReduceOp.SUM,
                                             group=self.group, async_op=True))

    def _on_grad(self, p):
        # called from the post-accumulate hook and/or from ops.Conv2dFn for gradients written in place; a
        # parameter counts once per step whichever route reports it (some torch versions run the hook even when
        # backward returned no gradient for the parameter)
        if not self._armed or id(p) in self._seen:
            return
        self._seen.add(id(p))
        b = self._bucket_of[id(p)]
        b['pending'] -= 1
        if b['pending'] == 0:
            self._launch(b)

    def finish_step(self):
        """Launch whatever never completed (parameters without a gradient this step) and wait.
        Returns the factor the optimiser must scale the summed gradient by (1 / world)."""
        self._armed = False
        if self.active:
            for b in self.buckets:
                self._launch(b)
            for h in self._handles:
                h.wait()
        self._handles = []
        return 1.0 / self.world

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
