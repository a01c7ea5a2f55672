"""Workers of the data-parallel tests (TEST INFRASTRUCTURE): one optimiser step on `world` ranks, each on its share of a
global clip batch, written to disk for the parent to compare with a single process on the whole batch.

  device 'cpu' : gloo + tests/fake_kernels.py -- the host logic (bucketing, hooks, use counting) in the build container;
  device 'cuda': gloo + the REAL HIP kernels, every rank on GPU 0 (EVE_AMD_FORCE_DEVICE=0) -- the paths that only exist
                 with the kernels: weight gradients written in place into the flat buffer -> _eve_grad_ready -> bucket
                 launch, hipGraph replay with eager collectives.
"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    """A port the OS just handed out on 127.0.0.1 (a fixed one can sit in TIME_WAIT from an earlier run)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _paths():
    for p in (REPO, os.path.join(REPO, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)


def install_kernels(device):
    _paths()
    from eve_amd import kernels
    if device == 'cpu':
        from fake_kernels import FakeKernels
        kernels.set_default_kernels(FakeKernels())
    else:
        kernels.set_default_kernels(None)           # the real library, loaded lazily; raises if it is not built


def build(case, device, dtype, base_lr=None):
    """-> (config, trainer factory(distributed, use_graph), global batch).  case: 'eyenet' | 'eyenet_per_frame' | 'eve'."""
    _paths()
    import eve_amd
    from eve_amd import losses, train
    from oracle import detweights
    dt = {'fp32': torch.float32, 'bf16': torch.bfloat16}[dtype]
    cfg = eve_amd.reset_standalone_config()
    if case.startswith('eyenet'):
        cfg.import_json(os.path.join(REPO, 'configs', 'eye_net.json'))
        if base_lr is not None:
            cfg.override('base_learning_rate', base_lr)
        net = detweights.fill_module(eve_amd.EyeNet())
        net.compute_dtype = dt
        net.to(device)
        size = 64 if device == 'cpu' else 128
        full = detweights.eyenet_batch(2, 2, seed=11, size=size)
        if case == 'eyenet':
            make = lambda distributed, use_graph=False: train.eyenet_trainer(net, cfg, distributed=distributed, use_graph=use_graph)
        else:
            # the reference's per-frame contract (src/models/eve.py:91-111): EyeNet.forward once per time step and eye
            # side, so every trunk weight is used 2 T times per optimiser step
            def loss_fn(batch):
                T = batch['left_eye_patch'].shape[1]
                steps, prev = [], None
                for t in range(T):
                    si = {k: v[:, t] for k, v in batch.items()}
                    so = {}
                    net(si, so, side='left', previous_output_dict=prev)
                    net(si, so, side='right', previous_output_dict=prev)
                    steps.append(so)
                    prev = so
                out = {k: torch.stack([s[k] for s in steps], dim=1) for k in steps[0]}
                return losses.eyenet_loss_terms(out, batch, cfg)
            make = lambda distributed, use_graph=False: train.Trainer([net], cfg, loss_fn, distributed=distributed)
        return cfg, make, full
    cfg.import_json(os.path.join(REPO, 'configs', 'refine_net.json'))
    # (the kappa draw is per process: switch the augmentation off so that 2 x 1 clip and 1 x 2 clips see the same data)
    cfg.import_dict({'refine_net_rnn_type': 'CGRU', 'eye_net_load_pretrained': False, 'refine_net_do_offset_augmentation': False})
    model = eve_amd.EVE()
    detweights.fill_module(model.eye_net, 0)
    detweights.fill_module(model.refine_net, 1)
    model.eye_net.compute_dtype = model.refine_net.compute_dtype = dt
    model.to(device).train()
    make = lambda distributed, use_graph=False: train.eve_trainer(model, cfg, distributed=distributed)
    return cfg, make, detweights.eve_batch(2, 2, seed=13)


def worker(rank, world, port, tmp, case, device, dtype, use_graph, backend='gloo'):
    _paths()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), EVE_AMD_DIST_BACKEND=backend)
    if backend == 'gloo':
        os.environ['EVE_AMD_FORCE_DEVICE'] = '0'          # every rank on GPU 0; RCCL ('nccl') needs a device per rank
    else:
        os.environ.pop('EVE_AMD_FORCE_DEVICE', None)
        os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
        if use_graph == 'captured':
            os.environ['EVE_AMD_GRAPH_COLLECTIVES'] = '1'
        device = 'cuda:%d' % rank
    torch.set_num_threads(2)
    import torch.distributed as dist
    from eve_amd import parallel
    parallel.GradSync.default_bucket_elems = 1000000          # several buckets on these small batches
    install_kernels('cpu' if device == 'cpu' else 'cuda')
    r, _, w = parallel.init_distributed(backend=backend)
    assert (r, w) == (rank, world) and dist.get_backend() == backend
    cfg, make, full = build(case, device, dtype, 1e-7 if use_graph else None)
    tr = make(True, bool(use_graph))
    assert tr.graph_collectives == (use_graph == 'captured')
    assert len(tr.sync.buckets) >= 2
    assert tr.sync.buckets[0]['hi'] == tr.fp.flat.numel() and tr.sync.buckets[-1]['lo'] == 0
    mine = {k: v[rank:rank + 1].to(device) for k, v in full.items()}
    steps = 3 if use_graph else 1            # graph mode: capture (with its restored warm-ups) + replays
    for _ in range(steps):
        terms = tr.step(mine)
    if device != 'cpu':
        torch.cuda.synchronize()
    assert all(c == 1 for c in tr.sync.launch_counts), tr.sync.launch_counts        # every bucket exactly once per step
    assert tr.sync.gate_timeouts() == 0, 'a gate of the replay never opened (the collective ran after the time-out)'
    torch.save({'flat': tr.fp.flat.cpu().clone(), 'grad': tr.fp.grad.cpu().clone(), 'loss': float(terms['full_loss'].detach()),
                'buckets': len(tr.sync.buckets)}, os.path.join(tmp, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def worker_rccl(rank, port, tmp, case, dtype, use_graph, steps):
    """ONE rank on the GPU with backend nccl (= RCCL): the transport the multi-GPU bench uses, on the only topology a
    one-GPU box offers.  Communicator creation, the bucket all-reduces launched from the gradient notifications onto
    the communication stream, their ordering against clip + Adam, and (use_graph) their coexistence with hipGraph replay."""
    _paths()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      EVE_AMD_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    os.environ.pop('EVE_AMD_DIST_BACKEND', None)
    if use_graph == 'captured':                # the bucket all-reduces inside the hipGraph as well
        os.environ['EVE_AMD_GRAPH_COLLECTIVES'] = '1'
    import torch.distributed as dist
    from eve_amd import parallel
    parallel.GradSync.default_bucket_elems = 1000000
    install_kernels('cuda')
    r, _, w = parallel.init_distributed(backend='nccl')
    assert (r, w) == (0, 1) and dist.get_backend() == 'nccl'
    cfg, make, full = build(case, 'cuda', dtype, 1e-7 if use_graph else None)
    tr = make(True, bool(use_graph))
    assert tr.graph_collectives == (use_graph == 'captured')
    batch = {k: v.to('cuda') for k, v in full.items()}
    for _ in range(steps):
        terms = tr.step(batch)
    torch.cuda.synchronize()
    assert all(c == 1 for c in tr.sync.launch_counts), tr.sync.launch_counts
    assert tr.sync.gate_timeouts() == 0, 'a gate of the replay never opened (the collective ran after the time-out)'
    torch.save({'flat': tr.fp.flat.cpu().clone(), 'grad': tr.fp.grad.cpu().clone(), 'loss': float(terms['full_loss'].detach()),
                'buckets': len(tr.sync.buckets)}, os.path.join(tmp, 'rccl.pt'))
    dist.barrier()
    dist.destroy_process_group()


def worker_gate_timeout(rank, port, tmp):
    """One RCCL rank, hipGraph replay + gated collectives, and a bucket whose ready signal comes LATE in one step: the capture
    holds a long sleep in front of one bucket's gate-signal node; while the gates wait with their default bound (seconds) the
    step is a normal one, with the bound cut below the sleep (eve_dispatch_config.gate_wait_polls) the gate gives up, the
    all-reduce runs on the half-written bucket -- and the step must NOT be applied: the gate poisons it, the Adam guard skips."""
    _paths()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      EVE_AMD_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    os.environ.pop('EVE_AMD_DIST_BACKEND', None)
    os.environ.pop('EVE_AMD_GRAPH_COLLECTIVES', None)
    import torch.distributed as dist
    from eve_amd import parallel
    from eve_amd.kernels import default_kernels
    parallel.GradSync.default_bucket_elems = 1000000
    install_kernels('cuda')
    parallel.init_distributed(backend='nccl')
    cfg, make, full = build('eyenet', 'cuda', 'bf16', 1e-3)
    tr = make(True, True)
    k = default_kernels()
    plain_launch = tr.sync._launch
    slept = []

    def launch(b, inline=False):
        # the SECOND bucket whose ready point is captured gets the stall in front of its gate-signal node
        if tr.sync._marking and not b['launched'] and b['hi'] > b['lo'] and len(tr.sync._gated) == 1:
            torch.cuda._sleep(300_000_000)                  # ~0.15 s, every replay
            slept.append(tr.sync.buckets.index(b))
        plain_launch(b, inline=inline)
    tr.sync._launch = launch
    batch = {kk: v.to('cuda') for kk, v in full.items()}
    rec = {}
    import time
    tr.step(batch)                                          # capture + replay 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(batch)                                          # replay 2: default bound, the gates sit the sleep out
    torch.cuda.synchronize()
    rec['after_two'] = dict(tr.optimizer_state(), timeouts=tr.sync.gate_timeouts(), gated=len(tr.sync._gated), slept=list(slept),
                            replay_s=time.perf_counter() - t0, order=[tr.sync.buckets.index(b) for b in tr.sync._gated])
    # diagnostics: does the communication stream run BESIDE the main stream on this box (or behind it: hardware-queue aliasing)?
    main, comm = torch.cuda.current_stream(), tr.sync._comm
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    probe = torch.zeros(64, device='cuda')
    e0.record(main)
    torch.cuda._sleep(100_000_000)
    e1.record(main)
    with torch.cuda.stream(comm):
        probe.add_(1.0)
        e2.record(comm)
    torch.cuda.synchronize()
    rec['probe_ms'] = dict(main_sleep=e0.elapsed_time(e1), comm_kernel_done=e0.elapsed_time(e2), overlaps=tr.sync.overlaps)
    before = tr.fp.flat.clone()
    m_before = tr.fp.m.clone()
    tr.gate_check_every = 1
    with k.dispatch_override(gate_wait_polls=2000):         # a few ms: far above this tiny backward, far below the sleep
        tr.step(batch)
        torch.cuda.synchronize()
    rec['after_late'] = dict(tr.optimizer_state(), timeouts=tr.sync.gate_timeouts(), gated=len(tr.sync._gated), polls=k.dispatch_config().gate_wait_polls,
                             weights_unchanged=bool(torch.equal(tr.fp.flat, before) and torch.equal(tr.fp.m, m_before)),
                             poison=float(tr.fp.poison[0]))
    tr.step(batch)                                          # the trainer has fallen back: collectives behind the replay
    torch.cuda.synchronize()
    rec['after_fallback'] = dict(tr.optimizer_state(), timeouts=tr.sync.gate_timeouts(), gated=len(tr.sync._gated),
                                 weights_moved=bool(not torch.equal(tr.fp.flat, before)), launch_counts=list(tr.sync.launch_counts))
    torch.save(rec, os.path.join(tmp, 'gate.pt'))
    dist.barrier()
    dist.destroy_process_group()


def run_gate_timeout(tmp):
    import torch.multiprocessing as mp
    mp.spawn(worker_gate_timeout, args=(free_port(), tmp), nprocs=1, join=True)
    return torch.load(os.path.join(tmp, 'gate.pt'))


def run_rccl_single_rank(tmp, case, dtype, use_graph):
    import torch.multiprocessing as mp
    steps = 3 if use_graph else 1        # eager: the FIRST step's gradient (later ones inherit Adam's sign-flip chaos)
    mp.spawn(worker_rccl, args=(free_port(), tmp, case, dtype, use_graph, steps), nprocs=1, join=True)
    a = torch.load(os.path.join(tmp, 'rccl.pt'))
    try:
        flat, grad, loss, lr = single_process(case, 'cuda', dtype, steps=steps, base_lr=1e-7 if use_graph else None)
    finally:
        from eve_amd import kernels
        import eve_amd
        kernels.set_default_kernels(None)
        eve_amd.reset_standalone_config()
    return a, flat, grad, loss, lr


def single_process(case, device, dtype, steps=1, base_lr=None):
    install_kernels(device)
    cfg, make, full = build(case, device, dtype, base_lr)
    tr = make(False)
    batch = {k: v.to(device) for k, v in full.items()}
    for _ in range(steps):
        terms = tr.step(batch)
    return tr.fp.flat.cpu().clone(), tr.fp.grad.cpu().clone(), float(terms['full_loss'].detach()), float(cfg.learning_rate)


def run_and_compare(tmp, case, device, dtype, use_graph=False, grad_tol=1e-3, backend='gloo'):
    import torch.multiprocessing as mp
    mp.spawn(worker, args=(2, free_port(), tmp, case, device, dtype, use_graph, backend), nprocs=2, join=True)
    a = torch.load(os.path.join(tmp, 'rank0.pt'))
    b = torch.load(os.path.join(tmp, 'rank1.pt'))
    def where(x, y):
        d = (x != y).nonzero().reshape(-1)
        return '%d of %d elements differ, first at %s, last at %s, max |d| %.3e' % (
            d.numel(), x.numel(), d[:3].tolist(), d[-3:].tolist(), float((x - y).abs().max()))
    assert torch.equal(a['grad'], b['grad']), 'ranks diverged (gradients): ' + where(a['grad'], b['grad'])
    assert torch.equal(a['flat'], b['flat']), 'ranks diverged (parameters): ' + where(a['flat'], b['flat'])
    try:
        # (graph case: three steps with a tiny learning rate -- Adam's first steps move every weight by ~lr * sign(g), and
        #  with the configured lr = 0.016 a handful of sign flips on noise-level gradients would change the NEXT step's
        #  gradients everywhere; the comparison is about launch plumbing, not about chaotic training dynamics)
        flat, grad, loss, lr = single_process(case, device, dtype, steps=3 if use_graph else 1,
                                              base_lr=1e-7 if use_graph else None)
        if not use_graph:
            # summed rank gradients / world == gradient of the mean-over-clips loss on the global batch
            rel = float((a['grad'] / 2 - grad).norm() / grad.norm())
            assert rel < grad_tol, rel
            assert abs(0.5 * (a['loss'] + b['loss']) - loss) < 1e-3 * max(1.0, abs(loss))      # mean of per-clip means
        # Adam's first steps are ~ lr * sign(g): elements whose gradient is round-off noise may flip, so the update is
        # compared in bulk rather than element by element
        d = (a['flat'] - flat).abs()
        frac = float((d > 0.25 * lr).float().mean())
        assert frac < (2e-3 if not use_graph else 1e-2), frac
    finally:
        from eve_amd import kernels
        import eve_amd
        kernels.set_default_kernels(None)
        eve_amd.reset_standalone_config()
    return a
