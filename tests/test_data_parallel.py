"""CPU, world_size 2 over gloo: the bucketed gradient all-reduce of eve_amd.parallel / the flat
trainer gives the same update as one process on the concatenated batch (clips are independent units,
losses are per-clip means, equal local batches)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A port the OS just handed out on 127.0.0.1 (a fixed one can sit in TIME_WAIT from an earlier run and stall the
    rendezvous for minutes)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), EVE_AMD_BUCKET_ELEMS='2000000')
    torch.set_num_threads(2)
    import eve_amd
    from eve_amd import kernels, parallel, train
    from fake_kernels import FakeKernels
    from oracle import detweights
    kernels.set_default_kernels(FakeKernels())
    r, lr, w = parallel.init_distributed(backend='gloo')
    assert (r, w) == (rank, world)
    cfg = eve_amd.reset_standalone_config()
    cfg.import_json(os.path.join(REPO, 'configs', 'eye_net.json'))
    net = detweights.fill_module(eve_amd.EyeNet())
    tr = train.eyenet_trainer(net, cfg, distributed=True)
    assert len(tr.sync.buckets) >= 3
    assert tr.sync.buckets[0]['hi'] == tr.fp.flat.numel() and tr.sync.buckets[-1]['lo'] == 0
    full = detweights.eyenet_batch(2, 2, seed=11, size=64)
    mine = {k: v[rank:rank + 1] for k, v in full.items()}
    tr.step(mine)
    torch.save({'flat': tr.fp.flat.clone(), 'grad': tr.fp.grad.clone()}, os.path.join(tmp, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_on_the_global_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(str(tmp_path), 'rank0.pt'))
    b = torch.load(os.path.join(str(tmp_path), 'rank1.pt'))
    assert torch.equal(a['flat'], b['flat']), 'ranks diverged'
    assert torch.equal(a['grad'], b['grad'])
    # single process, global batch of 2 clips
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import eve_amd
    from eve_amd import kernels, train
    from fake_kernels import FakeKernels
    from oracle import detweights
    kernels.set_default_kernels(FakeKernels())
    try:
        cfg = eve_amd.reset_standalone_config()
        cfg.import_json(os.path.join(REPO, 'configs', 'eye_net.json'))
        net = detweights.fill_module(eve_amd.EyeNet())
        tr = train.eyenet_trainer(net, cfg, distributed=False)
        tr.step(detweights.eyenet_batch(2, 2, seed=11, size=64))
        # summed rank gradients / world == gradient of the mean-over-clips loss on the global batch
        g_dp = a['grad'] / 2
        rel = float((g_dp - tr.fp.grad).norm() / tr.fp.grad.norm())
        assert rel < 1e-3, rel
        # Adam's first step is ~ lr * sign(g): elements whose gradient is round-off noise may flip, so the
        # update is compared in bulk rather than element by element
        d = (a['flat'] - tr.fp.flat).abs()
        assert float((d > 1e-4).float().mean()) < 2e-3, float((d > 1e-4).float().mean())
    finally:
        kernels.set_default_kernels(None)


# ---- BASELINE configs[3]: the whole EyeNet + RefineNet pipeline (eve_amd.EVE) data-parallel -------------------------
def _eve_setup():
    import eve_amd
    from oracle import detweights
    cfg = eve_amd.reset_standalone_config()
    cfg.import_json(os.path.join(REPO, 'configs', 'refine_net.json'))
    # (the kappa draw is per process: switch the augmentation off so that 2 x 1 clip and 1 x 2 clips see the same data)
    cfg.import_dict({'refine_net_rnn_type': 'CGRU', 'eye_net_load_pretrained': False,
                     'refine_net_do_offset_augmentation': False})
    model = eve_amd.EVE()
    detweights.fill_module(model.eye_net, 0)
    detweights.fill_module(model.refine_net, 1)
    return cfg, model.train(), detweights.eve_batch(2, 2, seed=13)


def _eve_worker(rank, world, port, tmp):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), EVE_AMD_BUCKET_ELEMS='1000000')
    torch.set_num_threads(2)
    from eve_amd import kernels, parallel, train
    from fake_kernels import FakeKernels
    kernels.set_default_kernels(FakeKernels())
    parallel.init_distributed(backend='gloo')
    cfg, model, full = _eve_setup()
    tr = train.eve_trainer(model, cfg, distributed=True)
    assert len(tr.modules) == 1 and tr.modules[0] is model.refine_net          # EyeNet is frozen in refine_net.json
    terms = tr.step({k: v[rank:rank + 1] for k, v in full.items()})
    torch.save({'flat': tr.fp.flat.clone(), 'grad': tr.fp.grad.clone(), 'loss': float(terms['full_loss'].detach())},
               os.path.join(tmp, 'eve_rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_eve_pipeline_step_equals_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_eve_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(str(tmp_path), 'eve_rank0.pt'))
    b = torch.load(os.path.join(str(tmp_path), 'eve_rank1.pt'))
    assert torch.equal(a['flat'], b['flat']) and torch.equal(a['grad'], b['grad']), 'ranks diverged'
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from eve_amd import kernels, train
    from fake_kernels import FakeKernels
    kernels.set_default_kernels(FakeKernels())
    try:
        cfg, model, full = _eve_setup()
        tr = train.eve_trainer(model, cfg, distributed=False)
        terms = tr.step(full)
        assert abs(0.5 * (a['loss'] + b['loss']) - float(terms['full_loss'].detach())) < 1e-4      # mean of per-clip means
        rel = float((a['grad'] / 2 - tr.fp.grad).norm() / tr.fp.grad.norm())
        assert rel < 1e-3, rel
    finally:
        kernels.set_default_kernels(None)
        import eve_amd
        eve_amd.reset_standalone_config()
