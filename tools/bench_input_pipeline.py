#!/usr/bin/env python
"""PCIe-inclusive train-step rate: the bench.py workload (EyeNet training, bf16, B=32 clips x T=30), but every
step consumes a NEW batch that starts in host memory.
  resident   the batch already on the device (what bench.py measures)
  uint8      decoded uint8 [B,T,H,W,C] clips (94 MB per step) through data.DevicePrefetcher -- pinned staging, copy stream, one
             batch of look-ahead -- normalised on the device straight into the stem's packed layout
  float      the reference's hand-over: float32 NCHW clips (377 MB per step) copied synchronously from pageable memory
             (core/training.py:257-261 does .to(device, non_blocking=True) on DataLoader output)"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import eve_amd  # noqa: E402
from eve_amd import data, train  # noqa: E402

B, T, S, STEPS = 32, 30, 128, 40
cfg = eve_amd.reset_standalone_config()
cfg.import_json(os.path.join(REPO, 'configs', 'eye_net.json'))
net = eve_amd.EyeNet()
net.compute_dtype = torch.bfloat16
net = net.cuda()
tr = train.eyenet_trainer(net, cfg)                       # eager launches: the input changes every step
g = np.random.Generator(np.random.PCG64(0))


def labels():
    out = {}
    for s in ('left', 'right'):
        out[s + '_h'] = torch.from_numpy(g.normal(0, 0.1, size=(B, T, 2)).astype(np.float32))
        out[s + '_g_tobii'] = torch.from_numpy(g.normal(0, 0.2, size=(B, T, 2)).astype(np.float32))
        out[s + '_p'] = torch.from_numpy(g.uniform(2, 5, size=(B, T)).astype(np.float32))
        out[s + '_g_tobii_validity'] = torch.ones(B, T, dtype=torch.bool)
        out[s + '_p_validity'] = torch.ones(B, T, dtype=torch.bool)
    return out


u8_batches, f32_batches = [], []
for _ in range(3):
    lab = labels()
    u8 = {s + '_eye_patch': torch.from_numpy(g.integers(0, 256, size=(B, T, S, S, 3), dtype=np.uint8)) for s in ('left', 'right')}
    u8_batches.append(dict(lab, **u8))
    f32_batches.append(dict(lab, **{k: (v.permute(0, 1, 4, 2, 3).float() * (2.0 / 255.0) - 1.0).contiguous() for k, v in u8.items()}))


def timed(it):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for batch in it:
        tr.step(batch)
        n += 1
    torch.cuda.synchronize()
    return B * T * n / (time.perf_counter() - t0)


resident = {k: v.cuda() for k, v in f32_batches[0].items()}
for _ in range(3):
    tr.step(resident)
print('resident float batch           : %8.0f frames/s' % timed(resident for _ in range(STEPS)))
print('uint8 host -> DevicePrefetcher : %8.0f frames/s  (%d MB per step over PCIe)' % (
    timed(data.DevicePrefetcher(u8_batches[i % 3] for i in range(STEPS))), 2 * B * T * S * S * 3 // 2**20))
pinned = [{k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in b.items()} for b in u8_batches]
print('uint8 pinned -> DevicePrefetcher: %7.0f frames/s  (DataLoader(pin_memory=True))' % timed(
    data.DevicePrefetcher(pinned[i % 3] for i in range(STEPS))))
fpinned = [{k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in b.items()} for b in f32_batches]
print('float pinned -> .to(non_blocking): %6.0f frames/s  (the reference: DataLoader(pin_memory=True) + training.py:257-261)' % timed(
    {k: v.to('cuda', non_blocking=True) for k, v in fpinned[i % 3].items()} for i in range(STEPS)))
print('float host -> .to(device)      : %8.0f frames/s  (%d MB per step over PCIe)' % (
    timed({k: v.to('cuda', non_blocking=True) for k, v in f32_batches[i % 3].items()} for i in range(STEPS)),
    2 * B * T * S * S * 3 * 4 // 2**20))
