#!/usr/bin/env python
"""A/B of the fused stem forward kernels on one box: eve_dispatch_config.stem_fwd_pairs = 0 (one wave per image, table-coded
pooling) and 1 (wave pairs, lean pooling since round 6).  Checks 1 against 0 (arg-max bit for bit, statistics to float
round-off) and times each at the given image counts:  python tools/ab_stem_fwd.py [N ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eve_amd.kernels import HipKernels  # noqa: E402

k = HipKernels()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for N in [int(a) for a in sys.argv[1:]] or [1920, 480, 240]:
    for dt in (torch.bfloat16, torch.float16):
        torch.manual_seed(N)
        src = torch.randn((N, 3, 128, 128), device='cuda') + 0.2
        w8 = (torch.randn((64, 7, 7, 8), device='cuda') * 0.05).to(dt)
        w8[..., 3:] = 0
        xp = k.stem_pack_input(src, dtype=dt)
        out = {}
        for mode in (0, 1):
            with k.dispatch_override(stem_fwd_pairs=mode):
                y, idx, mr = k.stem_fwd_fused(xp, w8)
                name = k.lib.eve_last_kernel().decode()
                ms = timeit(lambda: k.stem_fwd_fused(xp, w8))
            out[mode] = (y.clone(), idx.clone(), mr.clone())
            print('N=%5d %s stem_fwd_pairs=%d %-44s %.4f ms' % (N, str(dt).split('.')[1], mode, name.replace('eve::', ''), ms))
        (y1, i1, m1), (y2, i2, m2) = out[0], out[1]
        ok = torch.equal(y1.view(torch.int16), y2.view(torch.int16)) and torch.equal(i1, i2)
        print('   pairs == one-wave kernel: pooled/arg-max bit-equal %s, |d mean| %.2e, rel |d rstd| %.2e' % (
            ok, float((m1[..., 0] - m2[..., 0]).abs().max()), float(((m1[..., 1] - m2[..., 1]) / m1[..., 1]).abs().max())))
        if not ok:
            print('   y differs at %d of %d, idx at %d' % (int((y1.view(torch.int16) != y2.view(torch.int16)).sum()), y1.numel(),
                                                         int((i1 != i2).sum())))
