#!/usr/bin/env python
"""A few launches of the fused stem forward and backward, for rocprofv3 counter passes:  one_stem.py [N] [reps] [time]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eve_amd.kernels import HipKernels  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
k = HipKernels()
src = torch.randn((N, 3, 128, 128), device='cuda')
w8 = (torch.randn((64, 7, 7, 8), device='cuda') * 0.05).bfloat16()
w8[..., 3:] = 0
xp = k.stem_pack_input(src)
y, idx, mr = k.stem_fwd_fused(xp, w8)
dy, dy2 = torch.randn_like(y), torch.randn_like(y)
dw = torch.zeros((64, 7, 8, 4), device='cuda')
for _ in range(reps):
    k.stem_fwd_fused(xp, w8)
    k.stem_bwd_wgrad(xp, w8, mr, dy, y, idx, dw, dy_pool2=dy2)
torch.cuda.synchronize()
if len(sys.argv) > 3:
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        k.stem_bwd_wgrad(xp, w8, mr, dy, y, idx, dw, dy_pool2=dy2)
    e.record()
    torch.cuda.synchronize()
    print('stem backward + weight gradient, N = %d: %.3f ms' % (N, s.elapsed_time(e) / 10))
