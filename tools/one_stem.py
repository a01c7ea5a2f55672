#!/usr/bin/env python
"""Run the fused stem forward / backward a few times (for rocprofv3 --pmc runs):  one_stem.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eve_amd.kernels import HipKernels  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
k = HipKernels()
src = torch.randn((N, 3, 128, 128), device='cuda')
w8 = (torch.randn((64, 7, 7, 8), device='cuda') * 0.05).bfloat16()
w8[..., 3:] = 0
xp = k.stem_pack_input(src)
for _ in range(3):
    y, idx, mr = k.stem_fwd_fused(xp, w8)
    dy = torch.randn_like(y)
    k.stem_bwd_dx(xp, w8, mr, dy, y, idx)
torch.cuda.synchronize()
