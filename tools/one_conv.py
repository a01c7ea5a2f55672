#!/usr/bin/env python
"""Run one conv shape a few times (for rocprofv3 --pmc runs):  one_conv.py <pass> IH Cin Cout K stride pad [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eve_amd.kernels import HipKernels  # noqa: E402

which, IH, Cin, Cout, K, s, p = sys.argv[1], *map(int, sys.argv[2:8])
N = int(sys.argv[8]) if len(sys.argv) > 8 else 1920
k = HipKernels()
x = torch.randn((N, IH, IH, Cin), device='cuda').bfloat16()
w = (torch.randn((Cout, K, K, Cin), device='cuda') * 0.05).bfloat16()
wt = w.permute(3, 1, 2, 0).contiguous()
y = k.conv2d_fwd(x, w, None, s, p)
dy = torch.randn_like(y)
dw = torch.zeros((Cout, K, K, Cin), device='cuda')
torch.cuda.synchronize()
for _ in range(3):
    if which == 'fwd':
        k.conv2d_fwd(x, w, None, s, p)
    elif which == 'dgrad':
        k.conv2d_dgrad(dy, wt, (IH, IH), s, p)
    else:
        k.conv2d_wgrad(x, dy, K, K, s, p, dw)
torch.cuda.synchronize()
