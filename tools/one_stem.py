#!/usr/bin/env python
"""One pass of the fused stem kernels (for rocprofv3 --pmc):  one_stem.py [N] [reps]"""
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
yf, idf, mrf = k.stem_fwd_fused(xp, w8)
dyp = torch.randn_like(yf)
dw = torch.zeros((64, 7, 8, 4), device='cuda')
for _ in range(reps):
    k.stem_fwd_fused(xp, w8)
    k.stem_bwd_wgrad(xp, w8, mrf, dyp, yf, idf, dw)
    dconv = k.stem_bwd_dx(xp, w8, mrf, dyp, yf, idf)
    k.stem_wgrad(xp, dconv, dw)
torch.cuda.synchronize()
if len(sys.argv) > 3:
    def timeit(fn, n=10):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n
    print('fwd %.3f  bwd_wgrad %.3f  bwd_dx %.3f  wgrad %.3f ms' % (
        timeit(lambda: k.stem_fwd_fused(xp, w8)), timeit(lambda: k.stem_bwd_wgrad(xp, w8, mrf, dyp, yf, idf, dw)),
        timeit(lambda: k.stem_bwd_dx(xp, w8, mrf, dyp, yf, idf)), timeit(lambda: k.stem_wgrad(xp, dconv, dw))))
