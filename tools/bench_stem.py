#!/usr/bin/env python
"""Time the pieces of the ResNet stem (pack, 7x7/2 conv, IN stats, IN+ReLU+max-pool fwd/bwd, weight grad)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eve_amd.kernels import HipKernels  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
k = HipKernels()


def timeit(fn, reps=10):
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


src = torch.randn((N, 3, 128, 128), device='cuda')
w8 = (torch.randn((64, 7, 7, 8), device='cuda') * 0.05).bfloat16()
w8[..., 3:] = 0
xp = k.stem_pack_input(src)
y = k.stem7x7s2_fwd(xp, w8)
mr = k.instnorm_stats(y, 1e-5)
yp, idx = k.in_relu_maxpool_fwd(y, mr)
dyp = torch.randn_like(yp)
dx = k.in_relu_maxpool_bwd(dyp, yp, idx, y, mr)
print(f"pack            {timeit(lambda: k.stem_pack_input(src, out=xp)):.3f} ms")
print(f"stem conv fwd   {timeit(lambda: k.stem7x7s2_fwd(xp, w8)):.3f} ms")
print(f"IN stats        {timeit(lambda: k.instnorm_stats(y, 1e-5)):.3f} ms")
print(f"IN+relu+pool f  {timeit(lambda: k.in_relu_maxpool_fwd(y, mr)):.3f} ms")
print(f"IN+relu+pool b  {timeit(lambda: k.in_relu_maxpool_bwd(dyp, yp, idx, y, mr)):.3f} ms")
if hasattr(k.lib, 'eve_stem_fwd_fused'):
    print(f"stem fused fwd  {timeit(lambda: k.stem_fwd_fused(xp, w8)):.3f} ms")
if hasattr(k.lib, 'eve_stem_bwd_dx'):
    yf, idf, mrf = k.stem_fwd_fused(xp, w8)
    print(f"stem bwd dx     {timeit(lambda: k.stem_bwd_dx(xp, w8, mrf, dyp, yf, idf)):.3f} ms")
if hasattr(k.lib, 'eve_stem_bwd_wgrad'):
    dw = torch.zeros((64, 7, 8, 4), device='cuda')
    dyp2 = torch.randn_like(yp)
    print(f"stem bwd+wgrad  {timeit(lambda: k.stem_bwd_wgrad(xp, w8, mrf, dyp, yf, idf, dw)):.3f} ms (one summand)")
    print(f"stem bwd+wgrad  {timeit(lambda: k.stem_bwd_wgrad(xp, w8, mrf, dyp, yf, idf, dw, dy_pool2=dyp2)):.3f} ms (two summands, as in the training step)")
    print('kernel', k.lib.eve_last_kernel().decode())
