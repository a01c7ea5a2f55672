#!/usr/bin/env python
"""ResNet layer 1 on 256 x 256 patches (64 x 64 x 64 planes): conv3x3_ws64_kernel<.., 64> against the kernel it replaces."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eve_amd.kernels import HipKernels  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 960
W = int(sys.argv[2]) if len(sys.argv) > 2 else 64
C = int(sys.argv[3]) if len(sys.argv) > 3 else 64
KNOB = 'conv_ws64' if C == 64 else 'conv_wg8'
k = HipKernels()
x = torch.randn((N, W, W, C), device='cuda').half()
w = (torch.randn((C, 3, 3, C), device='cuda') * 0.05).half()
wt = w.permute(3, 1, 2, 0).contiguous()


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


flops = 2.0 * N * W * W * C * C * 9
for on in (1, 0):
    with k.dispatch_override(**{KNOB: on}):
        tf = timeit(lambda: k.conv2d_fwd(x, w, None, 1, 1))
        name = k.lib.eve_last_kernel().decode()
        td = timeit(lambda: k.conv2d_dgrad(x, wt, (W, W), 1, 1))
    print('N=%d W=%d conv_ws64=%d  %-44s forward %.3f ms (%.0f TFLOP/s, %.2f TB/s)   data gradient %.3f ms' % (
        N, W, on, name, tf, flops / tf / 1e9, 2 * x.numel() * 2 / tf / 1e9, td))
