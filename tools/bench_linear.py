#!/usr/bin/env python
"""Per-launch timing of the float32 tail products (linear_mm_kernel forward / data gradient, linear_wgrad_batch_kernel) at the
EyeNet tail's shapes:  bench_linear.py [M = 1920].  R back-to-back launches between one HIP event pair."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eve_amd.kernels import HipKernels  # noqa: E402

SHAPES = [('fc', 512, 128), ('fc_common.0', 132, 128), ('fc_common.2', 128, 128), ('gru_ih', 128, 384), ('head.0', 128, 128), ('head.2', 128, 4)]


def timeit(fn, reps=50):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
    k = HipKernels()
    g = torch.Generator(device='cpu').manual_seed(0)
    probs = []
    print('M = %d   (us per launch)' % M)
    for name, K, N in SHAPES:
        x = torch.randn((M, K), generator=g).cuda()
        w = (torch.randn((N, K), generator=g) * K ** -0.5).cuda()
        wt = w.t().contiguous()
        b = torch.randn((N,), generator=g).cuda()
        y = k.linear_fwd(x, wt, b, 3)
        dy = torch.randn((M, N), generator=g).cuda()
        t_f = timeit(lambda: k.linear_fwd(x, wt, b, 3))
        t_d = timeit(lambda: k.linear_dgrad(dy, y, 3, w))
        dw, db = torch.zeros((N, K), device='cuda'), torch.zeros((N,), device='cuda')
        t_w = timeit(lambda: k.linear_wgrad(dy, y, 3, x, dw, db))
        print('%-12s K=%3d N=%3d   fwd %6.1f   dgrad %6.1f   wgrad %6.1f' % (name, K, N, t_f, t_d, t_w))
        probs.append(dict(dY=dy, Y=y, act=3, X=x, dW=dw, db=db))
    print('wgrad batch of %d: %.1f us' % (len(probs), timeit(lambda: k.linear_wgrad_batch(probs))))


if __name__ == '__main__':
    main()
