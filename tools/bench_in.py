#!/usr/bin/env python
"""InstanceNorm kernels on the trunk's planes: ms and algorithmic TB/s per launch.  bench_in.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eve_amd.kernels import HipKernels  # noqa: E402


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
    k = HipKernels()
    over = dict(a.split('=') for a in sys.argv[2:])          # e.g. in_split=0 in_trunk_kernels=0
    with k.dispatch_override(**{n: int(v) for n, v in over.items()}):
        run(k, N)


def run(k, N):
    dt = torch.bfloat16
    for name, H, C in (('layer1', 32, 64), ('layer2', 16, 128), ('layer3', 8, 256), ('layer4', 4, 512)):
        x = torch.randn((N, H, H, C), device='cuda').to(dt)
        r = torch.randn_like(x)
        dy, dy2 = torch.randn_like(x), torch.randn_like(x)
        mb = x.numel() * 2 / 1e6
        t = timeit(lambda: k.instnorm_fwd_fused(x, None, None, None, 1))
        print('%-7s fwd  IN+ReLU            %.3f ms  %.2f TB/s' % (name, t, 2 * mb / t / 1e3))
        t = timeit(lambda: k.instnorm_fwd_fused(x, None, None, r, 1, want_mask=True))
        y, mr, mask = k.instnorm_fwd_fused(x, None, None, r, 1, want_mask=True)
        print('%-7s fwd  IN+res+ReLU+mask   %.3f ms  %.2f TB/s' % (name, t, (3 + 1 / 16) * mb / t / 1e3))
        t = timeit(lambda: k.instnorm_bwd_fused(dy, None, x, mr, None, 1, False))
        print('%-7s bwd  mid-block (dy,x)   %.3f ms  %.2f TB/s' % (name, t, 3 * mb / t / 1e3))
        t = timeit(lambda: k.instnorm_bwd_fused(dy, None, x, mr, None, 1, True, mask=mask, dy2=dy2))
        print('%-7s bwd  block end (dy,dy2,x,mask -> dx,dres) %.3f ms  %.2f TB/s  %s' % (name, t, (5 + 1 / 16) * mb / t / 1e3, k.lib.eve_last_kernel().decode()))


if __name__ == '__main__':
    main()
