#!/usr/bin/env python
"""InstanceNorm (affine + ReLU) on RefineNet's planes (960 frames): two-pass kernels vs the register-resident ones dealt by
channels (in_big_planes).  ms and algorithmic TB/s per launch.  bench_in_refine.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eve_amd.kernels import HipKernels  # noqa: E402
from bench_in import timeit  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 960
    k = HipKernels()
    dt = torch.bfloat16
    for name, H, W, C in (('L0 16', 72, 128, 16), ('L0 32', 72, 128, 32), ('L1 32', 36, 64, 32), ('L1 64', 36, 64, 64),
                          ('L2 64', 18, 32, 64), ('L2 128', 18, 32, 128), ('L3 128', 9, 16, 128)):
        x = torch.randn((N, H, W, C), device='cuda').to(dt)
        dy = torch.randn_like(x)
        g = torch.rand((C,), device='cuda') + 0.5
        b = torch.randn((C,), device='cuda')
        mb = x.numel() * 2 / 1e6
        for big in (0, 1):
            with k.dispatch_override(in_big_planes=big):
                def fwd():
                    out = k.instnorm_fwd_fused(x, g, b, None, 1)
                    if out is None:
                        mr = k.instnorm_stats(x, 1e-5)
                        return k.instnorm_act_fwd(x, mr, g, b, None, 1), mr
                    return out
                t = timeit(fwd)
                kn = k.lib.eve_last_kernel().decode()
                y, mr = fwd()

                def bwd():
                    out = k.instnorm_bwd_fused(dy, y, x, mr, g, 1, False)
                    if out is None:
                        out = k.instnorm_act_bwd(dy, y, x, mr, g, 1, False, beta=b)
                    return out
                t2 = timeit(bwd)
                print('%-7s big=%d  fwd %.3f ms %.2f TB/s   bwd %.3f ms %.2f TB/s   %s | %s' % (
                    name, big, t, 2 * mb / t / 1e3, t2, 4 * mb / t2 / 1e3, kn[:40], k.lib.eve_last_kernel().decode()[:40]))


if __name__ == '__main__':
    main()
