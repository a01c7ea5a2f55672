#!/usr/bin/env python
"""Two-head InstanceNorm backward on RefineNet's planes: ms per launch (EVE_IN_CLUSTER=0/1).  bench_in2.py [N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eve_amd.kernels import HipKernels
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N = int(sys.argv[1]) if len(sys.argv) > 1 else 960
k = HipKernels()
for name, H, W, C1, C2 in (('72x128 32+32', 72, 128, 32, 32), ('36x64 64+64', 36, 64, 64, 64), ('72x128 16', 72, 128, 16, 0), ('18x32 128+128', 18, 32, 128, 128)):
    xs = [torch.randn((N, H, W, C1), device='cuda').bfloat16()] + ([torch.randn((N, H, W, C2), device='cuda').bfloat16()] if C2 else [])
    ct = C1 + C2
    ga, ba, gb, bb = (torch.randn(ct, device='cuda') for _ in range(4))
    da, db = torch.randn((N, H, W, ct), device='cuda').bfloat16(), torch.randn((N, H, W, ct), device='cuda').bfloat16()
    mrs = [k.instnorm_stats(x) for x in xs]
    t = timeit(lambda: k.instnorm_act2_bwd(da, db, xs, mrs, ga, ba, gb, bb, 2))
    mb = (da.numel() * 2 * 2 + sum(x.numel() for x in xs) * 2 * 2) / 1e6
    print('%-16s %.3f ms  %.2f TB/s (single-pass bytes)  %s' % (name, t, mb / t / 1e3, k.lib.eve_last_kernel().decode()))
