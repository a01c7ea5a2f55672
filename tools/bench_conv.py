#!/usr/bin/env python
"""Per-shape timing of the conv kernels on the EyeNet trunk shapes (GPU box): R back-to-back launches
between one HIP event pair, algorithmic TFLOP/s per shape and pass."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eve_amd.kernels import HipKernels  # noqa: E402

SHAPES = [  # name, IH, Cin, Cout, K, stride, pad
    ('stem7x7s2', 128, 8, 64, 7, 2, 3),
    ('l1_3x3', 32, 64, 64, 3, 1, 1),
    ('l2.0_3x3s2', 32, 64, 128, 3, 2, 1),
    ('l2.0_ds1x1s2', 32, 64, 128, 1, 2, 0),
    ('l2_3x3', 16, 128, 128, 3, 1, 1),
    ('l3.0_3x3s2', 16, 128, 256, 3, 2, 1),
    ('l3_3x3', 8, 256, 256, 3, 1, 1),
    ('l4.0_3x3s2', 8, 256, 512, 3, 2, 1),
    ('l4_3x3', 4, 512, 512, 3, 1, 1),
]


def timeit(fn, reps):
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
    dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == 'bf16') else torch.float32
    reps = 10
    k = HipKernels()
    tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
    print('%-14s %9s %9s %9s   (ms | TFLOP/s)   N=%d %s' % ('shape', 'fwd', 'dgrad', 'wgrad', N, dt))
    for name, IH, Cin, Cout, K, s, p in SHAPES:
        x = torch.randn((N, IH, IH, Cin), device='cuda').to(dt)
        w = (torch.randn((Cout, K, K, Cin), device='cuda') * 0.05).to(dt)
        wt = w.permute(3, 1, 2, 0).contiguous()
        y = k.conv2d_fwd(x, w, None, s, p)
        dy = torch.randn_like(y)
        dw = torch.zeros((Cout, K, K, Cin), device='cuda')
        real_cin = 3 if name.startswith('stem') else Cin
        flops = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * Cout * K * K * real_cin
        t_f = timeit(lambda: k.conv2d_fwd(x, w, None, s, p), reps)
        t_d = timeit(lambda: k.conv2d_dgrad(dy, wt, (IH, IH), s, p), reps) if not name.startswith('stem') else 0.0
        t_w = timeit(lambda: k.conv2d_wgrad(x, dy, K, K, s, p, dw), reps)
        mult = {'l1_3x3': 4, 'l2_3x3': 3, 'l3_3x3': 3, 'l4_3x3': 3}.get(name, 1)
        tot['fwd'] += mult * t_f; tot['dgrad'] += mult * t_d; tot['wgrad'] += mult * t_w
        f = lambda t: '%6.3f|%4.0f' % (t, flops / t / 1e9) if t > 0 else '     -    '
        print('%-14s %s %s %s  x%d' % (name, f(t_f), f(t_d), f(t_w), mult))
    print('trunk totals per step (ms): fwd %.2f dgrad %.2f wgrad %.2f' % (tot['fwd'], tot['dgrad'], tot['wgrad']))


if __name__ == '__main__':
    main()
