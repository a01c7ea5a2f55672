#!/usr/bin/env python
"""conv3x3_wg8_kernel (conv_wg8.h) against ATen on the CPU and timing against the halo kernel:  check_wg8.py [N]"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_case(k, N, H, C, Co, dt, seed):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, H, H, C), generator=g).to(dt)
    w = (torch.randn((Co, 3, 3, C), generator=g) * (2.0 / (9 * C)) ** 0.5).to(dt)
    b = torch.randn((Co,), generator=g)
    y = k.conv2d_fwd(x.cuda(), w.cuda(), b.cuda(), 1, 1, epi_act=1)
    name = k.lib.eve_last_kernel().decode()
    want = torch.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, 1, 1)).permute(0, 2, 3, 1)
    e1 = float((y.float().cpu() - want).norm() / want.norm())
    dy = torch.randn((N, H, H, Co), generator=g).to(dt)
    wt = w.permute(3, 1, 2, 0).contiguous()
    dx = k.conv2d_dgrad(dy.cuda(), wt.cuda(), (H, H), 1, 1)
    wantdx = F.conv_transpose2d(dy.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
    e2 = float((dx.float().cpu() - wantdx).norm() / wantdx.norm())
    return name, e1, e2


def main():
    os.environ.setdefault('EVE_CONV_WG8', '2')          # all three instantiations (16 x 16 is opt-in)
    from eve_amd.kernels import HipKernels
    k = HipKernels()
    ok = True
    for dt in (torch.bfloat16, torch.float16):
        for (N, H, C, Co) in ((5, 16, 128, 128), (13, 8, 256, 256), (70, 4, 512, 512), (3, 8, 128, 256), (33, 4, 256, 512), (2, 16, 64, 128)):
            name, e1, e2 = run_case(k, N, H, C, Co, dt, 7)
            tol = 4e-3 if dt == torch.bfloat16 else 6e-4
            flag = 'ok' if (e1 < tol and e2 < tol) else 'FAIL'
            ok = ok and flag == 'ok'
            print('%-8s N%-3d %2dx%-2d %3d->%-3d fwd %.2e dgrad %.2e  %s  [%s]' % (str(dt).split('.')[1], N, H, H, C, Co, e1, e2, flag, name))
    print('ALL OK' if ok else 'FAILURES')
    if len(sys.argv) > 1:
        for wg8 in ('2', '0'):
            env = dict(os.environ, EVE_CONV_WG8=wg8)
            p = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_conv.py'), sys.argv[1]],
                               env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            print('EVE_CONV_WG8=%s' % wg8)
            print('\n'.join(l for l in p.stdout.splitlines() if l.startswith(('l2_3x3', 'l3_3x3', 'l4_3x3', 'shape'))))


if __name__ == '__main__':
    main()
