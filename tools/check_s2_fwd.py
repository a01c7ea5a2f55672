#!/usr/bin/env python
"""Stride-2 3x3 forward through the library vs torch (float32 math): check_s2_fwd.py  (EVE_CONV_WG8_S2_MIN_TILES=0 forces the
eight-wave kernel at small N)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eve_amd.kernels import HipKernels
k = HipKernels()
g = torch.Generator().manual_seed(7)
for dt in (torch.bfloat16, torch.float16):
    for N, OW, Cin, Co in ((5, 16, 64, 128), (9, 8, 128, 256), (35, 4, 256, 512), (3, 16, 128, 128), (2, 8, 64, 256), (240, 16, 64, 128)):
        x = torch.randn((N, 2 * OW, 2 * OW, Cin), generator=g).to(dt).cuda()
        w = (torch.randn((Co, 3, 3, Cin), generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(dt).cuda()      # OHWI
        b = torch.randn((Co,), generator=g).cuda()
        y = k.conv2d_fwd(x, w, b, 2, 1, epi_act=1)
        name = k.lib.eve_last_kernel().decode()
        ref = torch.relu(torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, 2, 1)).permute(0, 2, 3, 1)
        err = float((y.float() - ref).abs().max()); rel = float((y.float() - ref).norm() / ref.norm())
        print('%-8s N=%-4d OW=%-2d %3d->%3d  max|d| %.3e  rel %.3e  %s' % (str(dt)[6:], N, OW, Cin, Co, err, rel, name))
        if rel > (4e-3 if dt == torch.bfloat16 else 6e-4):
            d = (y.float() - ref).abs()
            print('   bad rows', sorted(set((d.amax(dim=(0, 2, 3)) > 0.05).nonzero().flatten().tolist())), 'cols', sorted(set((d.amax(dim=(0, 1, 3)) > 0.05).nonzero().flatten().tolist())),
                  'images', sorted(set((d.amax(dim=(1, 2, 3)) > 0.05).nonzero().flatten().tolist()))[:12])
            sys.exit(1)
print('ok')
