#!/usr/bin/env python
"""Stride-2 3x3 data gradient through the library vs torch (float32 math): check_s2_dgrad.py  (set EVE_CONV_WG8_MIN_TILES=0 to
force the eight-wave path at small N)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eve_amd.kernels import HipKernels
k = HipKernels()
g = torch.Generator().manual_seed(5)
for dt in (torch.bfloat16, torch.float16):
    for N, OW, Cdx, Co in ((5, 16, 64, 128), (9, 8, 128, 256), (35, 4, 256, 512), (3, 16, 128, 128), (1920 // 8, 16, 64, 128)):
        dy = torch.randn((N, OW, OW, Co), generator=g).to(dt).cuda()
        w = (torch.randn((Co, Cdx, 3, 3), generator=g) * (2.0 / (9 * Co)) ** 0.5).to(dt).cuda()      # OIHW of the forward conv
        w_ihwo = w.permute(1, 2, 3, 0).contiguous()
        dx = k.conv2d_dgrad(dy, w_ihwo, (2 * OW, 2 * OW), 2, 1)
        name = k.lib.eve_last_kernel().decode()
        ref = torch.nn.grad.conv2d_input((N, Cdx, 2 * OW, 2 * OW), w.float(), dy.float().permute(0, 3, 1, 2), stride=2, padding=1).permute(0, 2, 3, 1)
        err = float((dx.float() - ref).abs().max()); rel = float((dx.float() - ref).norm() / ref.norm())
        print('%-8s N=%-4d OW=%-2d %3d<-%3d  max|d| %.3e  rel %.3e  %s' % (str(dt)[6:], N, OW, Cdx, Co, err, rel, name))
        if rel > (4e-3 if dt == torch.bfloat16 else 6e-4):
            d = (dx.float() - ref)
            for py in (0, 1):
                for px in (0, 1):
                    e = d[:, py::2, px::2]; r = ref[:, py::2, px::2]
                    print('   class py=%d px=%d rel %.3e' % (py, px, float(e.norm() / r.norm())), 'rows bad:', sorted(set((e.abs().amax(dim=(0, 2, 3)) > 0.05).nonzero().flatten().tolist()))[:20], 'cols bad:', sorted(set((e.abs().amax(dim=(0, 1, 3)) > 0.05).nonzero().flatten().tolist()))[:20])
            sys.exit(1)
print('ok')
