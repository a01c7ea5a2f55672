import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from eve_amd.kernels import HipKernels
k = HipKernels()
g = torch.Generator().manual_seed(5)
for dt in (torch.float32, torch.float16):
    for shape in ((2, 4, 4, 512), (2, 16, 16, 128)):
        x = (torch.randn(shape, generator=g) * 1.7).to(dt).cuda(); r = torch.randn(shape, generator=g).to(dt).cuda()
        for res in (None, r):
            outs = {}
            for trunk in (0, 1):
                with k.dispatch_override(in_trunk_kernels=trunk):
                    y, mr = k.instnorm_fwd_fused(x, None, None, res, 1)
                    n1 = k.lib.eve_last_kernel().decode()
                    ym, mrm, mask = k.instnorm_fwd_fused(x, None, None, res, 1, want_mask=True)
                    n2 = k.lib.eve_last_kernel().decode()
                outs[trunk] = (y, mr, ym, mrm, n1, n2)
            a, b = outs[0], outs[1]
            print(dt, shape, 'res' if res is not None else 'nores', b[4][:22], b[5][:22],
                  'y', int((a[0] != b[0]).sum()), 'mr', int((a[1] != b[1]).sum()), 'ym', int((a[2] != b[2]).sum()), 'mrm', int((a[3] != b[3]).sum()),
                  'gen y vs gen ym', int((a[0] != a[2]).sum()), 'trunk/gen mix', int((b[0] != b[2]).sum()))
