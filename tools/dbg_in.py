import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eve_amd.kernels import HipKernels
k = HipKernels()
g = torch.Generator().manual_seed(5)
for shape in ((3, 32, 32, 64), (2, 16, 16, 128)):
    x, r, dy = (torch.randn(shape, generator=g).bfloat16().cuda() for _ in range(3))
    y, mr, mask = k.instnorm_fwd_fused(x, None, None, r, 1, want_mask=True)
    fm = k.instnorm_bwd_fused(dy, None, x, mr, None, 1, True, mask=mask)
    print(k.lib.eve_last_kernel())
    fy = k.instnorm_bwd_fused(dy, y, x, mr, None, 1, True)
    print(k.lib.eve_last_kernel())
    with k.dispatch_override(in_trunk_kernels=0):
        fo = k.instnorm_bwd_fused(dy, None, x, mr, None, 1, True, mask=mask)
        print(k.lib.eve_last_kernel())
    for name, a, b in (('fm-fy', fm, fy), ('fm-fo', fm, fo)):
        for i in range(3):
            d = (a[i].float() != b[i].float())
            print(shape, name, i, int(d.sum()), d.nonzero()[:5].tolist())
            if int(d.sum()):
                idx = d.nonzero()[0]
                print(a[i][tuple(idx)], b[i][tuple(idx)], a[i].view(torch.int16)[tuple(idx)] if a[i].dtype == torch.bfloat16 else '')
    ref = torch.where(y.float() > 0, dy.float(), torch.zeros_like(dy.float()))
    print('trunk vs ref', int((fm[1].float() != ref).sum()), 'generic-y vs ref', int((fy[1].float() != ref).sum()), 'generic-mask vs ref', int((fo[1].float() != ref).sum()))
    # run-to-run stability
    for t in range(3):
        fm2 = k.instnorm_bwd_fused(dy, None, x, mr, None, 1, True, mask=mask)
        print('rerun trunk vs ref', int((fm2[1].float() != ref).sum()), 'dx same', bool(torch.equal(fm2[0], fm[0])))
