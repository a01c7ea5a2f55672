import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eve_amd.kernels import HipKernels
k = HipKernels()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
g = torch.Generator().manual_seed(1)
x = torch.randn((N, 32, 32, 64), generator=g).bfloat16().cuda()
w = (torch.randn((64, 3, 3, 64), generator=g) * 0.05).bfloat16().cuda()
y = k.conv2d_fwd(x, w, None, 1, 1)
print(k.lib.eve_last_kernel().decode())
ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
bad = ((y.float() - ref).abs() > 0.1) | ~torch.isfinite(y.float())
print('bad elements', int(bad.sum()), 'of', bad.numel())
idx = bad.nonzero()
if len(idx):
    print('images', sorted(set(idx[:, 0].tolist()))[:40])
    print('rows', sorted(set(idx[:, 1].tolist())))
    print('cols', sorted(set(idx[:, 2].tolist())))
    print('chans', sorted(set(idx[:, 3].tolist())))
    print(idx[:10].tolist())
