"""Accuracy of eve_instnorm_stats on 72x128x16 bf16 planes against float64 (run with EVE_IN_STATS_ONE_PASS=0/1)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eve_amd import kernels
k = kernels.default_kernels()
torch.manual_seed(0)
for name, x in (
    ('noise', torch.randn(4, 72, 128, 16)),
    ('offset+small', 50 + 0.01 * torch.randn(4, 72, 128, 16)),
    ('corner outlier', torch.cat([torch.full((4, 1, 128, 16), 30.0), 0.02 * torch.randn(4, 71, 128, 16)], 1)),
    ('bump', torch.exp(-((torch.arange(128).view(1, 1, 128, 1) - 40.) ** 2 + (torch.arange(72).view(1, 72, 1, 1) - 30.) ** 2) / 50.).expand(4, 72, 128, 16).contiguous()),
):
    xb = x.bfloat16().cuda()
    xf = xb.float().double()
    mean = xf.mean(dim=(1, 2)); var = xf.var(dim=(1, 2), unbiased=False); rstd = (var + 1e-5).rsqrt()
    mr = k.instnorm_stats(xb, 1e-5).double().cpu()
    print(name, 'mean err %.2e' % float((mr[..., 0] - mean.cpu()).abs().max()), 'rstd rel err %.2e' % float(((mr[..., 1] - rstd.cpu()) / rstd.cpu()).abs().max()))
