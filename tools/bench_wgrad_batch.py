#!/usr/bin/env python
"""The tail's batched weight gradient, problem by problem (M = 2*B*T rows)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eve_amd.kernels import HipKernels  # noqa: E402
from bench_in import timeit  # noqa: E402

k = HipKernels()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
T = 30
r = lambda *s: torch.randn(s, device='cuda')
probs = {
    'fc 512->128 (dY ld 132)': dict(dY=r(M, 132), X=r(M, 512), dW=torch.zeros(128, 512, device='cuda'), db=torch.zeros(128, device='cuda')),
    'fc_common.0 130->128 selu': dict(dY=r(M, 128), Y=r(M, 128), act=3, X=r(M, 132), K1=130, dW=torch.zeros(128, 130, device='cuda'), db=torch.zeros(128, device='cuda')),
    'fc_common.2 128->128': dict(dY=r(M, 128), X=r(M, 128), dW=torch.zeros(128, 128, device='cuda'), db=torch.zeros(128, device='cuda')),
    'ih 128->384': dict(dY=r(M, 384), X=r(M, 128), dW=torch.zeros(384, 128, device='cuda'), db=torch.zeros(384, device='cuda')),
    'hh 128->384 shifted': dict(dY=r(M, 384), X=r(M, 128), x_shift_T=T, dW=torch.zeros(384, 128, device='cuda'), db=torch.zeros(384, device='cuda')),
    'g0 128->128 selu': dict(dY=r(M, 128), Y=r(M, 128), act=3, X=r(M, 128), dW=torch.zeros(128, 128, device='cuda'), db=torch.zeros(128, device='cuda')),
    'g2 128->2 tanh (ld 4)': dict(dY=r(M, 4), Y=r(M, 4), act=4, X=r(M, 128), dW=torch.zeros(2, 128, device='cuda')),
    'p2 128->1 relu (ld 4)': dict(dY=r(M, 4), Y=r(M, 4), act=1, X=r(M, 128), dW=torch.zeros(1, 128, device='cuda'), db=torch.zeros(1, device='cuda')),
}
for name, p in probs.items():
    print('%-28s %.1f us' % (name, 1e3 * timeit(lambda: k.linear_wgrad_batch([p]), reps=20)))
allp = list(probs.values()) + [probs['g0 128->128 selu']]
print('%-28s %.1f us' % ('all nine', 1e3 * timeit(lambda: k.linear_wgrad_batch(allp), reps=20)))
