#!/usr/bin/env python
"""Round-6 experiment: Winograd F(2x2, 3x3) for ONE MFMA-bound trunk layer (ResNet layer 3: 8 x 8 x 256 -> 256,
N = 1 920 images) -- float32 transforms, 16-bit batched GEMMs -- against conv3x3_wg8_kernel<2,4,8>.

  numerics (any device): error of the 16-bit direct convolution and of Winograd with 16-bit transformed operands against a float64
                         direct convolution on the same inputs (operands rounded to the storage format first, float accumulation)
  timing (GPU):          the shipped direct kernel; the 16 batched GEMMs [30 720 x 256] x [256 x 256] on hipBLASLt (torch.bmm) --
                         the part of Winograd that runs on the matrix pipe, at a tuned library's rate; the two transforms as plain
                         tensor expressions (un-fused: they move the 4x-expanded operands through HBM)
Prints one table; profiles/r06_winograd.md holds the run and the reading."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def tiles(x):
    """x [N, C, 8, 8] -> padded 4 x 4 input tiles [N, C, 4, 4, 4, 4] (tile row, tile column, 4, 4), stride 2"""
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    return xp.unfold(2, 4, 2).unfold(3, 4, 2)


def winograd(x, w, dt, acc=torch.float32):
    """x [N, C, 8, 8], w [K, C, 3, 3] (values already representable in dt): transforms in float32, transformed operands rounded to dt,
    the 16 channel contractions accumulated in `acc`, output transform in float32."""
    f = torch.float32
    g, bt, at = G.to(f).to(x.device), BT.to(f).to(x.device), AT.to(f).to(x.device)
    U = (g @ w.to(f) @ g.t()).to(dt).to(acc)                                          # [K, C, 4, 4]
    V = (bt @ tiles(x.to(f)) @ bt.t()).to(dt).to(acc)                                 # [N, C, 4, 4, 4, 4]
    M = torch.einsum('kcuv,nctsuv->nktsuv', U, V)                                     # contraction over the input channels
    Y = at @ M.to(f) @ at.t()                                                         # [N, K, 4, 4, 2, 2]
    N, K = Y.shape[:2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, 8, 8)


def numerics(device):
    torch.manual_seed(0)
    N, C, K = 16, 256, 256
    rows = []
    for name, dt in (('bf16', torch.bfloat16), ('fp16', torch.float16), ('fp32', torch.float32)):
        x = torch.relu(torch.randn(N, C, 8, 8, device=device)).to(dt)                 # what the layer sees: relu(InstanceNorm(.))
        w = (torch.randn(K, C, 3, 3, device=device) * (2.0 / (9 * K)) ** 0.5).to(dt)
        ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
        direct = torch.nn.functional.conv2d(x.float(), w.float(), padding=1)          # products of representable values, float sums
        wino = winograd(x, w, dt)
        e = lambda y: (float((y.double() - ref).norm() / ref.norm()), float((y.double() - ref).abs().max()))
        rows.append((name, e(direct), e(wino), float(ref.abs().max())))
    return rows


def timing():
    from eve_amd.kernels import HipKernels
    k = HipKernels()
    N, C = 1920, 256
    x = torch.relu(torch.randn(N, 8, 8, C, device='cuda')).bfloat16()
    w = (torch.randn(C, 3, 3, C, device='cuda') * 0.03).bfloat16()

    def t(fn, reps=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    direct = t(lambda: k.conv2d_fwd(x, w, None, 1, 1))
    name = k.lib.eve_last_kernel().decode()
    V = torch.randn(16, N * 16, C, device='cuda').bfloat16()
    U = torch.randn(16, C, C, device='cuda').bfloat16()
    out = torch.empty(16, N * 16, C, device='cuda', dtype=torch.bfloat16)
    gemm = t(lambda: torch.bmm(V, U, out=out))
    xn = x.permute(0, 3, 1, 2).float().contiguous()
    bt = BT.float().cuda()
    at = AT.float().cuda()
    tin = t(lambda: (bt @ tiles(xn) @ bt.t()).bfloat16(), reps=5)
    Mt = torch.randn(N, C, 4, 4, 4, 4, device='cuda')
    tout = t(lambda: (at @ Mt @ at.t()).bfloat16(), reps=5)
    return dict(direct_ms=direct, kernel=name, gemm_ms=gemm, gemm_tflops=2.0 * 16 * N * 16 * C * C / gemm / 1e9,
                direct_tflops=2.0 * N * 64 * C * 9 * C / direct / 1e9, in_transform_ms=tin, out_transform_ms=tout)


if __name__ == '__main__':
    dev = 'cuda' if torch.cuda.is_available() else 'cpu'
    print('numerics on %s: relative L2 / max abs error against a float64 direct convolution (8 x 8 x 256 -> 256, 16 images)' % dev)
    for name, d, wv, scale in numerics(dev):
        print('  %-5s direct %.3e / %.3e   Winograd F(2x2,3x3) %.3e / %.3e   (x %.1f; |y| max %.2f)' % (name, d[0], d[1], wv[0], wv[1], wv[0] / d[0], scale))
    if dev == 'cuda':
        r = timing()
        print('timing, N = 1 920 images, bf16:')
        print('  direct %-42s %.4f ms  %.0f TFLOP/s (algorithmic)' % (r['kernel'], r['direct_ms'], r['direct_tflops']))
        print('  Winograd GEMM part, 16 x [30720 x 256] x [256 x 256] on hipBLASLt (torch.bmm) %.4f ms  %.0f TFLOP/s of its own 64.4 GFLOP' % (r['gemm_ms'], r['gemm_tflops']))
        print('  input transform (tensor expressions, un-fused) %.3f ms; output transform %.3f ms' % (r['in_transform_ms'], r['out_transform_ms']))
