#!/usr/bin/env python
"""Stage-by-stage comparison of the HIP bf16 EyeNet with oracle/bf16_faithful.py (GPU box; diagnostic tool).
'local': the HIP stage is fed the ORACLE's (bf16-exact) input, so the line shows that stage's own error;
'cumul': the HIP chain runs on its own outputs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eve_amd  # noqa: E402
from eve_amd import ops  # noqa: E402
from eve_amd.kernels import default_kernels  # noqa: E402
from oracle import bf16_faithful as bf  # noqa: E402
from oracle import detweights, sequence  # noqa: E402
from oracle.config import OracleConfig  # noqa: E402
from oracle.eye_net import EyeNet as OracleEyeNet  # noqa: E402


def report(name, got, want):
    got = got.float().cpu().permute(0, 3, 1, 2) if got.dim() == 4 else got.float().cpu()
    want = want.detach()
    d = (got - want).abs()
    print('%-34s differ %8.4f%%  max|d| %.3e  rms d %.3e  rms(want) %.3f' % (
        name, 100.0 * float((d > 0).float().mean()), float(d.max()), float(d.pow(2).mean().sqrt()), float(want.pow(2).mean().sqrt())))


def nhwc(t):
    return t.detach().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()


def main():
    cfg = OracleConfig(batch_size=16, weight_decay=0.005, base_learning_rate=0.001)
    eve_amd.reset_standalone_config()
    ref = detweights.fill_module(OracleEyeNet(cfg), seed=0)
    net = eve_amd.EyeNet()
    net.compute_dtype = torch.bfloat16
    detweights.fill_module(net, seed=0)
    net.cuda()
    P = net._get_packs()
    k = default_kernels()
    B, T = 2, 3
    batch = detweights.eyenet_batch(B, T, seed=0)
    x = batch['left_eye_patch'].reshape(B * T, 3, 128, 128)
    cnn = ref.cnn_layers
    with torch.no_grad():
        y = bf.resnet_stem(cnn, x)
        xp = torch.empty((B * T, 134, 136, 4), dtype=torch.bfloat16, device='cuda')
        k.stem_pack_input(x.cuda(), out=xp)
        hy, idx, mr = k.stem_fwd_fused(xp, P['conv1'].ohwi, 1e-5)
        report('stem (fused)', hy, y)
        hcum = hy
        for name, blk in net.cnn_layers.blocks():
            oblk = dict(cnn.named_modules())[name]
            packs = (P[name + '.conv1'], P[name + '.conv2'], P[name + '.downsample.0'] if blk.downsample is not None else None)
            yo = bf.resnet_block(oblk, y)
            hl, _ = ops._block_forward(k, nhwc(y), packs, blk.stride, 1e-5)
            report(name + ' local', hl, yo)
            hcum, _ = ops._block_forward(k, hcum, packs, blk.stride, 1e-5)
            report(name + ' cumul', hcum, yo)
            y = yo
        feats = bf.R(y.mean(dim=(2, 3)))
        report('avgpool cumul', k.avgpool_fwd(hcum), feats)
    # the tail and the whole network
    out = net.forward_sequence({kk: v.cuda() for kk, v in batch.items()})
    rout = bf.eyenet_sequence(ref, batch)
    for kk in ('left_g_initial', 'left_pupil_size', 'left_eye_rnn_states_0'):
        report('network ' + kk, out[kk].detach(), rout[kk])
    # tail alone on the oracle's features
    with torch.no_grad():
        feats = bf.resnet_trunk(cnn, x).view(B, T, -1)
        gaze, pupil, st = net._tail(feats.reshape(B * T, -1).cuda(), batch['left_h'].reshape(B * T, 2).cuda(), B, T, None, P)
        report('tail on oracle features (gaze)', gaze.view(B, T, 2), rout['left_g_initial'])
    # gradient entering the trunk: d loss / d feats, both sides
    terms = sequence.eyenet_losses(rout, batch, cfg)
    terms['full_loss'].backward()
    dbatch = {kk: v.cuda() for kk, v in batch.items()}
    sequence.eyenet_losses(out, dbatch, cfg)['full_loss'].backward()
    for n in ('fc_to_gaze.2.weight', 'fc_common.0.weight', 'cnn_layers.fc.weight', 'cnn_layers.layer4.1.conv2.weight',
              'cnn_layers.layer4.0.conv1.weight', 'cnn_layers.layer1.0.conv1.weight', 'cnn_layers.conv1.weight'):
        a = dict(net.named_parameters())[n].grad.cpu().double()
        b = dict(ref.named_parameters())[n].grad.double()
        print('grad %-36s rel L2 %.3e  |ref| %.3e' % (n, float((a - b).norm() / b.norm()), float(b.norm())))


if __name__ == '__main__':
    main()
