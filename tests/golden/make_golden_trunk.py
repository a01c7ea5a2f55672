#!/usr/bin/env python
"""Golden fixture for the ResNet-18(InstanceNorm) trunk from an INDEPENDENT third-party implementation.

    python tests/golden/make_golden_trunk.py          (build container; needs `transformers`, no /root/reference)

Why: the trunk arithmetic of the reference lives in torchvision 0.6.1 (`torchvision/models/resnet.py`, used at
/root/reference/src/models/eye_net.py:26,48-50,106), which is neither vendored nor installable here, and the reference
holds no test for it; the other fixtures that pass through the trunk were produced with oracle/resnet_in.py injected as
the torchvision stand-in, i.e. they compare the restatement with itself.  The image does ship another implementation of
the same published architecture: `transformers.models.resnet.modeling_resnet` (`layer_type='basic'`).  This script
builds it with depths [2,2,2,2] / hidden sizes [64,128,256,512], swaps every BatchNorm2d for
`nn.InstanceNorm2d(C)` (what `norm_layer=nn.InstanceNorm2d` does in eye_net.py:48-50), loads the deterministic weights of
oracle/detweights.py under torchvision's parameter names, and stores stage outputs, the `fc` output and per-parameter
gradients.  tests/test_oracle_golden.py checks oracle/resnet_in.py against it (CPU), tests/test_gpu_eyenet.py the HIP
trunk.  Only numbers are written.  Nothing from oracle/ computes anything here (detweights only generates inputs).
"""
import os
import sys

import numpy as np
import torch
from torch import nn

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, REPO)

from oracle import detweights  # noqa: E402  (deterministic weights / inputs only)


def torchvision_name(hf_name):
    """transformers parameter name -> torchvision ResNet name (the reference's state_dict key under `cnn_layers.`)."""
    if hf_name == 'embedder.embedder.convolution.weight':
        return 'conv1.weight'
    p = hf_name.split('.')                      # encoder.stages.S.layers.B.(layer.J|shortcut).convolution.weight
    assert p[0] == 'encoder' and p[1] == 'stages' and p[3] == 'layers' and p[-2:] == ['convolution', 'weight'], hf_name
    s, b = int(p[2]), int(p[4])
    if p[5] == 'shortcut':
        return 'layer%d.%d.downsample.0.weight' % (s + 1, b)
    return 'layer%d.%d.conv%d.weight' % (s + 1, b, int(p[6]) + 1)


def build_trunk(seed=0):
    from transformers.models.resnet.modeling_resnet import ResNetConfig, ResNetModel
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[64, 128, 256, 512], depths=[2, 2, 2, 2],
                       layer_type='basic', hidden_act='relu', downsample_in_first_stage=False)
    model = ResNetModel(cfg)

    def swap(module):
        for name, child in module.named_children():
            if isinstance(child, nn.BatchNorm2d):
                setattr(module, name, nn.InstanceNorm2d(child.num_features))     # affine=False, eps=1e-5, no running stats
            else:
                swap(child)
    swap(model)
    assert not any(isinstance(m, nn.BatchNorm2d) for m in model.modules())
    names = {}
    with torch.no_grad():
        for n, p in model.named_parameters():
            tv = torchvision_name(n)
            p.copy_(detweights.tensor_for('cnn_layers.' + tv, p.shape, seed))
            names[n] = tv
    fc = nn.Linear(512, 128)
    with torch.no_grad():
        fc.weight.copy_(detweights.tensor_for('cnn_layers.fc.weight', fc.weight.shape, seed))
        fc.bias.copy_(detweights.tensor_for('cnn_layers.fc.bias', fc.bias.shape, seed))
    return model.train(), fc, names


def run_case(size, B, T, seed):
    model, fc, names = build_trunk(0)
    batch = detweights.eyenet_batch(B, T, size=size, seed=seed)
    x = torch.cat([batch['left_eye_patch'].reshape(B * T, 3, size, size),
                   batch['right_eye_patch'].reshape(B * T, 3, size, size)], dim=0)
    out = model(x, output_hidden_states=True)
    feats = out.pooler_output.flatten(1)
    y = fc(feats)
    # a fixed, non-degenerate scalar of the output for the gradient check
    g = np.random.Generator(np.random.PCG64(77 + seed))
    proj = torch.from_numpy(g.standard_normal(tuple(y.shape)).astype(np.float32))
    (y * proj).sum().backward()
    fix = {'size': size, 'B': B, 'T': T, 'seed': seed, 'proj': proj.numpy()}
    taps = dict(zip(('maxpool', 'layer1', 'layer2', 'layer3', 'layer4'), out.hidden_states))
    for nm, v in taps.items():
        v = v.detach().double()
        fix['tap_%s_sum' % nm] = v.sum().numpy()
        fix['tap_%s_sqsum' % nm] = (v * v).sum().numpy()
        fix['tap_%s_head' % nm] = v.reshape(v.shape[0], -1)[:, :256].float().numpy()
    fix['layer4'] = taps['layer4'].detach().numpy().astype(np.float32)
    fix['pooled'] = feats.detach().numpy().astype(np.float32)
    fix['fc'] = y.detach().numpy().astype(np.float32)
    gn, gnorm, ghead = [], [], []
    for n, p in list(model.named_parameters()) + [('fc.weight', fc.weight), ('fc.bias', fc.bias)]:
        gn.append(names.get(n, n))
        gnorm.append(float(p.grad.double().norm()))
        h = np.zeros(64, np.float32)
        flat = p.grad.reshape(-1)
        h[:min(64, flat.numel())] = flat[:64].numpy()
        ghead.append(h)
    fix['grad_names'], fix['grad_norms'], fix['grad_heads'] = np.array(gn), np.array(gnorm, np.float64), np.stack(ghead)
    return fix


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    fix = {}
    for tag, (size, B, T, seed) in {'p128': (128, 2, 2, 5), 'p256': (256, 1, 1, 6)}.items():
        for k, v in run_case(size, B, T, seed).items():
            fix['%s_%s' % (tag, k)] = v
    path = os.path.join(OUT, 'trunk_independent.npz')
    np.savez_compressed(path, **fix)
    print(path, os.path.getsize(path), 'bytes; fc head', fix['p128_fc'][0, :4])


if __name__ == '__main__':
    main()
