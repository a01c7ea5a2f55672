"""Synthetic clips and weights for the benchmark / sanity tools (no dataset, no checkpoint on the box): tensors with the
schema of the reference's data loader (/root/reference/src/datasources/eve_sequences.py:215-299) and the value ranges of
SURVEY.md 8(d).  Independent of the test oracle."""
import math

import torch


def fill_module(module, seed=0):
    """Deterministic non-degenerate weights: N(0, fan-in-scaled) for matrices / filters (also the zero-initialised last
    layers of the reference, which would otherwise make every output constant), 1 +- 0.1 for norm scales, small biases."""
    g = torch.Generator().manual_seed(1000 + seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.dim() >= 2:
                fan_in = p[0].numel()
                v = torch.randn(p.shape, generator=g) * math.sqrt(1.0 / fan_in)
            elif name.endswith('weight'):
                v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
            else:
                v = 0.05 * torch.randn(p.shape, generator=g)
            p.copy_(v.to(p.device))
    return module


def _rotations(g, shape, sigma):
    a, b, c = (sigma * torch.randn(shape, generator=g) for _ in range(3))
    ca, sa, cb, sb, cc, sc = a.cos(), a.sin(), b.cos(), b.sin(), c.cos(), c.sin()
    R = torch.stack([cc * cb, cc * sb * sa - sc * ca, cc * sb * ca + sc * sa,
                     sc * cb, sc * sb * sa + cc * ca, sc * sb * ca - cc * sa,
                     -sb, cb * sa, cb * ca], dim=-1)
    return R.view(*shape, 3, 3)


def eyenet_batch(B, T, size=128, seed=0):
    g = torch.Generator().manual_seed(2000 + seed)
    b = {}
    for side in ('left', 'right'):
        b[side + '_eye_patch'] = torch.rand((B, T, 3, size, size), generator=g) * 2 - 1
        b[side + '_h'] = 0.1 * torch.randn((B, T, 2), generator=g)
        b[side + '_g_tobii'] = 0.2 * torch.randn((B, T, 2), generator=g)
        b[side + '_p'] = 2 + 3 * torch.rand((B, T), generator=g)
        b[side + '_g_tobii_validity'] = torch.ones((B, T), dtype=torch.bool)
        b[side + '_p_validity'] = torch.ones((B, T), dtype=torch.bool)
    return b


def eve_batch(B, T, seed=0, with_screen=True):
    """eyenet_batch + camera / screen geometry (a camera above a 1920 x 1080 px, 0.288 mm/px screen, user ~600 mm away),
    PoG labels, timestamps and 72 x 128 screen frames."""
    b = eyenet_batch(B, T, seed=seed)
    g = torch.Generator().manual_seed(3000 + seed)
    mpp = 0.288
    Rc = _rotations(g, (B,), 0.03)
    c = torch.stack([276.5 + 5 * torch.randn(B, generator=g), -12 + 2 * torch.randn(B, generator=g), torch.randn(B, generator=g)], dim=-1)
    cam = torch.zeros(B, 4, 4); cam[:, 3, 3] = 1
    cam[:, :3, :3] = Rc
    cam[:, :3, 3] = -torch.einsum('bij,bj->bi', Rc, c)
    inv = torch.zeros(B, 4, 4); inv[:, 3, 3] = 1
    inv[:, :3, :3] = Rc.transpose(1, 2)
    inv[:, :3, 3] = c
    rep = lambda t: t.unsqueeze(1).expand(B, T, *t.shape[1:]).contiguous()
    b['camera_transformation'], b['inv_camera_transformation'] = rep(cam), rep(inv)
    b['millimeters_per_pixel'] = torch.full((B, T, 2), mpp)
    b['pixels_per_millimeter'] = torch.full((B, T, 2), 1.0 / mpp)
    head = torch.tensor([0.0, 165.0, 600.0]) + torch.tensor([15.0, 10.0, 25.0]) * torch.randn((B, 1, 3), generator=g)
    head = head + torch.cumsum(1.5 * torch.randn((B, T, 3), generator=g), dim=1)
    for side, dx in (('left', 30.0), ('right', -30.0)):
        b[side + '_o'] = head + torch.tensor([dx, 0.0, 0.0]) + 0.5 * torch.randn((B, T, 3), generator=g)
        b[side + '_o_validity'] = torch.ones((B, T), dtype=torch.bool)
    R = _rotations(g, (B, T), 0.08)
    for k in ('left_R', 'right_R', 'head_R'):
        b[k] = R.clone()
    for side in ('left', 'right'):
        b[side + '_PoG_tobii'] = torch.stack([100 + 1720 * torch.rand((B, T), generator=g), 80 + 920 * torch.rand((B, T), generator=g)], dim=-1)
        b[side + '_PoG_tobii_validity'] = torch.ones((B, T), dtype=torch.bool)
    b['timestamps'] = 1 + torch.arange(T, dtype=torch.int64).unsqueeze(0) * 100000000 + torch.randint(0, 1000000, (B, T), generator=g)
    if with_screen:
        b['screen_frame'] = torch.rand((B, T, 3, 72, 128), generator=g)
    return b


def refinenet_batch(B, T, seed=0):
    """Heat-map pairs around a smooth on-screen trajectory + screen frames (RefineNet alone)."""
    g = torch.Generator().manual_seed(4000 + seed)
    H, W = 72, 128
    ys = torch.arange(H).view(1, 1, H, 1).float()
    xs = torch.arange(W).view(1, 1, 1, W).float()
    cx = torch.cumsum(4 * torch.randn((B, T), generator=g), dim=1) + 30 + 68 * torch.rand((B, 1), generator=g)
    cy = torch.cumsum(3 * torch.randn((B, T), generator=g), dim=1) + 20 + 32 * torch.rand((B, 1), generator=g)

    def maps(cx, cy, sigma):
        d2 = (xs - cx[..., None, None]) ** 2 + (ys - cy[..., None, None]) ** 2
        return (torch.exp(-d2 / (2.0 * sigma ** 2)) + 1e-8).unsqueeze(2)
    return {'heatmap_initial': maps(cx + 3 * torch.randn((B, T), generator=g), cy + 3 * torch.randn((B, T), generator=g), 10.0),
            'heatmap_final_gt': maps(cx, cy, 5.0), 'validity': torch.ones((B, T), dtype=torch.bool),
            'screen_frame': torch.rand((B, T, 3, H, W), generator=g)}
