"""Deterministic, name-keyed weights and synthetic inputs shared by the golden generator and the tests.

TEST INFRASTRUCTURE.  Weight files (45 MB + 21 MB) are too big to commit, so both the
reference (in make_golden.py) and the checked implementations are loaded from this
generator: for every state_dict key, a numpy PCG64 stream seeded by crc32(key) ^ seed.
The two zero-initialised output layers (eye_net.py:96 fc_to_gaze.2.weight,
refine_net.py:235 final.2.weight) get NON-zero values, otherwise g_initial == 0 and
heatmap_final == 0.5 identically and parity would be vacuous (SURVEY 7 "vacuous-parity traps").
"""
import zlib

import numpy as np
import torch


def _rng(name, seed):
    return np.random.Generator(np.random.PCG64((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0xFFFFFFFF))


def tensor_for(name, shape, seed=0):
    g = _rng(name, seed)
    shape = tuple(shape)
    leaf = name.rsplit('.', 1)[-1]
    if leaf.startswith('bias'):
        v = 0.05 * g.standard_normal(shape)
    elif len(shape) == 1:                       # InstanceNorm affine scale
        v = 1.0 + 0.1 * g.standard_normal(shape)
    else:
        fan_in = int(np.prod(shape[1:]))
        v = g.standard_normal(shape) * np.sqrt(2.0 / fan_in)
        if name.endswith('fc_to_gaze.2.weight') or name.endswith('final.2.weight'):
            v = v * 0.5
    return torch.from_numpy(v.astype(np.float32))


def fill_module(module, seed=0):
    """Overwrite every parameter/buffer of `module` in place, keyed by state_dict name."""
    with torch.no_grad():
        for name, t in module.state_dict().items():
            t.copy_(tensor_for(name, t.shape, seed))
    return module


def eyenet_batch(B, T, size=128, seed=0, invalid_fraction=0.0):
    """Synthetic EyeNet clip batch, schema of /root/reference/src/datasources/eve_sequences.py:215-299."""
    g = np.random.Generator(np.random.PCG64(1000 + seed))
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    batch = {}
    for side in ('left', 'right'):
        # low-frequency structure + noise so InstanceNorm statistics are non-trivial
        base = g.uniform(-1, 1, size=(B, T, 3, size // 8, size // 8))
        img = np.kron(base, np.ones((8, 8))) * 0.6 + 0.4 * g.uniform(-1, 1, size=(B, T, 3, size, size))
        batch[side + '_eye_patch'] = f32(np.clip(img, -1, 1))
        batch[side + '_h'] = f32(g.normal(0, 0.1, size=(B, T, 2)))
        batch[side + '_g_tobii'] = f32(g.normal(0, 0.2, size=(B, T, 2)))
        batch[side + '_p'] = f32(g.uniform(2, 5, size=(B, T)))
        for k in ('_g_tobii', '_p'):
            valid = g.uniform(size=(B, T)) >= invalid_fraction
            batch[side + k + '_validity'] = torch.from_numpy(valid)
    return batch


def refinenet_batch(B, T, seed=0, with_screen=True, invalid_fraction=0.0):
    """Synthetic RefineNet inputs: Gaussian heat-maps (models/common.py:226-239 form) around a
    smooth on-screen trajectory, screen frames in [0,1] (eve_sequences.py:205-211)."""
    g = np.random.Generator(np.random.PCG64(2000 + seed))
    H, W = 72, 128
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    cx = np.cumsum(g.normal(0, 4, size=(B, T)), axis=1) + g.uniform(30, 98, size=(B, 1))
    cy = np.cumsum(g.normal(0, 3, size=(B, T)), axis=1) + g.uniform(20, 52, size=(B, 1))

    def maps(cx, cy, sigma):
        d2 = (xs[None, None] - cx[..., None, None]) ** 2 + (ys[None, None] - cy[..., None, None]) ** 2
        return (np.exp(-d2 / (2.0 * sigma ** 2)) + 1e-8)[:, :, None]

    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    out = {
        'heatmap_initial': f32(maps(cx + g.normal(0, 3, size=(B, T)), cy + g.normal(0, 3, size=(B, T)), 10.0 * 128 / 1920 * 15)),
        'heatmap_final_gt': f32(maps(cx, cy, 5.0)),
        'validity': torch.from_numpy(g.uniform(size=(B, T)) >= invalid_fraction),
    }
    if with_screen:
        base = g.uniform(0, 1, size=(B, T, 3, H // 4, W // 4))
        out['screen_frame'] = f32(np.clip(np.kron(base, np.ones((4, 4))) * 0.7 +
                                          0.3 * g.uniform(0, 1, size=(B, T, 3, H, W)), 0, 1))
    return out


def _small_rotations(g, shape, sigma):
    """Rotation matrices R = Rz(c) Ry(b) Rx(a) with a, b, c ~ N(0, sigma^2)."""
    a, b, c = (g.normal(0, sigma, size=shape) for _ in range(3))
    ca, sa, cb, sb, cc, sc = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    R = np.empty(shape + (3, 3))
    R[..., 0, 0], R[..., 0, 1], R[..., 0, 2] = cc * cb, cc * sb * sa - sc * ca, cc * sb * ca + sc * sa
    R[..., 1, 0], R[..., 1, 1], R[..., 1, 2] = sc * cb, sc * sb * sa + cc * ca, sc * sb * ca - cc * sa
    R[..., 2, 0], R[..., 2, 1], R[..., 2, 2] = -sb, cb * sa, cb * ca
    return R


def eve_batch(B, T, seed=0, invalid_fraction=0.0, with_screen=True):
    """Full synthetic clip batch for the EVE sequence harness: eyenet_batch plus camera / screen geometry, PoG labels,
    timestamps and screen frames -- the schema of /root/reference/src/datasources/eve_sequences.py:215-299 with the
    value ranges of SURVEY.md 8(d).  The camera sits above the top edge of a 1920x1080 px (0.288 mm/px) screen and
    looks at a user ~600 mm away, so gaze rays of a few tenths of a radian land on the screen."""
    batch = eyenet_batch(B, T, seed=seed, invalid_fraction=invalid_fraction)
    g = np.random.Generator(np.random.PCG64(3000 + seed))
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    mpp = 0.288
    # screen -> camera: x_cam = Rc (x_screen - c);  camera origin c (mm, screen frame), small tilt Rc
    Rc = _small_rotations(g, (B,), 0.03)
    c = np.stack([g.normal(276.5, 5, size=B), g.normal(-12, 2, size=B), g.normal(0, 1, size=B)], axis=-1)
    cam = np.zeros((B, 4, 4)); cam[:, 3, 3] = 1
    cam[:, :3, :3] = Rc
    cam[:, :3, 3] = -np.einsum('bij,bj->bi', Rc, c)
    inv = np.zeros((B, 4, 4)); inv[:, 3, 3] = 1
    inv[:, :3, :3] = np.transpose(Rc, (0, 2, 1))
    inv[:, :3, 3] = c
    rep = lambda a: np.repeat(a[:, None], T, axis=1)
    batch['camera_transformation'] = f32(rep(cam))
    batch['inv_camera_transformation'] = f32(rep(inv))
    batch['millimeters_per_pixel'] = f32(np.full((B, T, 2), mpp))
    batch['pixels_per_millimeter'] = f32(np.full((B, T, 2), 1.0 / mpp))
    # head / eye origins in the camera frame: slow drift around (0, 165, 600) mm, eyes 60 mm apart
    head = np.stack([g.normal(0, 15, size=(B, 1)), g.normal(165, 10, size=(B, 1)), g.normal(600, 25, size=(B, 1))], axis=-1)
    head = head + np.cumsum(g.normal(0, 1.5, size=(B, T, 3)), axis=1)
    for side, dx in (('left', 30.0), ('right', -30.0)):
        batch[side + '_o'] = f32(head + np.array([dx, 0.0, 0.0]) + g.normal(0, 0.5, size=(B, T, 3)))
    batch['left_o_validity'] = torch.from_numpy(g.uniform(size=(B, T)) >= invalid_fraction)
    batch['right_o_validity'] = batch['left_o_validity'].clone()
    R = _small_rotations(g, (B, T), 0.08)
    for k in ('left_R', 'right_R', 'head_R'):           # by definition the same matrix (eve.py:160)
        batch[k] = f32(R)
    for side in ('left', 'right'):
        px = np.stack([g.uniform(100, 1820, size=(B, T)), g.uniform(80, 1000, size=(B, T))], axis=-1)
        batch[side + '_PoG_tobii'] = f32(px)
        batch[side + '_PoG_tobii_validity'] = torch.from_numpy(g.uniform(size=(B, T)) >= invalid_fraction)
    ts = 1 + np.arange(T, dtype=np.int64)[None, :] * 100000000 + g.integers(0, 1000000, size=(B, T))
    batch['timestamps'] = torch.from_numpy(ts.astype(np.int64))
    if with_screen:
        H, W = 72, 128
        base = g.uniform(0, 1, size=(B, T, 3, H // 4, W // 4))
        batch['screen_frame'] = f32(np.clip(np.kron(base, np.ones((4, 4))) * 0.7 +
                                            0.3 * g.uniform(0, 1, size=(B, T, 3, H, W)), 0, 1))
    return batch
