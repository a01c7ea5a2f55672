"""Oracle: normalisation of decoded video frames.

TEST INFRASTRUCTURE.  Restates /root/reference/src/datasources/eve_sequences.py:196-211
  preprocess_frames         N x H x W x C uint8 -> N x C x H x W float32, x * (2/255) - 1   (eye patches)
  preprocess_screen_frames  N x H x W x C uint8 -> N x C x H x W float32, x * (1/255)       (screen content)
(in-place float32 numpy arithmetic: one rounded multiply by the float32-cast scalar, one rounded subtract).
Pinned by tests/golden/frames.npz, produced by calling the reference's own two methods (make_golden_frames.py).
"""
import numpy as np


def preprocess_frames(frames):
    out = np.transpose(frames, [0, 3, 1, 2]).astype(np.float32)
    out *= 2.0 / 255.0
    out -= 1.0
    return out


def preprocess_screen_frames(frames):
    out = np.transpose(frames, [0, 3, 1, 2]).astype(np.float32)
    out *= 1.0 / 255.0
    return out
