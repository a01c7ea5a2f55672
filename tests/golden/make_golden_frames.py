#!/usr/bin/env python
"""Golden fixture for the frame normalisation: calls the REFERENCE's own
EVESequencesBase.preprocess_frames / preprocess_screen_frames (src/datasources/eve_sequences.py:196-211) on every uint8
value and on a random frame stack.  Run in the build container only:  python tests/golden/make_golden_frames.py

The datasource module imports cv2, h5py and ffmpeg at the top (video decoding / label files; absent from this image).
They are replaced by EMPTY module objects for the import only: the two methods called here use numpy and nothing else.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import OUT, import_reference  # noqa: E402


def main():
    import_reference()
    for name in ('cv2', 'h5py', 'ffmpeg'):
        sys.modules.setdefault(name, types.ModuleType(name))
    from datasources.eve_sequences import EVESequencesBase
    g = np.random.Generator(np.random.PCG64(77))
    ramp = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1).repeat(3, axis=3)           # every uint8 value
    frames = g.integers(0, 256, size=(3, 12, 20, 3), dtype=np.uint8)
    fix = {'ramp': ramp, 'frames': frames}
    for key, arr in (('ramp', ramp), ('frames', frames)):
        fix[key + '_eye'] = EVESequencesBase.preprocess_frames(None, arr.copy())
        fix[key + '_screen'] = EVESequencesBase.preprocess_screen_frames(None, arr.copy())
    np.savez_compressed(os.path.join(OUT, 'frames.npz'), **fix)
    print('frames.npz', {k: (v.shape, str(v.dtype)) for k, v in fix.items()})


if __name__ == '__main__':
    main()
