#!/usr/bin/env python
"""Golden vectors for the OTHER recurrent-stage variants of the reference EyeNet (src/models/eye_net.py:58-78:
RNNCell / LSTMCell / stacked cells / static_fc), produced by running the reference's own EyeNet.forward per step.

Run here only (needs /root/reference):   python tests/golden/make_golden_variants.py
Writes tests/golden/eyenet_variants.npz (numbers only; inputs and weights come from oracle/detweights.py seeds).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from oracle import detweights  # noqa: E402

VARIANTS = {
    'RNN1': dict(eye_net_use_rnn=True, eye_net_rnn_type='RNN', eye_net_rnn_num_cells=1),
    'LSTM1': dict(eye_net_use_rnn=True, eye_net_rnn_type='LSTM', eye_net_rnn_num_cells=1),
    'GRU2': dict(eye_net_use_rnn=True, eye_net_rnn_type='GRU', eye_net_rnn_num_cells=2),
    'LSTM2': dict(eye_net_use_rnn=True, eye_net_rnn_type='LSTM', eye_net_rnn_num_cells=2),
    'STATIC': dict(eye_net_use_rnn=False),
}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    config = mg.import_reference()
    from models.eye_net import EyeNet
    config.import_json(os.path.join(mg.REF_SRC, 'configs', 'eye_net.json'))
    B, T = 2, 3
    batch = detweights.eyenet_batch(B, T, seed=5, invalid_fraction=0.2)
    fix = {'B': B, 'T': T, 'seed': 5, 'invalid_fraction': 0.2}
    for name, over in VARIANTS.items():
        for k, v in over.items():
            config.override(k, v)
        net = detweights.fill_module(EyeNet(), seed=0)
        steps, prev = [], None
        with torch.no_grad():
            for t in range(T):
                si = {k: v[:, t] for k, v in batch.items()}
                so = {}
                net(si, so, side='left', previous_output_dict=prev)
                net(si, so, side='right', previous_output_dict=prev)
                steps.append(so)
                prev = so
        for k in steps[0]:
            if isinstance(steps[0][k], tuple):
                for j in range(len(steps[0][k])):
                    fix['%s/%s/%d' % (name, k, j)] = torch.stack([s[k][j] for s in steps], 1).numpy()
            else:
                fix['%s/%s' % (name, k)] = torch.stack([s[k] for s in steps], 1).numpy()
    np.savez_compressed(os.path.join(mg.OUT, 'eyenet_variants.npz'), **fix)
    print('eyenet_variants.npz', sorted(fix)[:8], '...', len(fix), 'arrays')


if __name__ == '__main__':
    main()
