"""Oracle: EyeNet (per-eye, per-frame encoder + recurrent cell + gaze/pupil heads).

TEST INFRASTRUCTURE.  Restates /root/reference/src/models/eye_net.py:37-150:
construction :38-96, forward :98-150 (patch -> ResNet-18(IN) -> cat head pose
-> fc_common(SELU) -> GRU/LSTM/RNN cell or static_fc -> gaze head
(pi/2 * tanh, last Linear bias-free and zero-initialised :84,:96) and pupil
head (ReLU, reshape(-1) :146)).  Same dict-in / dict-out contract and the same
state_dict keys.
"""
import math

import torch
from torch import nn

from .resnet_in import BasicBlock, ResNet

half_pi = 0.5 * math.pi


class EyeNet(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        nf = (config.eye_net_rnn_num_features if config.eye_net_use_rnn
              else config.eye_net_static_num_features)
        self.cnn_layers = ResNet(block=BasicBlock, layers=[2, 2, 2, 2], num_classes=nf,
                                 norm_layer=nn.InstanceNorm2d)
        self.fc_common = nn.Sequential(
            nn.Linear(nf + (2 if config.eye_net_use_head_pose_input else 0), nf),
            nn.SELU(inplace=True),
            nn.Linear(nf, nf),
        )
        if config.eye_net_use_rnn:
            cells = []
            for _ in range(config.eye_net_rnn_num_cells):
                kind = config.eye_net_rnn_type
                n = config.eye_net_rnn_num_features
                if kind == 'RNN':
                    cells.append(nn.RNNCell(input_size=n, hidden_size=n))
                elif kind == 'LSTM':
                    cells.append(nn.LSTMCell(input_size=n, hidden_size=n))
                elif kind == 'GRU':
                    cells.append(nn.GRUCell(input_size=n, hidden_size=n))
                else:
                    raise ValueError('Unknown RNN type for EyeNet: %s' % kind)
            self.rnn_cells = nn.ModuleList(cells)
        else:
            self.static_fc = nn.Sequential(nn.Linear(nf, nf), nn.SELU(inplace=True))
        self.fc_to_gaze = nn.Sequential(
            nn.Linear(nf, nf), nn.SELU(inplace=True), nn.Linear(nf, 2, bias=False), nn.Tanh())
        self.fc_to_pupil = nn.Sequential(
            nn.Linear(nf, nf), nn.SELU(inplace=True), nn.Linear(nf, 1), nn.ReLU(inplace=True))
        nn.init.zeros_(self.fc_to_gaze[-2].weight)

    def forward(self, input_dict, output_dict, side, previous_output_dict=None):
        cfg = self.config
        key = side + '_eye_patch'
        image = output_dict[key] if key in output_dict else input_dict[key]
        feats = self.cnn_layers(image)
        if cfg.eye_net_use_head_pose_input:
            feats = torch.cat([feats, input_dict[side + '_h']], dim=1)
        feats = self.fc_common(feats)
        if cfg.eye_net_use_rnn:
            for i, cell in enumerate(self.rnn_cells):
                skey = side + '_eye_rnn_states_%d' % i
                prev = None if previous_output_dict is None else previous_output_dict[skey]
                states = cell(feats, prev)
                output_dict[skey] = states
                feats = states[0] if isinstance(states, tuple) else states
        else:
            feats = self.static_fc(feats)
        gaze = half_pi * self.fc_to_gaze(feats)
        pupil = self.fc_to_pupil(feats)
        output_dict[side + '_g_initial'] = gaze
        output_dict[side + '_pupil_size'] = pupil.reshape(-1)
        if cfg.eye_net_frozen:
            output_dict[side + '_g_initial'] = output_dict[side + '_g_initial'].detach()
