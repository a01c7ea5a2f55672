"""Oracle: RefineNet (72x128 pre-activation U-Net with a conv-RNN bottleneck).

TEST INFRASTRUCTURE.  Restates /root/reference/src/models/refine_net.py:35-255
and the conv-RNN cells of /root/reference/src/models/common.py:331-415:

  PreActBlock          refine_net.py:35-67   IN(affine)->act->conv3x3 twice, 1x1 pre-act skip if Cin != Cout
  Level                refine_net.py:70-129  encoder blocks -> adaptive max-pool -> inner -> bilinear up
                                             (align_corners=False) -> cat[up, enc] -> decoder block (LeakyReLU)
  Bottleneck           refine_net.py:132-176 cell choice by config; ONLY a non-tuple state replaces the
                                             features (:168-174), so a CLSTM's output is computed, stored, unused
  RefineNet            refine_net.py:179-255 level table :188-212, init :226-235, forward :237-255
  CRNN / CLSTM / CGRU  common.py:331-352 / :355-385 / :388-415

Module attribute names equal the reference's so state_dict keys match.
"""
import torch
from torch import nn
from torch.nn import functional as F


class CRNNCell(nn.Module):
    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.cell = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size=3, padding=1)

    def forward(self, x, previous_states=None):
        h = x.new_zeros(x.shape[0], self.hidden_size, *x.shape[2:]) \
            if previous_states is None else previous_states
        return torch.tanh(self.cell(torch.cat([x, h], dim=1)))


class CLSTMCell(nn.Module):
    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.gates = nn.Conv2d(input_size + hidden_size, 4 * hidden_size, kernel_size=3, padding=1)

    def forward(self, x, previous_states=None):
        if previous_states is None:
            h = x.new_zeros(x.shape[0], self.hidden_size, *x.shape[2:])
            c = torch.zeros_like(h)
        else:
            h, c = previous_states
        i, f, o, g = self.gates(torch.cat([x, h], dim=1)).chunk(4, 1)  # common.py:376
        c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        return torch.sigmoid(o) * torch.tanh(c_new), c_new


class CGRUCell(nn.Module):
    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.gates_1 = nn.Conv2d(input_size + hidden_size, 2 * hidden_size, kernel_size=3, padding=1)
        self.gate_2 = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size=3, padding=1)

    def forward(self, x, previous_states=None):
        h = x.new_zeros(x.shape[0], self.hidden_size, *x.shape[2:]) \
            if previous_states is None else previous_states
        r, u = torch.sigmoid(self.gates_1(torch.cat([x, h], dim=1))).chunk(2, 1)  # common.py:409-410
        o = torch.tanh(self.gate_2(torch.cat([r * h, x], dim=1)))                  # common.py:412 (order flips)
        return (1. - u) * o + u * h


class BasicBlock(nn.Module):  # pre-activation block; name kept for key parity
    def __init__(self, in_shape, out_shape, act_func=nn.ReLU):
        super().__init__()
        ic, oc = in_shape[0], out_shape[0]
        assert tuple(in_shape[1:]) == tuple(out_shape[1:])
        self.layers = nn.Sequential(
            nn.InstanceNorm2d(ic, affine=True), act_func(inplace=True),
            nn.Conv2d(ic, oc, kernel_size=3, stride=1, padding=1),
            nn.InstanceNorm2d(oc, affine=True), act_func(inplace=True),
            nn.Conv2d(oc, oc, kernel_size=3, stride=1, padding=1),
        )
        self.skip_layer = None
        if ic != oc:
            self.skip_layer = nn.Sequential(
                nn.InstanceNorm2d(ic, affine=True), act_func(inplace=True),
                nn.Conv2d(ic, oc, kernel_size=1, stride=1))

    def forward(self, x):
        skip = x if self.skip_layer is None else self.skip_layer(x)
        return self.layers(x) + skip


class WrapEncoderDecoder(nn.Module):
    def __init__(self, in_shape, out_shape, module_to_wrap, add_skip_connection=False,
                 num_encoder_blocks=1, num_decoder_blocks=1):
        super().__init__()
        ic, ih, iw = in_shape
        oc, oh, ow = out_shape
        assert ih == oh and iw == ow
        self.in_shape, self.out_shape = in_shape, out_shape
        b_ic, bh, bw = module_to_wrap.in_shape
        b_oc = module_to_wrap.out_shape[0]
        self.add_skip_connection = add_skip_connection
        self.encoder_blocks = nn.ModuleList(
            [BasicBlock([ic, ih, iw], [b_ic, ih, iw])] +
            [BasicBlock([b_ic, ih, iw], [b_ic, ih, iw]) for _ in range(num_encoder_blocks - 1)])
        self.downsample = nn.AdaptiveMaxPool2d([bh, bw]) if (ih, iw) != (bh, bw) else None
        self.between_module = module_to_wrap
        self.upsample = (nn.Upsample(size=[oh, ow], mode='bilinear', align_corners=False)
                         if (bh, bw) != (oh, ow) else None)
        dec_in = b_oc + (b_ic if add_skip_connection else 0)
        self.decoder_blocks = nn.ModuleList(
            [BasicBlock([dec_in, oh, ow], [oc, oh, ow], nn.LeakyReLU)] +
            [BasicBlock([oc, oh, ow], [oc, oh, ow], nn.LeakyReLU)
             for _ in range(num_decoder_blocks - 1)])

    def forward(self, x, output_dict, previous_output_dict):
        for blk in self.encoder_blocks:
            x = blk(x)
        enc = x
        if self.downsample is not None:
            x = self.downsample(x)
        x = self.between_module(x, output_dict, previous_output_dict)
        if self.upsample is not None:
            x = self.upsample(x)
        if self.add_skip_connection:
            x = torch.cat([x, enc], dim=1)
        for blk in self.decoder_blocks:
            x = blk(x)
        return x


class Bottleneck(nn.Module):
    def __init__(self, tensor_shape, config):
        super().__init__()
        self.config = config
        self.in_shape = self.out_shape = tensor_shape
        if config.refine_net_use_rnn:
            kinds = {'CRNN': CRNNCell, 'CLSTM': CLSTMCell, 'CGRU': CGRUCell}
            cells = []
            for _ in range(config.refine_net_rnn_num_cells):
                if config.refine_net_rnn_type in kinds:  # unknown type => silently no cell (:143-151)
                    cells.append(kinds[config.refine_net_rnn_type](
                        input_size=config.refine_net_num_features,
                        hidden_size=config.refine_net_num_features))
            self.rnn_cells = nn.ModuleList(cells)

    def forward(self, x, output_dict, previous_output_dict):
        if self.config.refine_net_use_rnn:
            for i, cell in enumerate(self.rnn_cells):
                key = 'refinenet_rnn_states_%d' % i
                prev = None if previous_output_dict is None else previous_output_dict[key]
                states = cell(x, prev)
                output_dict[key] = states
                if not isinstance(states, tuple):   # refine_net.py:168-174
                    x = states
        return x


class RefineNet(nn.Module):
    # (channels, H, W, encoder blocks) from the innermost wrapped level outwards (refine_net.py:189-212)
    LEVELS = [(256, 5, 8, 2), (128, 9, 16, 2), (64, 18, 32, 2), (32, 36, 64, 2), (16, 72, 128, 1)]

    def __init__(self, config):
        super().__init__()
        self.config = config
        in_c = 4 if config.load_screen_content else 1
        skip = config.refine_net_use_skip_connections
        wrapped = Bottleneck((config.refine_net_num_features, 5, 8), config)
        for c, h, w, n_enc in self.LEVELS:
            wrapped = WrapEncoderDecoder([c, h, w], [c, h, w], wrapped, add_skip_connection=skip,
                                         num_encoder_blocks=n_enc)
        self.initial = nn.Sequential(
            nn.Conv2d(in_c, 16, kernel_size=3, padding=1), nn.InstanceNorm2d(16, affine=True),
            nn.ReLU(inplace=True), nn.Conv2d(16, 16, kernel_size=3, padding=1))
        self.network = wrapped
        self.final = nn.Sequential(
            nn.Conv2d(16, 16, kernel_size=3, padding=1), nn.LeakyReLU(inplace=True),
            nn.Conv2d(16, 1, kernel_size=1), nn.Sigmoid())
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.InstanceNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        nn.init.zeros_(self.final[-2].weight)

    def forward(self, input_dict, output_dict, previous_output_dict=None):
        cfg = self.config
        heat = F.interpolate(output_dict['heatmap_initial'],
                             (cfg.screen_size[1], cfg.screen_size[0]),
                             mode='bilinear', align_corners=False)
        x = torch.cat([input_dict['screen_frame'], heat], dim=1) if cfg.load_screen_content else heat
        x = self.network(self.initial(x), output_dict, previous_output_dict)
        output_dict['heatmap_final'] = self.final(x)
