"""RefineNet drop-in: 72x128 pre-activation U-Net with a conv-RNN bottleneck refining the point-of-gaze
heat-map, computed by the gfx950 HIP kernels of libeve_hip.so.

Mirrors /root/reference/src/models/refine_net.py:
  * `RefineNet()` (:179-235) reads the config singleton; sub-module names `initial`, `network`
    (`encoder_blocks`, `between_module`, `decoder_blocks`, `layers`, `skip_layer`, `rnn_cells.0.
    {gates_1,gate_2|gates|cell}`), `final` so state_dict keys equal the reference's;
    Kaiming fan_out init, IN weight 1 / bias 0, zero last conv weight (:226-235);
  * `forward(input_dict, output_dict, previous_output_dict=None) -> None` (:237-255): reads
    `output_dict['heatmap_initial']` (+ `input_dict['screen_frame']`), writes `heatmap_final` and
    `refinenet_rnn_states_0` (NCHW float at the boundary, like the reference);
  * Bottleneck quirks kept (:132-176): unknown rnn type => no cell; a tuple state (CLSTM) is stored but
    its output is NOT used downstream, so its weights get no gradient.

`forward_sequence` folds all T frames into the image batch for the encoder and decoder (InstanceNorm
is per-sample) and runs only the 64x5x8 conv-RNN sequentially.

nn.Conv2d / nn.InstanceNorm2d objects are PARAMETER HOLDERS; their ATen forward is never called.
"""

import torch
from torch import nn

from . import ops
from .config import get_config
from .eye_net import default_compute_dtype
from .kernels import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_TANH, HALF_DTYPES, default_kernels, dispatch_flag, pad_channels
from .ops import PackedWeight


class CRNNCell(nn.Module):       # holder, common.py:331-339
    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.cell = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size=3, padding=1)


class CLSTMCell(nn.Module):      # holder, common.py:355-363
    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.gates = nn.Conv2d(input_size + hidden_size, 4 * hidden_size, kernel_size=3, padding=1)


class CGRUCell(nn.Module):       # holder, common.py:388-398
    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.gates_1 = nn.Conv2d(input_size + hidden_size, 2 * hidden_size, kernel_size=3, padding=1)
        self.gate_2 = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size=3, padding=1)


class BasicBlock(nn.Module):     # holder, refine_net.py:35-62
    def __init__(self, in_shape, out_shape, act_func=nn.ReLU):
        super().__init__()
        ic, oc = in_shape[0], out_shape[0]
        assert tuple(in_shape[1:]) == tuple(out_shape[1:])
        self.act = ACT_LEAKY if act_func is nn.LeakyReLU else ACT_RELU
        self.layers = nn.Sequential(
            nn.InstanceNorm2d(ic, affine=True), act_func(inplace=True),
            nn.Conv2d(ic, oc, kernel_size=3, stride=1, padding=1),
            nn.InstanceNorm2d(oc, affine=True), act_func(inplace=True),
            nn.Conv2d(oc, oc, kernel_size=3, stride=1, padding=1))
        self.skip_layer = None
        if ic != oc:
            self.skip_layer = nn.Sequential(nn.InstanceNorm2d(ic, affine=True), act_func(inplace=True),
                                            nn.Conv2d(ic, oc, kernel_size=1, stride=1))


class WrapEncoderDecoder(nn.Module):   # holder, refine_net.py:70-113
    def __init__(self, in_shape, out_shape, module_to_wrap, add_skip_connection=False,
                 num_encoder_blocks=1, num_decoder_blocks=1):
        super().__init__()
        ic, ih, iw = in_shape
        oc, oh, ow = out_shape
        assert ih == oh and iw == ow
        self.in_shape, self.out_shape = in_shape, out_shape
        b_ic, bh, bw = module_to_wrap.in_shape
        b_oc = module_to_wrap.out_shape[0]
        self.inner_hw = (bh, bw)
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
            [BasicBlock([oc, oh, ow], [oc, oh, ow], nn.LeakyReLU) for _ in range(num_decoder_blocks - 1)])


class Bottleneck(nn.Module):     # holder, refine_net.py:132-152
    def __init__(self, tensor_shape, config):
        super().__init__()
        self.in_shape = self.out_shape = tensor_shape
        if config.refine_net_use_rnn:
            kinds = {'CRNN': CRNNCell, 'CLSTM': CLSTMCell, 'CGRU': CGRUCell}
            cells = []
            for _ in range(config.refine_net_rnn_num_cells):
                if config.refine_net_rnn_type in kinds:
                    cells.append(kinds[config.refine_net_rnn_type](
                        input_size=config.refine_net_num_features,
                        hidden_size=config.refine_net_num_features))
            self.rnn_cells = nn.ModuleList(cells)


class RefineNet(nn.Module):
    LEVELS = [(256, 5, 8, 2), (128, 9, 16, 2), (64, 18, 32, 2), (32, 36, 64, 2), (16, 72, 128, 1)]

    def __init__(self):
        super(RefineNet, self).__init__()
        config = get_config()
        self.config = config
        self.compute_dtype = default_compute_dtype()
        self.in_c = 4 if config.load_screen_content else 1
        skip = config.refine_net_use_skip_connections
        wrapped = Bottleneck((config.refine_net_num_features, 5, 8), config)
        for c, h, w, n_enc in self.LEVELS:
            wrapped = WrapEncoderDecoder([c, h, w], [c, h, w], wrapped, add_skip_connection=skip,
                                         num_encoder_blocks=n_enc)
        self.initial = nn.Sequential(
            nn.Conv2d(self.in_c, 16, kernel_size=3, padding=1), nn.InstanceNorm2d(16, affine=True),
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
        self._packs = None
        self._packs_key = None
        self._probe = None          # test hook: callable(name, NHWC tensor) -> tensor at every stage boundary

    def _tap(self, name, x):
        return x if self._probe is None else self._probe(name, x)

    # ------------------------------------------------------------------ packed weights
    def invalidate_packs(self):
        self._packs = None

    def _get_packs(self):
        dt = self.compute_dtype
        key = (dt,) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packs is not None and self._packs_key == key:
            return self._packs
        P = {}
        for name, m in self.named_modules():
            if isinstance(m, nn.Conv2d):
                cin_pad = pad_channels(m.in_channels, dt)
                cout_pad = pad_channels(m.out_channels, dt)
                P[name] = PackedWeight(m.weight, dt, cin_pad=cin_pad, cout_pad=cout_pad, defer=True)
        PackedWeight.pack_many(list(P.values()), dt)      # every filter bank of the network in one launch per 48 (was one each)
        self._packs, self._packs_key = P, key
        return P

    def _conv(self, x, name, m, P, act=ACT_NONE, acc=None, want_stats=False):
        return ops.conv2d(x, m.weight, m.bias, P[name], stride=1, pad=m.padding[0], act=act, acc=acc, want_stats=want_stats)

    # ------------------------------------------------------------------ blocks
    def _block(self, x, blk, prefix, P):
        """x: the block input, or a tuple of NHWC sources whose channel-concatenation is the block input (the decoder's
        torch.cat of refine_net.py:125-126, which is then never materialised)."""
        L, S = blk.layers, blk.skip_layer
        xs = x if isinstance(x, tuple) else (x,)
        # both heads of a block with a skip convolution normalise the same input: one statistics pass and one read per source,
        # each head written into its channel range (planes too large for the register-resident kernel; small ones keep it)
        two_heads = S is not None and all(t.shape[1] * t.shape[2] * t.shape[3] > 65536 for t in xs)
        if two_heads:
            a, skip = ops.instnorm_act2(xs, L[0].weight, L[0].bias, S[0].weight, S[0].bias, act=blk.act)
        else:
            x = xs[0] if len(xs) == 1 else torch.cat(xs, dim=-1)
            if S is None and torch.is_grad_enabled() and x.requires_grad and hasattr(ops, 'InstNormActSkipFn'):
                # identity skip: x feeds `layers` and the final add -- one node, the two gradients meet in its backward
                a, skip = ops.InstNormActSkipFn.apply(x, L[0].weight, L[0].bias, blk.act, 1e-5)
            else:
                a = ops.instnorm_act(x, L[0].weight, L[0].bias, act=blk.act)
                skip = x if S is None else ops.instnorm_act(x, S[0].weight, S[0].bias, act=blk.act)
        # the mid-block InstanceNorm's statistics come out of the convolution's own epilogue where its kernel walks whole
        # images (the row-streaming 3x3 kernel of the two outer levels): no statistics pass over the plane
        a, mr = self._conv(a, prefix + '.layers.2', L[2], P, want_stats=True)
        a = ops.instnorm_act(a, L[3].weight, L[3].bias, act=blk.act, stats=mr)
        a = self._conv(a, prefix + '.layers.5', L[5], P)
        if S is not None:           # layers(x) + skip_layer(x): the 1x1 convolution accumulates into the 3x3 branch's output
            return self._conv(skip, prefix + '.skip_layer.2', S[2], P, acc=a)
        return ops.add(a, skip)

    def _encode(self, x, P):
        """Runs `initial` and every level's encoder on the folded frame batch.
        Returns the bottleneck input and the per-level encoder outputs (outermost first)."""
        x = self._tap('input', x)
        x = self._conv(x, 'initial.0', self.initial[0], P)
        x = ops.instnorm_act(x, self.initial[1].weight, self.initial[1].bias, act=ACT_RELU)
        x = self._tap('initial', self._conv(x, 'initial.3', self.initial[3], P))
        skips, level, prefix, depth = [], self.network, 'network', 0
        while isinstance(level, WrapEncoderDecoder):
            for i, blk in enumerate(level.encoder_blocks):
                x = self._tap('enc%d.%d' % (depth, i), self._block(x, blk, '%s.encoder_blocks.%d' % (prefix, i), P))
            if level.downsample is not None and level.add_skip_connection and torch.is_grad_enabled() and x.requires_grad:
                # the level's output goes to the pool and to the decoder's skip: one node, their gradients meet in its backward
                pooled, x_skip = ops.PoolForkFn.apply(x, tuple(level.inner_hw))
                skips.append(x_skip)
                x = self._tap('pool%d' % depth, pooled)
            else:
                skips.append(x)
                if level.downsample is not None:
                    x = self._tap('pool%d' % depth, ops.AdaptiveMaxPoolFn.apply(x, tuple(level.inner_hw)))
            level, prefix, depth = level.between_module, prefix + '.between_module', depth + 1
        return x, skips, prefix

    def _decode(self, x, skips, P):
        levels, level, prefix = [], self.network, 'network'
        while isinstance(level, WrapEncoderDecoder):
            levels.append((level, prefix))
            level, prefix = level.between_module, prefix + '.between_module'
        for depth, (level, prefix), enc in zip(reversed(range(len(levels))), reversed(levels), reversed(skips)):
            if level.upsample is not None:
                x = self._tap('up%d' % depth, ops.BilinearFn.apply(x, (level.out_shape[1], level.out_shape[2])))
            if level.add_skip_connection:
                x = (x, enc)                                   # concatenated by the first decoder block's InstanceNorms
            for i, blk in enumerate(level.decoder_blocks):
                x = self._tap('dec%d.%d' % (depth, i), self._block(x, blk, '%s.decoder_blocks.%d' % (prefix, i), P))
        x = self._tap('final0', self._conv(x, 'final.0', self.final[0], P, act=ACT_LEAKY))
        # logits of the last 1x1 convolution (channel 0); the sigmoid is evaluated in float by the head kernel
        logits = self._tap('logits', self._conv(x, 'final.2', self.final[2], P))
        return ops.HeatmapHeadFn.apply(logits)                                 # [N, 1, H, W] float

    # ------------------------------------------------------------------ conv-RNN bottleneck, one step
    def _cell_step(self, x, state, cell, prefix, P):
        """x: [B,5,8,C] NHWC.  state: previous state (tensor, or (h, c) for CLSTM) or None.
        Returns (features for the decoder, new state)."""
        k = default_kernels()
        if isinstance(cell, CGRUCell):
            h = torch.zeros_like(x) if state is None else state
            g1 = self._conv(torch.cat([x, h], dim=-1), prefix + '.gates_1', cell.gates_1, P)
            ru, rh = ops.CGRUGates1Fn.apply(g1, h)
            g2 = self._conv(torch.cat([rh, x], dim=-1), prefix + '.gate_2', cell.gate_2, P)
            hnew = ops.CGRUGates2Fn.apply(g2, ru, h)
            return hnew, hnew
        if isinstance(cell, CRNNCell):
            h = torch.zeros_like(x) if state is None else state
            hnew = self._conv(torch.cat([x, h], dim=-1), prefix + '.cell', cell.cell, P, act=ACT_TANH)
            return hnew, hnew
        # CLSTM: state computed and stored, output dead (refine_net.py:168-174); forward-only kernels
        with torch.no_grad():
            if state is None:
                h, c = torch.zeros_like(x), torch.zeros_like(x)
            else:
                h, c = state
            gates = k.conv2d_fwd(torch.cat([x.detach(), h], dim=-1).contiguous(), P[prefix + '.gates'].ohwi,
                                 cell.gates.bias.detach().float().contiguous(), 1, 1)
            hn, cn = k.clstm_gates_fwd(gates, c.contiguous())
        return x, (hn, cn)

    def _bottleneck(self, x, states, prefix, P):
        """x: [B,5,8,C].  states: list (one per cell) of previous states or None.  -> (x, new states)"""
        bott = self.network
        while isinstance(bott, WrapEncoderDecoder):
            bott = bott.between_module
        new_states = []
        if self.config.refine_net_use_rnn:
            for i, cell in enumerate(bott.rnn_cells):
                prev = None if states is None else states[i]
                x, st = self._cell_step(x, prev, cell, '%s.rnn_cells.%d' % (prefix, i), P)
                new_states.append(st)
        return x, new_states

    # ------------------------------------------------------------------ boundary helpers
    def _input_nhwc(self, heatmap, screen):
        cfg, dt = self.config, self.compute_dtype
        H, W = cfg.screen_size[1], cfg.screen_size[0]
        if tuple(heatmap.shape[-2:]) != (H, W):       # F.interpolate(..., bilinear) of refine_net.py:240-243
            hm = ops.ToNHWCFn.apply(heatmap, dt, pad_channels(1, dt))
            hm = ops.BilinearFn.apply(hm, (H, W))
            heatmap = ops.FromNHWCFn.apply(hm, 1)
        x = torch.cat([screen, heatmap], dim=1) if cfg.load_screen_content else heatmap
        return ops.ToNHWCFn.apply(x, dt, pad_channels(x.shape[1], dt))

    def _state_in(self, st):
        dt = self.compute_dtype
        f = lambda t: ops.ToNHWCFn.apply(t, dt, pad_channels(t.shape[1], dt))
        return tuple(f(t) for t in st) if isinstance(st, tuple) else f(st)

    @staticmethod
    def _state_out(st):
        f = lambda t: ops.FromNHWCFn.apply(t, t.shape[-1])
        return tuple(f(t) for t in st) if isinstance(st, tuple) else f(st)

    # ------------------------------------------------------------------ reference per-step contract
    def forward(self, input_dict, output_dict, previous_output_dict=None):
        P = self._get_packs()
        screen = input_dict['screen_frame'] if self.config.load_screen_content else None
        x = self._input_nhwc(output_dict['heatmap_initial'], screen)
        x, skips, prefix = self._encode(x, P)
        states = None
        if previous_output_dict is not None and self.config.refine_net_use_rnn:
            bott = self.network
            while isinstance(bott, WrapEncoderDecoder):
                bott = bott.between_module
            states = [self._state_in(previous_output_dict['refinenet_rnn_states_%d' % i])
                      for i in range(len(bott.rnn_cells))]
        x, new_states = self._bottleneck(x, states, prefix, P)
        for i, st in enumerate(new_states):
            output_dict['refinenet_rnn_states_%d' % i] = self._state_out(st)
        y = self._decode(x, skips, P)
        output_dict['heatmap_final'] = y

    # ------------------------------------------------------------------ whole clips in one pass
    def forward_sequence(self, heatmap_initial, screen_frame=None):
        """heatmap_initial [B,T,1,h,w], screen_frame [B,T,3,H,W] -> (heatmap_final [B,T,1,H,W],
        list over cells of the stacked states [B,T,C,5,8] (tuple of two for CLSTM))."""
        P = self._get_packs()
        B, T = heatmap_initial.shape[:2]
        if screen_frame is not None and screen_frame.dtype == torch.uint8:     # decoded frames [B,T,H,W,3]: normalise here
            from . import data
            screen_frame = data.preprocess_screen_frames(screen_frame)
        fold = lambda t: None if t is None else t.reshape((B * T,) + tuple(t.shape[2:]))
        x = self._input_nhwc(fold(heatmap_initial), fold(screen_frame) if self.config.load_screen_content else None)
        x, skips, prefix = self._encode(x, P)
        C = x.shape[-1]
        xs = x.view(B, T, x.shape[1], x.shape[2], C)
        bott = self.network
        while isinstance(bott, WrapEncoderDecoder):
            bott = bott.between_module
        cells = list(bott.rnn_cells) if self.config.refine_net_use_rnn else []
        # one cell on the model's 5 x 8 x 64 bottleneck: the whole clip goes through it in ONE persistent launch (hidden state
        # resident in LDS); CGRU in the 16-bit formats on cgru_scan.hip, CGRU in float32 and CRNN / CLSTM in any format on the
        # float32 scans of cell_scan_f32.hip (round 5; the per-frame loop below remains for stacked cells / other geometries)
        scan = (len(cells) == 1 and tuple(xs.shape[2:]) == (5, 8, 64) and dispatch_flag(default_kernels(), 'cgru_scan', 1) != 0 and
                xs.dtype in HALF_DTYPES + (torch.float32,))
        if scan:
            cell, name = cells[0], '%s.rnn_cells.0' % prefix
            h5, w5 = x.shape[1], x.shape[2]
            to_ref = lambda t: ops.FromNHWCFn.apply(t.reshape(B * T, h5, w5, C), C).view(B, T, C, h5, w5)
            if isinstance(cell, CGRUCell):
                hs = ops.CGRUScanFn.apply(xs.contiguous(), cell.gates_1.weight, cell.gates_1.bias, cell.gate_2.weight,
                                          cell.gate_2.bias, None, P[name + '.gates_1'], P[name + '.gate_2'])
                states = [to_ref(hs)]
            elif isinstance(cell, CRNNCell):
                hs = ops.CRNNScanFn.apply(xs.float(), cell.cell.weight, cell.cell.bias, None, P[name + '.cell'])
                states = [to_ref(hs)]
                hs = hs.to(xs.dtype)
            else:                       # CLSTM: the state is computed and stored, the features pass through (refine_net.py:168-174)
                hcs = ops.clstm_scan(xs, cell.gates.weight, cell.gates.bias, P[name + '.gates'])
                states = [tuple(to_ref(t) for t in hcs)]
                hs = xs
            x = self._tap('rnn', hs.reshape(B * T, h5, w5, C))
            hf = self._decode(x, skips, P)
            return hf.view(B, T, 1, hf.shape[2], hf.shape[3]), states
        else:
            outs, states, hist = [], None, []
            for t in range(T):
                xt, states = self._bottleneck(xs[:, t].contiguous(), states, prefix, P)
                outs.append(xt)
                hist.append(states)
            x = torch.stack(outs, dim=1).view(B * T, x.shape[1], x.shape[2], C)
        hf = self._decode(x, skips, P)
        stacked = []
        for i in range(len(hist[0]) if hist else 0):
            per_t = [self._state_out(h[i]) for h in hist]
            if isinstance(per_t[0], tuple):
                stacked.append(tuple(torch.stack([p[j] for p in per_t], dim=1) for j in range(2)))
            else:
                stacked.append(torch.stack(per_t, dim=1))
        return hf.view(B, T, 1, hf.shape[2], hf.shape[3]), stacked
