"""EyeNet drop-in: per-eye, per-frame ResNet-18(InstanceNorm) encoder -> GRU -> gaze / pupil heads,
computed by the gfx950 HIP kernels of libeve_hip.so.

Mirrors /root/reference/src/models/eye_net.py:
  * constructor signature `EyeNet()` (:38) reading the config singleton, same sub-module / parameter
    names, so `state_dict()` keys equal the reference's (checkpoints split per first prefix by
    src/core/checkpoint_manager.py:56-67 and released weights load with strict=True);
  * `forward(input_dict, output_dict, side, previous_output_dict=None) -> None` (:98) writing
    `<side>_g_initial`, `<side>_pupil_size`, `<side>_eye_rnn_states_0` into `output_dict` and reading
    the previous state from `previous_output_dict` (:116-133); frozen => detached gaze (:149-150);
  * same error behaviour: ValueError for an unknown RNN type (:72), KeyError for missing dict entries.

In addition `forward_sequence` runs all T steps and both eyes of a clip batch in ONE pass: InstanceNorm
has no cross-sample coupling, so folding T and left/right into the image batch is numerically the same
as the reference's per-time-step loop (src/models/eve.py:91-111); only the GRU is sequential, and that
runs as one persistent scan kernel.

The nn.Conv2d / nn.Linear / nn.GRUCell objects below are PARAMETER HOLDERS (names, shapes, init); their
ATen forward is never called.
"""
import math
import os

import torch
from torch import nn

from . import ops
from .config import get_config
from .kernels import ACT_NONE, ACT_RELU, ACT_SELU, ACT_TANH, HALF_DTYPES, default_kernels, dispatch_flag, pad_channels
from .ops import PackedWeight

half_pi = 0.5 * math.pi


def default_compute_dtype():
    name = os.environ.get('EVE_AMD_DTYPE', 'fp32').lower()
    if name in ('bf16', 'bfloat16'):
        return torch.bfloat16
    if name in ('fp16', 'float16', 'half'):
        return torch.float16
    if name in ('fp32', 'float32', 'f32'):
        return torch.float32
    raise ValueError('EVE_AMD_DTYPE must be fp32, bf16 or fp16, got %r' % name)


class _Block(nn.Module):
    """Parameter holder for one torchvision BasicBlock (conv1, conv2, optional downsample.0)."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.stride = stride
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False),
                                            nn.InstanceNorm2d(cout))


class _ResNet18IN(nn.Module):
    """Parameter holder with torchvision ResNet-18 names (conv1, layer1..4, fc)."""

    def __init__(self, num_classes):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        cin = 64
        for li, (cout, stride) in enumerate([(64, 1), (128, 2), (256, 2), (512, 2)], start=1):
            setattr(self, 'layer%d' % li, nn.Sequential(_Block(cin, cout, stride), _Block(cout, cout, 1)))
            cin = cout
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def blocks(self):
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(self, 'layer%d' % li)):
                yield 'layer%d.%d' % (li, bi), blk


class EyeNet(nn.Module):
    def __init__(self):
        super(EyeNet, self).__init__()
        config = get_config()
        self.config = config
        self.compute_dtype = default_compute_dtype()
        nf = (config.eye_net_rnn_num_features if config.eye_net_use_rnn
              else config.eye_net_static_num_features)
        self.num_features = nf
        self.cnn_layers = _ResNet18IN(nf)
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
        self._packs = None
        self._packs_key = None

    # ------------------------------------------------------------------ packed weights
    def invalidate_packs(self):
        """Call after the parameters were updated behind torch's back (the fused Adam kernel)."""
        self._packs = None

    def _get_packs(self):
        dt = self.compute_dtype
        key = (dt,) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packs is not None and self._packs_key == key:
            return self._packs
        f32 = torch.float32
        P = {}
        cnn = self.cnn_layers
        P['conv1'] = PackedWeight(cnn.conv1.weight, dt, cin_pad=pad_channels(3, dt), want_ihwo=False, defer=True)
        for name, blk in cnn.blocks():
            P[name + '.conv1'] = PackedWeight(blk.conv1.weight, dt, defer=True)
            P[name + '.conv2'] = PackedWeight(blk.conv2.weight, dt, defer=True)
            if blk.downsample is not None:
                P[name + '.downsample.0'] = PackedWeight(blk.downsample[0].weight, dt, defer=True)
        PackedWeight.pack_many(list(P.values()), dt)          # all trunk weights in one launch
        # the tail (linears + GRU) always runs in float32: < 0.03 % of the FLOPs, and it carries the recurrence
        T = {}
        T['fc'] = PackedWeight(cnn.fc.weight, f32, defer=True)
        T['fc_common.0'] = PackedWeight(self.fc_common[0].weight, f32,
                                        cin_pad=pad_channels(self.fc_common[0].in_features, f32), defer=True)
        T['fc_common.2'] = PackedWeight(self.fc_common[2].weight, f32, defer=True)
        if self.config.eye_net_use_rnn:
            for i, cell in enumerate(self.rnn_cells):
                T['rnn.%d.ih' % i] = PackedWeight(cell.weight_ih, f32, defer=True)
                if i == 0 and self.config.eye_net_rnn_type == 'GRU':        # W_hh and its transpose for ops.EyeTailLossFn
                    T['rnn.0.hh'] = PackedWeight(cell.weight_hh, f32, defer=True)
        else:
            T['static_fc.0'] = PackedWeight(self.static_fc[0].weight, f32, defer=True)
        T['fc_to_gaze.0'] = PackedWeight(self.fc_to_gaze[0].weight, f32, defer=True)
        T['fc_to_gaze.2'] = PackedWeight(self.fc_to_gaze[2].weight, f32, cout_pad=4, defer=True)
        T['fc_to_pupil.0'] = PackedWeight(self.fc_to_pupil[0].weight, f32, defer=True)
        T['fc_to_pupil.2'] = PackedWeight(self.fc_to_pupil[2].weight, f32, cout_pad=4, defer=True)
        PackedWeight.pack_many(list(T.values()), f32)
        P.update(T)
        self._packs, self._packs_key = P, key
        return P

    # ------------------------------------------------------------------ trunk
    def _trunk(self, x, P, x_padded=None):
        """x: [N, H, W, Cpad] NHWC compute dtype -> [N, 512] float32 (torchvision ResNet._forward_impl).
        x_padded: optional [N, H+6, W+8, 4] bf16 repack for the dedicated stem kernel."""
        y = self._trunk_layers(x, P, x_padded)
        if hasattr(ops.default_kernels(), 'avgpool_fwd_f32'):
            return ops.AvgPoolF32Fn.apply(y)                  # pool + cast in one launch each way (same bits)
        return ops.cast(ops.AvgPoolFn.apply(y), torch.float32)

    def _trunk_layers(self, x, P, x_padded=None):
        """conv1 .. layer4 of the trunk: -> [N, H/32, W/32, 512] NHWC compute dtype."""
        cnn = self.cnn_layers
        blocks, weights = [], []
        for name, blk in cnn.blocks():
            ds = blk.downsample
            blocks.append(((P[name + '.conv1'], P[name + '.conv2'], P[name + '.downsample.0'] if ds is not None else None),
                           blk.stride))
            weights += [blk.conv1.weight, blk.conv2.weight] + ([ds[0].weight] if ds is not None else [])
        if x_padded is not None and x_padded.shape[2] == 136 and x_padded.shape[1] % 4 == 2:
            # 128-wide patches: conv1 -> bn1 -> relu -> maxpool in one launch, inside the trunk node
            y = ops.ResNetTrunkFn.apply(None, x, x_padded, (P['conv1'], tuple(blocks)), 1e-5, cnn.conv1.weight, *weights)
        else:
            if x_padded is not None:
                y = ops.StemConvFn.apply(x, x_padded, cnn.conv1.weight, P['conv1'])
            else:
                y = ops.conv2d(x, cnn.conv1.weight, None, P['conv1'], stride=2, pad=3)
            y = ops.InReluMaxPoolFn.apply(y, 1e-5)      # bn1 -> relu -> maxpool, fused
            y = ops.ResNetTrunkFn.apply(y, None, None, (None, tuple(blocks)), 1e-5, *weights)
        return y

    # ------------------------------------------------------------------ tail: fc -> fc_common -> GRU -> heads
    def _tail(self, feats, head_pose, S, T, h0, P):
        """feats [S*T, 512] float32 ordered (sequence, time); head_pose [S*T, 2] or None; h0: None or one initial
        state per cell ([S, H] tensor, (h, c) pair for LSTM, or None).
        Returns gaze [S*T, 2] (rad), pupil [S*T], states: None or per cell [S, T, H] / ((h) [S, T, H], (c) [S, T, H])."""
        cfg = self.config
        cnn = self.cnn_layers
        k = default_kernels()
        f = ops.linear(feats, cnn.fc.weight, cnn.fc.bias, P['fc'])
        if cfg.eye_net_use_head_pose_input:
            f = torch.cat([f, head_pose.to(f.dtype)], dim=1)
        cin_p = P['fc_common.0'].ohwi.shape[3]
        if f.shape[1] != cin_p:
            f = torch.nn.functional.pad(f, (0, cin_p - f.shape[1]))
        f = ops.linear(f.contiguous(), self.fc_common[0].weight, self.fc_common[0].bias, P['fc_common.0'],
                       act=ACT_SELU)
        f = ops.linear(f, self.fc_common[2].weight, self.fc_common[2].bias, P['fc_common.2'])
        states = None
        if cfg.eye_net_use_rnn:
            # the cells form a stack over depth; cell i at step t reads cell i-1 at step t and its own state at t-1, so
            # each cell is one scan over the whole sequence of its predecessor's outputs (eye_net.py:112-135)
            states = []
            for i, cell in enumerate(self.rnn_cells):
                init = h0[i] if h0 is not None else None
                gi = ops.linear(f, cell.weight_ih, cell.bias_ih, P['rnn.%d.ih' % i])
                H = cell.hidden_size
                kind = cfg.eye_net_rnn_type
                if kind == 'GRU':
                    st = ops.GRUScanFn.apply(gi.view(S, T, 3 * H), cell.weight_hh, cell.bias_hh, init)
                    hs = st
                elif kind == 'RNN':
                    st = ops.RNNScanFn.apply(gi.view(S, T, H), cell.weight_hh, cell.bias_hh, init)
                    hs = st
                else:                                   # LSTM: the state is the (h, c) pair, the output is h
                    h_init, c_init = init if init is not None else (None, None)
                    st = ops.LSTMScanFn.apply(gi.view(S, T, 4 * H), cell.weight_hh, cell.bias_hh, h_init, c_init)
                    hs = st[0]
                states.append(st)
                f = hs.reshape(S * T, H)
        else:
            f = ops.linear(f, self.static_fc[0].weight, self.static_fc[0].bias, P['static_fc.0'], act=ACT_SELU)
        g = ops.linear(f, self.fc_to_gaze[0].weight, self.fc_to_gaze[0].bias, P['fc_to_gaze.0'], act=ACT_SELU)
        g = ops.linear(g, self.fc_to_gaze[2].weight, None, P['fc_to_gaze.2'], act=ACT_TANH)
        gaze = half_pi * g[:, :2]
        p = ops.linear(f, self.fc_to_pupil[0].weight, self.fc_to_pupil[0].bias, P['fc_to_pupil.0'], act=ACT_SELU)
        p = ops.linear(p, self.fc_to_pupil[2].weight, self.fc_to_pupil[2].bias, P['fc_to_pupil.2'], act=ACT_RELU)
        return gaze, p[:, 0], states

    # ------------------------------------------------------------------ train step: tail + losses as one node
    tail_loss_node = None        # None: eve_dispatch_config.tail_loss_node (default on); a test sets True / False on the instance

    def _tail_loss_node_ok(self, batch, T, feats=None):
        cfg = self.config
        if feats is not None and not feats.requires_grad:
            # (frozen trunk + trainable tail: the node takes the tail parameters through `self`, not as autograd inputs, so
            # its outputs would carry no graph -- the per-layer path handles that configuration)
            return False
        k = default_kernels()
        on = self.tail_loss_node if self.tail_loss_node is not None else bool(dispatch_flag(k, 'tail_loss_node', 1))
        if not (on and hasattr(k, 'tail_outputs_fwd') and torch.is_grad_enabled() and cfg.eye_net_use_rnn and
                cfg.eye_net_rnn_type == 'GRU' and len(self.rnn_cells) == 1 and cfg.eye_net_use_head_pose_input and
                not cfg.eye_net_frozen and self.rnn_cells[0].hidden_size == 128 and self.fc_common[0].in_features == 130 and
                self.cnn_layers.fc.in_features == 512 and T <= 256 and batch['left_eye_patch'].is_cuda and
                batch['left_g_tobii'].dtype == torch.float32 and batch['left_h'].dtype == torch.float32):
            return False
        return all(ops._direct_grad_ok(p_) for p_ in ops.EyeTailLossFn.tail_parameters(self))

    def loss_terms_sequence(self, batch, config=None):
        """The EyeNet train step's forward: whole clips of both eyes -> the loss terms of eve.py:286-325 that carry weight in
        eye_net.json and their weighted sum (:234-265) -- what losses.eyenet_loss_terms(self.forward_sequence(batch), ...) returns,
        with the tail and the losses as ONE autograd node (ops.EyeTailLossFn) when the configuration is the product one
        (train.eyenet_trainer: one GRU cell, head-pose input, parameters in the trainer's flat buffers); any other
        configuration takes the per-layer path.  Also returns the predictions (detached) under the forward_sequence keys."""
        from . import losses
        config = config if config is not None else self.config
        P = self._get_packs()
        feats, B, T = self._sequence_features(batch, P)
        self.last_tail_path = 'node' if self._tail_loss_node_ok(batch, T, feats) else 'layers'
        if self.last_tail_path == 'layers':
            out = self._sequence_tail(feats, batch, B, T, None, P)
            terms = losses.eyenet_loss_terms(out, batch, config)
            terms.update({k_: v.detach() for k_, v in out.items() if torch.is_tensor(v)})
            return terms
        tgt = tuple(batch[k_] for k_ in ('left_g_tobii', 'right_g_tobii', 'left_g_tobii_validity', 'right_g_tobii_validity',
                                         'left_p', 'right_p', 'left_p_validity', 'right_p_validity'))
        packs = tuple(P[n] for n in ('fc', 'fc_common.0', 'fc_common.2', 'rnn.0.ih', 'rnn.0.hh', 'fc_to_gaze.0', 'fc_to_gaze.2',
                                     'fc_to_pupil.0', 'fc_to_pupil.2'))
        t = ops.EyeTailLossFn.apply(feats, batch['left_h'], batch['right_h'], tgt, float(config.loss_coeff_g_ang_initial),
                                    float(config.loss_coeff_pupil_size), self, packs, B, T)
        gaze, pupil, hs = t[5], t[6], t[7]
        BT = B * T
        return {'loss_ang_left_g_initial': t[0], 'loss_l1_left_pupil_size': t[1], 'loss_ang_right_g_initial': t[2],
                'loss_l1_right_pupil_size': t[3], 'full_loss': t[4],
                'left_g_initial': gaze[:BT].view(B, T, 2), 'right_g_initial': gaze[BT:].view(B, T, 2),
                'left_pupil_size': pupil[:BT].view(B, T), 'right_pupil_size': pupil[BT:].view(B, T),
                'left_eye_rnn_states_0': hs[:B], 'right_eye_rnn_states_0': hs[B:]}

    # ------------------------------------------------------------------ reference per-step contract
    def forward(self, input_dict, output_dict, side, previous_output_dict=None):
        key = side + '_eye_patch'
        image = output_dict[key] if key in output_dict else input_dict[key]
        P = self._get_packs()
        dt = self.compute_dtype
        x = ops.ToNHWCFn.apply(image, dt, pad_channels(image.shape[1], dt))
        feats = self._trunk(x, P)
        head_pose = input_dict[side + '_h'] if self.config.eye_net_use_head_pose_input else None
        h0 = None
        ncell = len(self.rnn_cells) if self.config.eye_net_use_rnn else 0
        if ncell and previous_output_dict is not None:
            h0 = [previous_output_dict[side + '_eye_rnn_states_%d' % i] for i in range(ncell)]
        B = image.shape[0]
        gaze, pupil, states = self._tail(feats, head_pose, B, 1, h0, P)
        for i in range(ncell):
            st = states[i]
            output_dict[side + '_eye_rnn_states_%d' % i] = tuple(s_[:, 0] for s_ in st) if isinstance(st, tuple) else st[:, 0]
        output_dict[side + '_g_initial'] = gaze
        output_dict[side + '_pupil_size'] = pupil.reshape(-1)
        if self.config.eye_net_frozen:
            output_dict[side + '_g_initial'] = output_dict[side + '_g_initial'].detach()

    # ------------------------------------------------------------------ whole clips, both eyes, one pass
    def forward_sequence(self, batch, initial_states=None):
        """batch: {left,right}_eye_patch [B, T, 3, H, W] float, {left,right}_h [B, T, 2].
        Returns the B x T x ... tensors eve.py:174-182 would stack: <side>_g_initial [B,T,2],
        <side>_pupil_size [B,T], <side>_eye_rnn_states_<i> [B,T,H] per cell ((h, c) pair of them for LSTM).
        initial_states: {side: h [B,H]} or {side: [per-cell h | (h, c) | None]}."""
        P = self._get_packs()
        feats, B, T = self._sequence_features(batch, P)
        return self._sequence_tail(feats, batch, B, T, initial_states, P)

    def _sequence_features(self, batch, P):
        """Both eyes' clips through the trunk: -> feats [2*B*T, 512] float32 (left clips' frames, then the right ones'), B, T."""
        k = default_kernels()
        dt = self.compute_dtype
        left, right = batch['left_eye_patch'], batch['right_eye_patch']
        x = x_padded = None
        if left.dtype == torch.uint8:
            # decoded frames [B, T, H, W, C] (eve_sequences.py:196-203 not applied yet): normalise on the device --
            # straight into the stem's packed layout when the fused stem takes it, else to the reference's float NCHW
            from . import data
            B, T, Hh, Ww, C = left.shape
            if dt in HALF_DTYPES and C <= 4 and Hh % 4 == 0 and Ww == 128:
                x_padded = torch.empty((2 * B * T, Hh + 6, Ww + 8, 4), dtype=dt, device=left.device)
                k.frames_u8_to_stem(left.reshape(B * T, Hh, Ww, C), data.EYE_SCALE, data.EYE_SHIFT, out=x_padded[:B * T])
                k.frames_u8_to_stem(right.reshape(B * T, Hh, Ww, C), data.EYE_SCALE, data.EYE_SHIFT, out=x_padded[B * T:])
            else:
                left, right = data.preprocess_frames(left), data.preprocess_frames(right)
        if x_padded is None:
            B, T, C, Hh, Ww = left.shape
        cpad = pad_channels(C, dt)
        if x_padded is not None:                                                   # packed from the uint8 frames above
            pass
        elif dt in HALF_DTYPES and C <= 4 and Hh % 4 == 0 and Ww == 128:        # fused stem: packed patches only
            x_padded = torch.empty((2 * B * T, Hh + 6, Ww + 8, 4), dtype=dt, device=left.device)
            k.stem_pack_input(left.reshape(B * T, C, Hh, Ww), out=x_padded[:B * T])
            k.stem_pack_input(right.reshape(B * T, C, Hh, Ww), out=x_padded[B * T:])
        else:
            stem_kernel = dt in HALF_DTYPES and C <= 4 and Hh % 4 == 0 and Ww % 128 == 0          # dedicated stem conv kernel
            # its weight gradient reads the packed patches too (ops.StemConvFn) while they stay below 2 GiB: the 8-channel NHWC
            # copy (2 GB at configs[4]'s 1 920 frames of 256 x 256) is then not made at all
            packed_wgrad = stem_kernel and 2 * B * T * (Hh + 6) * (Ww + 8) * 8 < (1 << 31)
            if not packed_wgrad:
                x = torch.empty((2 * B * T, Hh, Ww, cpad), dtype=dt, device=left.device)
                k.nchw_to_nhwc(left.reshape(B * T, C, Hh, Ww), dt, cpad, out=x[:B * T])
                k.nchw_to_nhwc(right.reshape(B * T, C, Hh, Ww), dt, cpad, out=x[B * T:])
            if stem_kernel:
                x_padded = torch.empty((2 * B * T, Hh + 6, Ww + 8, 4), dtype=dt, device=left.device)
                k.stem_pack_input(left.reshape(B * T, C, Hh, Ww), out=x_padded[:B * T])
                k.stem_pack_input(right.reshape(B * T, C, Hh, Ww), out=x_padded[B * T:])
        return self._trunk(x, P, x_padded), B, T

    def _sequence_tail(self, feats, batch, B, T, initial_states, P):
        head_pose = None
        if self.config.eye_net_use_head_pose_input:
            head_pose = torch.cat([batch['left_h'].reshape(B * T, 2), batch['right_h'].reshape(B * T, 2)], dim=0)
        ncell = len(self.rnn_cells) if self.config.eye_net_use_rnn else 0
        h0 = None
        if initial_states is not None and ncell:
            def per_cell(v):                      # a bare tensor is the first cell's hidden state
                return list(v) if isinstance(v, (list,)) else [v] + [None] * (ncell - 1)
            lft, rgt = per_cell(initial_states['left']), per_cell(initial_states['right'])
            h0 = []
            for a, b in zip(lft, rgt):
                if a is None:
                    h0.append(None)
                elif isinstance(a, tuple):
                    h0.append(tuple(torch.cat([x_, y_], dim=0) for x_, y_ in zip(a, b)))
                else:
                    h0.append(torch.cat([a, b], dim=0))
        gaze, pupil, states = self._tail(feats, head_pose, 2 * B, T, h0, P)
        out = {}
        for si, side in enumerate(('left', 'right')):
            sl = slice(si * B * T, (si + 1) * B * T)
            g = gaze[sl].reshape(B, T, 2)
            out[side + '_g_initial'] = g.detach() if self.config.eye_net_frozen else g
            out[side + '_pupil_size'] = pupil[sl].reshape(B, T)
            for i in range(ncell):
                st = states[i]
                cut = slice(si * B, (si + 1) * B)
                out[side + '_eye_rnn_states_%d' % i] = tuple(s_[cut] for s_ in st) if isinstance(st, tuple) else st[cut]
        return out
