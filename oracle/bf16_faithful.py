"""Oracle, rounding-faithful mode: the reference's EyeNet / RefineNet arithmetic with bfloat16 rounding applied at
exactly the tensors the HIP bf16 instantiation stores in bfloat16 (and float32 everywhere the kernels keep float32:
MFMA accumulators, InstanceNorm statistics, biases, the EyeNet tail, parameter gradients).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The float32 oracle (oracle/eye_net.py, oracle/refine_net.py, pinned
against the reference's own classes) says what the path computes; this module says what the SAME path computes when
its activations and activation gradients live in bf16, so that the benchmarked bf16 kernels can be compared with a
tight tolerance instead of "somewhere within bf16 noise of fp32".  It is tied to the pinned oracle by construction:

  * it runs on the oracle's own modules (same parameters, same state_dict) and only re-states the order of operations
    of /root/reference/src/models/eye_net.py:98-150, the torchvision 0.6.1 ResNet it constructs (:48-50),
    refine_net.py:35-255 and common.py:388-415;
  * with ``rounding(False)`` every R()/Rb()/Rf() below is the identity and the functions reproduce the float32
    oracle's outputs and (autograd) gradients to float32 accuracy -- tests/test_oracle_golden.py checks that.

Rounding model (what eve_amd/ops.py + the kernels do; file:line of the kernel behaviour in the comments):
  R(x)   value rounded to bf16 on the way forward AND its gradient rounded on the way back: a tensor an op writes to
         HBM in bf16 whose gradient tensor is bf16 as well (every conv / InstanceNorm / add / pool output);
  Rb(x)  gradient rounded only: one consumer's contribution to a fan-in that the kernels keep as separate bf16
         summands (residual forks) or the bf16 tensor of d(pre-activation) an epilogue / act_bwd kernel writes;
  Rf(x)  value rounded only: inputs and the bf16 copies of the float32 master weights (their gradient stays float32).
Gradients are autograd's, so they are derivatives of exactly the rounded forward by construction.
"""
import contextlib
import math

import torch
from torch.nn import functional as F

_ENABLED = True
_FORMAT = torch.bfloat16          # the 16-bit storage format the rounding points model: bfloat16 or float16


@contextlib.contextmanager
def rounding(enabled, fmt=torch.bfloat16):
    """rounding(False): every rounding point becomes the identity (float32 oracle arithmetic).
    rounding(True, torch.float16): the float16 instantiation of the kernels (BASELINE configs[4]) -- same rounding points,
    IEEE half instead of bfloat16 (the kernels are templates over the format; nothing else differs)."""
    global _ENABLED, _FORMAT
    assert fmt in (torch.bfloat16, torch.float16)
    old, _ENABLED = (_ENABLED, _FORMAT), bool(enabled)
    _FORMAT = fmt
    try:
        yield
    finally:
        _ENABLED, _FORMAT = old


def _bf16(x):
    """x rounded to the active 16-bit format (named after the default one)."""
    return x.to(_FORMAT).to(torch.float32) if _ENABLED else x


class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return _bf16(x) if fwd else x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return (_bf16(g) if ctx.bwd else g), None, None


def R(x):
    return _Round.apply(x, True, True)


def Rb(x):
    return _Round.apply(x, False, True)


def Rf(x):
    return _Round.apply(x, True, False)


def _tap(taps, name, x):
    """Record a stage boundary (for teacher-forced, stage-local comparisons): the tensor with its gradient retained."""
    if taps is not None:
        if x.requires_grad:
            x.retain_grad()
        taps[name] = x
    return x


class _ActFromOutput(torch.autograd.Function):
    """y = bf16(act(z)); dz = bf16(dy * act'(.)) with act' evaluated from the STORED y, as the conv epilogues /
    eve_act_bwd do for sigmoid and tanh (eve_amd/csrc/common.h act_grad_from_out)."""

    @staticmethod
    def forward(ctx, z, kind):
        y = _bf16(torch.sigmoid(z) if kind == 'sigmoid' else torch.tanh(z))
        ctx.kind = kind
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        d = y * (1. - y) if ctx.kind == 'sigmoid' else 1. - y * y
        return _bf16(_bf16(dy) * d), None


def conv(x, m, stride=None, pre_act_out=False):
    """nn.Conv2d `m` on a bf16-valued x: bf16 operands, float32 accumulation and bias, bf16 result
    (conv_fast.h epilogues pack with v_cvt_pk_bf16_f32 after the float bias add)."""
    y = F.conv2d(Rb(x), Rf(m.weight), m.bias, stride if stride is not None else m.stride, m.padding)
    return Rb(y) if pre_act_out else R(y)


def in_act(x, norm=None, res=None, act=None, eps=1e-5):
    """act(IN(x) * gamma + beta + res) as ONE kernel (norm_fused.hip in_fwd_fused_kernel / norm_act.hip): float32
    statistics of the bf16 x, bf16 result; backward: g = bf16(dy * act'), then the float32 InstanceNorm backward on
    that rounded g (norm_fused.hip in_bwd_fused_kernel "the SAME rounded g")."""
    gamma = norm.weight if norm is not None and getattr(norm, 'weight', None) is not None else None
    beta = norm.bias if gamma is not None else None
    z = F.instance_norm(Rb(x), weight=gamma, bias=beta, eps=eps)
    if res is not None:
        z = z + Rb(res)
    z = Rb(z)
    if act == 'relu':
        z = F.relu(z)
    elif act == 'leaky':
        z = F.leaky_relu(z, 0.01)
    return R(z)


# ----------------------------------------------------------------------------------------------------- EyeNet
def resnet_stem(cnn, x):
    """conv1 -> bn1 -> relu -> maxpool as stem_fused.hip computes it: the convolution output is never stored, statistics
    come from the float32 accumulators, only the pooled tensor is bf16; its backward stores d(conv) in bf16."""
    c = F.conv2d(Rf(x), Rf(cnn.conv1.weight), None, 2, 3)
    c = Rb(c)
    mean = c.mean(dim=(2, 3), keepdim=True)
    var = c.var(dim=(2, 3), unbiased=False, keepdim=True)
    # relu(IN(.)) is monotone per channel, so the kernel pools the RAW convolution values, parks the window maxima in
    # bf16 (stem_fused.hip: "writes only the pooled 32x32x64 tensor (raw, bf16)") and normalises them once the plane
    # statistics are known: the pooled value is rounded before AND after the normalisation
    p = Rf(F.max_pool2d(c, 3, 2, 1))
    return R(F.relu((p - mean) * torch.rsqrt(var + 1e-5)))


def resnet_block(blk, y):
    """torchvision BasicBlock with InstanceNorm2d: out = relu(IN(conv2(relu(IN(conv1(x))))) + identity)."""
    a = conv(y, blk.conv1)
    an = in_act(a, act='relu')
    b = conv(an, blk.conv2)
    idn = y
    if blk.downsample is not None:
        idn = in_act(conv(y, blk.downsample[0]))
    return in_act(b, res=idn, act='relu')


def resnet_trunk(cnn, x, taps=None):
    """oracle.resnet_in.ResNet `cnn` up to the pooled 512 features, N x 3 x H x W float -> N x 512 (bf16-valued).
    taps (dict): receives 'stem', 'layer<L>.<B>' (block outputs) and 'pooled'."""
    y = _tap(taps, 'stem', resnet_stem(cnn, x))
    for li, layer in enumerate((cnn.layer1, cnn.layer2, cnn.layer3, cnn.layer4), start=1):
        for bi, blk in enumerate(layer):
            y = _tap(taps, 'layer%d.%d' % (li, bi), resnet_block(blk, y))
    return _tap(taps, 'pooled', R(y.mean(dim=(2, 3))))     # avg-pool kernel: float sum of the bf16 plane, bf16 result


def eyenet_tail_step(net, feats, head_pose, prev_states):
    """eye_net.py:106-150 after the CNN trunk, float32 (the HIP tail is float32 as well): fc -> cat head pose ->
    fc_common -> recurrent cells / static_fc -> heads.  Returns (gaze, pupil, [states])."""
    cfg = net.config
    f = net.cnn_layers.fc(feats)
    if cfg.eye_net_use_head_pose_input:
        f = torch.cat([f, head_pose], dim=1)
    f = net.fc_common(f)
    states = []
    if cfg.eye_net_use_rnn:
        for i, cell in enumerate(net.rnn_cells):
            st = cell(f, None if prev_states is None else prev_states[i])
            states.append(st)
            f = st[0] if isinstance(st, tuple) else st
    else:
        f = net.static_fc(f)
    gaze = 0.5 * math.pi * net.fc_to_gaze(f)
    pupil = net.fc_to_pupil(f).reshape(-1)
    return gaze, pupil, states


def eyenet_sequence(net, batch):
    """Same contract as oracle.sequence.eyenet_sequence (B x T x ... dict), bf16-faithful trunk.  T and left/right
    are folded into the image batch exactly like EyeNet.forward_sequence does (InstanceNorm is per image)."""
    B, T = batch['left_eye_patch'].shape[:2]
    out = {}
    for side in ('left', 'right'):
        x = batch[side + '_eye_patch'].reshape((B * T,) + tuple(batch[side + '_eye_patch'].shape[2:]))
        feats = resnet_trunk(net.cnn_layers, x).view(B, T, -1)
        prev, gs, ps, sts = None, [], [], []
        for t in range(T):
            g, p, prev = eyenet_tail_step(net, feats[:, t], batch[side + '_h'][:, t], prev)
            gs.append(g)
            ps.append(p)
            sts.append(prev)
        g = torch.stack(gs, dim=1)
        out[side + '_g_initial'] = g.detach() if net.config.eye_net_frozen else g
        out[side + '_pupil_size'] = torch.stack(ps, dim=1)
        for i in range(len(sts[0])):
            if isinstance(sts[0][i], tuple):
                out[side + '_eye_rnn_states_%d' % i] = tuple(torch.stack([s[i][j] for s in sts], dim=1) for j in range(2))
            else:
                out[side + '_eye_rnn_states_%d' % i] = torch.stack([s[i] for s in sts], dim=1)
    return out


# ----------------------------------------------------------------------------------------------------- RefineNet
def _preact_block(x, blk, act):
    """refine_net.py:35-67: IN -> act -> conv3x3 -> IN -> act -> conv3x3, plus the 1x1 pre-activation skip; the sum is
    one bf16 add kernel."""
    L = blk.layers
    a = conv(in_act(x, L[0], act=act), L[2])
    a = conv(in_act(a, L[3], act=act), L[5])
    skip = x
    if blk.skip_layer is not None:
        S = blk.skip_layer
        skip = conv(in_act(x, S[0], act=act), S[2])
    return R(Rb(a) + Rb(skip))


def cgru_step(cell, x, h):
    """common.py:400-415 as cgru_scan.hip evaluates it: both gate convolutions accumulate in float32 and go through
    sigmoid / tanh before anything is stored; the stored bf16 gates are what every later stage sees."""
    g1 = F.conv2d(torch.cat([Rb(x), Rb(h)], dim=1), Rf(cell.gates_1.weight), cell.gates_1.bias, 1, 1)
    ru = _ActFromOutput.apply(g1, 'sigmoid')
    r, u = ru.chunk(2, 1)
    rh = R(r * Rb(h))
    g2 = F.conv2d(torch.cat([rh, Rb(x)], dim=1), Rf(cell.gate_2.weight), cell.gate_2.bias, 1, 1)
    o = _ActFromOutput.apply(g2, 'tanh')
    return R((1. - u) * o + u * Rb(h))


def refinenet_sequence(net, heatmap_initial, screen_frame=None, float_sigmoid=True, taps=None):
    """oracle.sequence.refinenet_sequence's contract (heatmap_final B x T x 1 x 72 x 128, per-step CGRU states) with the
    encoder / decoder on the folded B*T frame batch, as RefineNet.forward_sequence runs it.  CGRU only (the benchmarked
    cell); the final sigmoid is evaluated in float32 from the bf16 logits when `float_sigmoid` (eve_amd does so since
    round 2), else as the bf16 epilogue of the last convolution."""
    from .refine_net import CGRUCell, WrapEncoderDecoder
    cfg = net.config
    B, T = heatmap_initial.shape[:2]
    fold = lambda t: t.reshape((B * T,) + tuple(t.shape[2:]))
    heat = fold(heatmap_initial)
    size = (cfg.screen_size[1], cfg.screen_size[0])
    if tuple(heat.shape[-2:]) != size:
        heat = R(F.interpolate(Rf(heat), size, mode='bilinear', align_corners=False))
    x = torch.cat([fold(screen_frame), heat], dim=1) if cfg.load_screen_content else heat
    x = _tap(taps, 'input', Rf(x))
    x = conv(x, net.initial[0])
    x = _tap(taps, 'initial', conv(in_act(x, net.initial[1], act='relu'), net.initial[3]))
    levels, level = [], net.network
    while isinstance(level, WrapEncoderDecoder):
        levels.append(level)
        level = level.between_module
    skips = []
    for d, lv in enumerate(levels):
        for i, blk in enumerate(lv.encoder_blocks):
            x = _tap(taps, 'enc%d.%d' % (d, i), _preact_block(x, blk, 'relu'))
        skips.append(x)
        if lv.downsample is not None:
            x = _tap(taps, 'pool%d' % d, R(lv.downsample(Rb(x))))
    states = []
    cells = list(level.rnn_cells) if cfg.refine_net_use_rnn else []
    if cells:
        assert len(cells) == 1 and isinstance(cells[0], CGRUCell), 'bf16-faithful mode restates the CGRU cell only'
        xs = x.view((B, T) + tuple(x.shape[1:]))
        h, hs = torch.zeros_like(xs[:, 0]), []
        for t in range(T):
            h = cgru_step(cells[0], xs[:, t], h)
            hs.append(h)
        states = hs
        x = _tap(taps, 'rnn', torch.stack(hs, dim=1).view_as(x))
    for d, lv, enc in zip(reversed(range(len(levels))), reversed(levels), reversed(skips)):
        if lv.upsample is not None:
            x = _tap(taps, 'up%d' % d, R(lv.upsample(Rb(x))))
        if lv.add_skip_connection:
            x = torch.cat([Rb(x), Rb(enc)], dim=1)
        for i, blk in enumerate(lv.decoder_blocks):
            x = _tap(taps, 'dec%d.%d' % (d, i), _preact_block(x, blk, 'leaky'))
    x = _tap(taps, 'final0', R(F.leaky_relu(conv(x, net.final[0], pre_act_out=True), 0.01)))
    if float_sigmoid:
        hf = torch.sigmoid(_tap(taps, 'logits', conv(x, net.final[2])))
    else:
        hf = _ActFromOutput.apply(F.conv2d(Rb(x), Rf(net.final[2].weight), net.final[2].bias), 'sigmoid')
    return hf.view(B, T, 1, hf.shape[2], hf.shape[3]), states
