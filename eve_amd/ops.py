"""Differentiable operators of the hot path: torch.autograd.Function shells whose forward AND
backward are HIP kernels launched through the C ABI (eve_amd/kernels.py -> include/eve_hip.h).

All activations here are NHWC in the compute dtype.  Autograd is used only to order the backward
launches and to sum gradient fan-in; no arithmetic on the path is done by ATen.
"""

import torch

from . import kernels as K
from .kernels import (ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SELU, ACT_SIGMOID, ACT_TANH,  # noqa: F401
                      HALF_DTYPES, default_kernels, dispatch_flag)


class _LazyGroup(object):
    """(F, grouped filter) of a PackedWeight, the filter built on first use: pair[0] (what _group_ok asks) costs nothing, and the
    layers the streaming kernels took over never build theirs (each is a cat + fill + gather: 75 launches per configs[2] step
    when every candidate layer built both of its grouped filters eagerly, round 5)."""
    __slots__ = ('F', 'ohwi', 'transpose', 'w')

    def __init__(self, F, ohwi, transpose):
        self.F, self.ohwi, self.transpose, self.w = F, ohwi, transpose, None

    def weights(self):
        if self.w is None:
            self.w = pixel_group_weights(self.ohwi, self.F, self.transpose)
        return self.w

    def __getitem__(self, i):
        return self.F if i == 0 else self.weights()

    def __iter__(self):
        yield self.F
        yield self.weights()


class PackedWeight(object):
    """Compute-dtype copies of one conv/linear weight: OHWI for forward/wgrad, IHWO for dgrad.

    `param` has the reference's OIHW (or [out, in]) SHAPE; its memory may already be OHWI (the
    flat-parameter harness stores it that way), in which case no permute copy is made."""

    __slots__ = ('ohwi', 'ihwo', 'shape_oihw', 'algo', 'pair_fwd', 'pair_dgrad', '_pairs_for')

    def __init__(self, param, dtype, cin_pad=None, cout_pad=None, want_ihwo=True, defer=False):
        k = default_kernels()
        w = param.detach()
        if w.dim() == 2:
            w = w.view(w.shape[0], w.shape[1], 1, 1)
        self.shape_oihw = tuple(w.shape)
        self.algo = (w.shape[0], w.shape[1] * w.shape[2] * w.shape[3])     # true Cout, true K
        w = w.permute(0, 2, 3, 1)                      # OHWI view
        self.pair_fwd = self.pair_dgrad = None
        self._pairs_for = (dtype, param.dim(), want_ihwo)
        if defer and hasattr(k, 'dispatch_config'):   # packed later, together with others (pack_many); the batched launch writes
            padded = (cout_pad or w.shape[0], cin_pad or w.shape[3])         # the padding channels itself (no fill + copy per weight)
            self.ohwi, self.ihwo = (w.contiguous().float(), want_ihwo, padded), None
            return
        if cin_pad is not None and cin_pad != w.shape[3]:
            w = torch.nn.functional.pad(w, (0, cin_pad - w.shape[3]))
        if cout_pad is not None and cout_pad != w.shape[0]:
            w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, cout_pad - w.shape[0]))
        w = w.contiguous().float()
        if defer:
            self.ohwi, self.ihwo = (w, want_ihwo, None), None
        else:
            self.ohwi, self.ihwo = k.pack_weights(w, dtype, want_ihwo=want_ihwo)
            self._make_pairs()

    def _make_pairs(self):
        """Grouped filters (built lazily, see _LazyGroup) of the layers narrow enough to run over pixel groups."""
        dtype, pdim, want_ihwo = self._pairs_for
        w = self.ohwi
        if dtype in HALF_DTYPES and w.shape[1] == w.shape[2] and pdim == 4:
            cout_p, cin_p = w.shape[0], w.shape[3]
            fac = _group_factors(w.shape[1])
            if cin_p in fac:
                self.pair_fwd = _LazyGroup(fac[cin_p], self.ohwi, False)
            elif (w.shape[1], cin_p, cout_p) in PAIR_NARROW_OUT:
                F = PAIR_NARROW_OUT[(w.shape[1], cin_p, cout_p)]
                self.pair_fwd = _LazyGroup(F, self.ohwi, False)
            if want_ihwo and cout_p in fac:
                self.pair_dgrad = _LazyGroup(fac[cout_p], self.ohwi, True)

    @staticmethod
    def pack_many(packs, dtype):
        """Finish PackedWeight(..., defer=True) objects with one batched launch."""
        todo = [p for p in packs if isinstance(p.ohwi, tuple)]
        if not todo:
            return
        k = default_kernels()
        if any(p.ohwi[2] is not None for p in todo):
            res = k.pack_weights_batch([p.ohwi[0] for p in todo], dtype, [p.ohwi[1] for p in todo], padded=[p.ohwi[2] for p in todo])
        else:
            res = k.pack_weights_batch([p.ohwi[0] for p in todo], dtype, [p.ohwi[1] for p in todo])
        for p, (ohwi, ihwo) in zip(todo, res):
            p.ohwi, p.ihwo = ohwi, ihwo
            p._make_pairs()


# ---- 3x3 / stride 1 / pad 1 convolutions over 8- or 16-channel tensors (RefineNet's 72x128 level, refine_net.py:
# 96-131 of the reference) -------------------------------------------------------------------------------------------
# An NHWC row [W][C] with C = 32/F channels is, byte for byte, a row [W/F][32]: F neighbouring pixels form one
# 32-channel "group pixel".  The same convolution over group pixels is again 3x3 / pad 1, from 32 input channels to
# F*Cout output channels (the F output pixels of a group), with a weight tensor that holds each original tap once per
# (output pixel, input pixel) pair of the groups it connects and zeros elsewhere.  That is exactly the shape the
# halo-resident MFMA kernel is built for (32-channel K steps), so these layers run there instead of on the generic
# gather kernel (551 -> 200-260 us per launch at 960 x 72 x 128); the matrix units do F x the arithmetic on a
# bandwidth-bound layer.
# 1x1 convolutions group pixels up to 64 channels (the LDS-DMA gather kernel's K step) with a block-diagonal filter.
# (32-channel rows 128 pixels wide are grouped as well: the halo kernel's 256-pixel x 64-channel tile would need a
#  two-row halo of 520 pixels, more than its DMA schedule holds, and the gather kernel wants 64-channel K steps.)
PAIR_FACTOR = {32: 2, 16: 2, 8: 4}          # 3x3: channels -> pixels per group
PAIR_FACTOR_1X1 = {32: 2, 16: 4, 8: 8}
# (kernel size, Cin, Cout) -> pixels per group, for layers whose OUTPUT is too narrow for the halo kernel's 32-channel tile: the
# outermost decoder's first 3x3 (64 -> 16 at 72x128, refine_net.py:96-131 with the skip concatenation) ran on the first-generation
# gather kernel at 1.15 ms (1.2 TB/s); as 128 -> 32 over pixel pairs it is a shape the halo kernel already serves at 36x64
PAIR_NARROW_OUT = {(3, 64, 16): 2}
_pair_index_cache = {}


def _pixel_group_index(cout, cin, ks, F, transpose, device):
    key = (cout, cin, ks, F, transpose, str(device))
    idx = _pair_index_cache.get(key)
    if idx is None:
        ar = lambda n: torch.arange(n, device=device)
        pad = (ks - 1) // 2                                 # of the plain and of the grouped convolution alike
        # output channel (po, o), taps (kh, kwg), input channel (pi, i) of the grouped filter
        no, ni = (cin, cout) if transpose else (cout, cin)
        po, o, kh, kwg, pi, i = torch.meshgrid(ar(F), ar(no), ar(ks), ar(ks), ar(F), ar(ni), indexing='ij')
        kw = F * (kwg - pad) + pi - po + pad                # original tap: input pixel minus output pixel, plus pad
        valid = (kw >= 0) & (kw < ks)
        kw = kw.clamp(0, ks - 1)
        if transpose:      # data gradient as a convolution: Wd[ci][kh][kw][co] = W[co][ks-1-kh][ks-1-kw][ci]
            src = ((i * ks + (ks - 1 - kh)) * ks + (ks - 1 - kw)) * cin + o
        else:
            src = ((o * ks + kh) * ks + kw) * cin + i
        idx = torch.where(valid, src, torch.full_like(src, cout * ks * ks * cin)).reshape(F * no, ks, ks, F * ni)
        _pair_index_cache[key] = idx
    return idx


def pixel_group_weights(ohwi, F, transpose):
    """Filter [F*Cout][k][k][F*Cin] of the convolution over groups of F pixels (see above) from the packed OHWI filter
    [Cout][k][k][Cin], k = 1 or 3; with `transpose`, the filter [F*Cin][k][k][F*Cout] of its data gradient."""
    cout, ks, _, cin = ohwi.shape
    idx = _pixel_group_index(cout, cin, ks, F, transpose, ohwi.device)
    flat = torch.cat([ohwi.reshape(-1), ohwi.new_zeros(1)])
    return flat[idx]


def _group_factors(ks):
    return PAIR_FACTOR if ks == 3 else (PAIR_FACTOR_1X1 if ks == 1 else {})


# (Cin, Cout) pairs the streaming 1x1 kernel serves (csrc/conv_1x1.h launch_conv1x1_stream): those layers are NOT grouped
STREAM_1X1 = {(16, 32), (32, 16), (16, 64), (64, 16), (32, 64), (64, 32), (32, 128), (128, 32), (64, 128), (128, 64), (16, 16),
              (32, 32), (64, 64)}


def _stream_1x1(x, c_out):
    """True when eve_conv2d_fwd / eve_conv2d_dgrad take the 1x1 convolution x [.., Cin] -> [.., c_out] on the streaming kernel."""
    k = default_kernels()
    return (x.dtype in HALF_DTYPES and (x.shape[-1], c_out) in STREAM_1X1 and x.numel() // x.shape[-1] >= 16384 and
            hasattr(k, 'dispatch_config') and bool(k.dispatch_config().conv1x1_stream))


# ... and the 3x3 layers the row-streaming kernel serves (csrc/conv_3x3s.h launch_conv3x3_stream: 64 / 128-wide images)
STREAM_3X3 = {(16, 16), (16, 32), (32, 16), (32, 32), (16, 64), (64, 16), (32, 64), (32, 128)}


def _stream_3x3(x, c_out):
    k = default_kernels()
    return (x.dtype in HALF_DTYPES and x.dim() == 4 and (x.shape[3], c_out) in STREAM_3X3 and x.shape[2] in (64, 128) and
            x.shape[1] >= 2 and x.shape[0] * x.shape[1] * x.shape[2] >= 65536 and
            x.shape[1] * x.shape[2] * max(x.shape[3], c_out) * 2 < (1 << 31) and
            hasattr(k, 'dispatch_config') and bool(getattr(k.dispatch_config(), 'conv3x3_stream', 0)))


def _group_ok(pair, x, ks, stride, pad, c_out=None):
    if pair is None or stride != 1 or pad != (ks - 1) // 2 or x.dtype not in HALF_DTYPES:
        return False
    if ks == 1 and c_out is not None and _stream_1x1(x, c_out):
        return False
    if ks == 3 and c_out is not None and _stream_3x3(x, c_out):
        return False
    wg = x.shape[2] // pair[0]
    if x.shape[2] % pair[0]:
        return False
    if ks == 3 and x.shape[3] == 32 and x.shape[2] != 128:
        return False                                                 # handled directly by the halo kernel
    return ks == 1 or (4 <= wg <= 128 and (wg & (wg - 1)) == 0)      # 3x3: the halo kernel's row widths


def _direct_grad_ok(p):
    """True for a parameter whose .grad is a persistent, contiguous (OHWI for conv weights) slice of the trainer's
    flat gradient buffer and which is used ONCE per step (train.FlatParameters sets the flag)."""
    return getattr(p, '_eve_flat_grad', False) and p.grad is not None and p.requires_grad


def _note_use(p, wanted):
    """A parameter whose gradient is written in place can be used several times per step (the per-frame contract: one
    Conv2dFn per time step).  Count the uses whose backward will run, so that the data-parallel bookkeeping is told
    only when the LAST of them has accumulated (an all-reduce launched after the first would ship a partial sum)."""
    if p is not None and wanted and getattr(p, '_eve_grad_ready', None) is not None:
        p._eve_pending_uses = getattr(p, '_eve_pending_uses', 0) + 1


def _notify_grad_ready(p):
    cb = getattr(p, '_eve_grad_ready', None)        # data-parallel bucket bookkeeping (parallel.GradSync)
    if cb is None:
        return
    n = getattr(p, '_eve_pending_uses', 0)
    if n > 1:                                       # more backward passes of this parameter are still to come
        p._eve_pending_uses = n - 1
        return
    p._eve_pending_uses = 0
    cb(p)


class Conv2dFn(torch.autograd.Function):
    """y = act(conv(x, W) + b).  x NHWC; `weight` has shape [Cout, Cin, KH, KW] (or [out, in]);
    `pack` holds the packed copies (possibly with zero-padded Cin/Cout)."""

    @staticmethod
    def forward(ctx, x, weight, bias, pack, stride, pad, epi_act, acc=None, want_stats=False):
        """want_stats: also return the InstanceNorm statistics (mean, rstd) [N, Cout, 2] of the output, or None when the
        dispatched kernel cannot form them in its epilogue (round 5: the row-streaming 3x3 kernel can) -> (y, stats)."""
        k = default_kernels()
        stats = None
        b = bias
        cout_p = pack.ohwi.shape[0]
        if bias is not None:
            b = bias.detach().float()
            if b.numel() != cout_p:
                b = torch.nn.functional.pad(b, (0, cout_p - b.numel()))
            b = b.contiguous()
        ks = pack.ohwi.shape[1]
        if acc is None and _group_ok(pack.pair_fwd, x, ks, stride, pad, cout_p):
            F, wg = pack.pair_fwd
            N, H, W, C = x.shape
            y = k.conv2d_fwd(x.view(N, H, W // F, F * C), wg, None if b is None else b.repeat(F), 1, pad, epi_act,
                             algo=pack.algo).view(N, H, W, cout_p)
        elif acc is not None:
            grouped = _group_ok(pack.pair_fwd, x, ks, stride, pad, cout_p)
            # y = acc + conv(x): accumulated in the kernel epilogue, `acc` (another branch's output) is updated in place
            # (the caller hands over the other branch's freshly produced output, which nothing else reads or saved; the
            #  result is returned as a separate tensor object over the same storage -- `acc` may itself be a view made
            #  inside another custom Function, which autograd refuses to mark dirty)
            assert epi_act == ACT_NONE and acc.is_contiguous()
            n_, h_, w_ = x.shape[0], (x.shape[1] + 2 * pad - ks) // stride + 1, (x.shape[2] + 2 * pad - ks) // stride + 1
            assert tuple(acc.shape) == (n_, h_, w_, cout_p) and acc.dtype == x.dtype, \
                'accumulate_into: expected %s, got %s' % ((n_, h_, w_, cout_p), tuple(acc.shape))
            # the kernel overwrites `acc` behind autograd's back: no Function may have SAVED it (its backward would read the
            # sum).  The contract -- the producer is a Conv2dFn with no epilogue activation, which saves no output -- is
            # checked where it can be
            prod = acc.grad_fn
            assert prod is None or getattr(prod, 'epi_act', ACT_NONE) == ACT_NONE, \
                'accumulate_into: the producer of `acc` saved its output (activation epilogue)'
            if grouped:             # F pixels as one (narrow layers): same memory, F * C channels per grouped pixel
                F, wg = pack.pair_fwd
                N, H, W, C = x.shape
                k.conv2d_fwd(x.view(N, H, W // F, F * C), wg, None if b is None else b.repeat(F), 1, pad, epi_act,
                             algo=pack.algo, accumulate_into=acc.view(N, H, W // F, F * cout_p))
            else:
                k.conv2d_fwd(x, pack.ohwi, b, stride, pad, epi_act, algo=pack.algo, accumulate_into=acc)
            y = torch.empty(0, dtype=acc.dtype, device=acc.device).set_(acc.untyped_storage(), acc.storage_offset(),
                                                                        acc.shape, acc.stride())
        elif want_stats and epi_act == ACT_NONE and x.dtype in HALF_DTYPES and hasattr(k, 'conv2d_fwd_stats'):
            y, stats = k.conv2d_fwd_stats(x, pack.ohwi, b, stride, pad, epi_act, algo=pack.algo)
        else:
            y = k.conv2d_fwd(x, pack.ohwi, b, stride, pad, epi_act, algo=pack.algo)
        ctx.pack, ctx.stride, ctx.pad, ctx.epi_act = pack, stride, pad, epi_act
        ctx.has_acc = acc is not None
        ctx.has_bias = bias is not None
        ctx.wshape = tuple(weight.shape)
        # parameters re-homed by train.FlatParameters take their gradient straight into the flat buffer
        ctx.w_direct = weight if _direct_grad_ok(weight) else None
        ctx.b_direct = bias if (bias is not None and _direct_grad_ok(bias)) else None
        _note_use(ctx.w_direct, ctx.needs_input_grad[1])
        _note_use(ctx.b_direct, ctx.needs_input_grad[2])
        ctx.save_for_backward(x, y if epi_act != ACT_NONE else None)
        if want_stats:
            if stats is not None:
                ctx.mark_non_differentiable(stats)
            return y, stats
        return y

    @staticmethod
    def backward(ctx, dy, *_):
        k = default_kernels()
        x, y = ctx.saved_tensors
        pack = ctx.pack
        dy = dy.contiguous()
        if ctx.epi_act != ACT_NONE:
            dy = k.act_bwd(dy, y, ctx.epi_act)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if _group_ok(pack.pair_dgrad, dy, pack.ohwi.shape[1], ctx.stride, ctx.pad, x.shape[3]):
                F, wg = pack.pair_dgrad
                N, H, W, C = dy.shape
                dx = k.conv2d_fwd(dy.view(N, H, W // F, F * C), wg, None, 1, ctx.pad, ACT_NONE,
                                  algo=pack.algo).view(N, H, W, x.shape[3])
            else:
                dx = k.conv2d_dgrad(dy, pack.ihwo, (x.shape[1], x.shape[2]), ctx.stride, ctx.pad, algo=pack.algo)
        # bias gradient buffer first: when the weight gradient is wanted too, its kernel sums dy's columns on the way
        dbuf, b_direct = None, False
        if ctx.has_bias and ctx.needs_input_grad[2]:
            cout_p = pack.ohwi.shape[0]
            bp = ctx.b_direct
            if bp is not None and cout_p == pack.shape_oihw[0]:
                dbuf, b_direct = bp.grad, True
            else:
                dbuf = torch.zeros((cout_p,), dtype=torch.float32, device=x.device)
                db = dbuf[:pack.shape_oihw[0]]
        db_pending = dbuf
        if ctx.needs_input_grad[1]:
            cout_p, KH, KW, cin_p = pack.ohwi.shape
            O, I = pack.shape_oihw[0], pack.shape_oihw[1]
            wp = ctx.w_direct
            if wp is not None and (cout_p, cin_p) == (O, I):
                # the wgrad kernel accumulates into the parameter's slice of the flat gradient buffer (already OHWI):
                # no temporary, no zero-fill launch, no AccumulateGrad add
                g = wp.grad
                buf = g.permute(0, 2, 3, 1) if g.dim() == 4 else g.view(O, 1, 1, I)
                k.conv2d_wgrad(x, dy, KH, KW, ctx.stride, ctx.pad, buf, algo=pack.algo, db=db_pending)
                _notify_grad_ready(wp)
            else:
                dwp = torch.zeros((cout_p, KH, KW, cin_p), dtype=torch.float32, device=x.device)
                if not _wgrad_pixel_pairs(k, x, dy, KH, ctx.stride, ctx.pad, dwp, db_pending):
                    k.conv2d_wgrad(x, dy, KH, KW, ctx.stride, ctx.pad, dwp, algo=pack.algo, db=db_pending)
                dw = dwp[:O, :, :, :I].permute(0, 3, 1, 2)          # OIHW-shaped view of OHWI memory
                if len(ctx.wshape) == 2:
                    dw = dw.reshape(ctx.wshape)
            db_pending = None
        if db_pending is not None:
            k.bias_grad(dy, db_pending)
        if b_direct:
            _notify_grad_ready(ctx.b_direct)
        return dx, dw, db, None, None, None, None, (dy if ctx.has_acc else None), None


class StemConvFn(torch.autograd.Function):
    """ResNet stem conv (7x7/2, 3 -> 64) through the dedicated bf16 kernel.  x8 is the NHWC patch batch with
    channels padded to 8 (used by the weight gradient), x_padded the [N, H+6, W+8, 4] repack the forward kernel
    reads.  The patches are inputs, so there is no data gradient."""

    @staticmethod
    def forward(ctx, x8, x_padded, weight, pack):
        y = default_kernels().stem7x7s2_fwd(x_padded, pack.ohwi)
        ctx.pack = pack
        # the weight gradient reads the packed patches (K = 7 x 8 x 4 = 224: one MFMA tile row) when they stay below the kernel's
        # 2 GiB addressing; round 4: also for patch widths other than 128 -- 256 x 256 patches (configs[4]) took the generic
        # 8-channel path, whose K = 392 falls to the first-generation weight-gradient kernel: 8.7 ms per step
        packed_ok = x_padded.dtype in HALF_DTYPES and x_padded.numel() * 2 < (1 << 31) and x_padded.shape[1] % 2 == 0 and \
            hasattr(default_kernels(), 'stem_wgrad')
        ctx.packed = packed_ok
        ctx.save_for_backward(x_padded if packed_ok else x8)
        return y

    @staticmethod
    def backward(ctx, dy):
        k = default_kernels()
        (xs,) = ctx.saved_tensors
        pack = ctx.pack
        O, I = pack.shape_oihw[0], pack.shape_oihw[1]
        if ctx.packed:
            dwp = torch.zeros((64, 7, 8, 4), dtype=torch.float32, device=xs.device)
            k.stem_wgrad(xs, dy.contiguous(), dwp)
            return None, None, dwp[:O, :, :7, :I].permute(0, 3, 1, 2), None
        cout_p, KH, KW, cin_p = pack.ohwi.shape
        dwp = torch.zeros((cout_p, KH, KW, cin_p), dtype=torch.float32, device=xs.device)
        k.conv2d_wgrad(xs, dy.contiguous(), KH, KW, 2, 3, dwp, algo=pack.algo)
        return None, None, dwp[:O, :, :, :I].permute(0, 3, 1, 2), None


class StemFusedFn(torch.autograd.Function):
    """conv1 -> bn1 (InstanceNorm2d, no affine) -> relu -> maxpool of the ResNet stem (eye_net.py:48-50,106)
    in one launch; the 64-channel convolution output is never stored.  Backward recomputes it to form
    d(conv out) and hands that to the weight-gradient kernel.  x8 / x_padded as in StemConvFn."""

    @staticmethod
    def forward(ctx, x8, x_padded, weight, pack, eps):
        y, idx, mr = default_kernels().stem_fwd_fused(x_padded, pack.ohwi, eps)
        ctx.pack = pack
        ctx.save_for_backward(x8, x_padded, y, idx, mr)
        return y

    @staticmethod
    def backward(ctx, dy):
        k = default_kernels()
        x8, x_padded, y, idx, mr = ctx.saved_tensors
        pack = ctx.pack
        dconv = k.stem_bwd_dx(x_padded, pack.ohwi, mr, dy.contiguous(), y, idx)
        cout_p, KH, KW, cin_p = pack.ohwi.shape
        dwp = torch.zeros((cout_p, KH, KW, cin_p), dtype=torch.float32, device=x8.device)
        k.conv2d_wgrad(x8, dconv, KH, KW, 2, 3, dwp, algo=pack.algo)
        O, I = pack.shape_oihw[0], pack.shape_oihw[1]
        return None, None, dwp[:O, :, :, :I].permute(0, 3, 1, 2), None, None


def conv2d(x, weight, bias, pack, stride=1, pad=0, act=ACT_NONE, acc=None, want_stats=False):
    """acc: a tensor of the output's shape that the result is added to IN PLACE (returned); no activation then.
    want_stats: -> (y, InstanceNorm statistics of y or None), see Conv2dFn."""
    if want_stats:
        return Conv2dFn.apply(x, weight, bias, pack, stride, pad, act, acc, True)
    return Conv2dFn.apply(x, weight, bias, pack, stride, pad, act, acc)


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) for the float32 tail (nn.Linear): small-tile FMA kernels, the activation derivative is
    applied while the backward kernels load dy.  x: [M, Cin_padded]; pack as for Conv2dFn (1x1)."""

    @staticmethod
    def forward(ctx, x, weight, bias, pack, act):
        k = default_kernels()
        cout_p, _, _, cin_p = pack.ohwi.shape
        b = None
        if bias is not None:
            b = bias.detach().float()
            if b.numel() != cout_p:
                b = torch.nn.functional.pad(b, (0, cout_p - b.numel()))
            b = b.contiguous()
        y = k.linear_fwd(x.contiguous(), pack.ihwo.view(cin_p, cout_p), b, act)
        ctx.pack, ctx.act, ctx.has_bias = pack, act, bias is not None
        ctx.wshape = tuple(weight.shape)
        ctx.w_direct = weight if _direct_grad_ok(weight) else None
        ctx.b_direct = bias if (bias is not None and _direct_grad_ok(bias)) else None
        _note_use(ctx.w_direct, ctx.needs_input_grad[1])
        _note_use(ctx.b_direct, ctx.needs_input_grad[2])
        ctx.save_for_backward(x, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        k = default_kernels()
        x, y = ctx.saved_tensors
        pack, act = ctx.pack, ctx.act
        cout_p, _, _, cin_p = pack.ohwi.shape
        O, I = pack.shape_oihw[0], pack.shape_oihw[1]
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = k.linear_dgrad(dy, y, act, pack.ohwi.view(cout_p, cin_p))
        want_w, want_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if want_w or want_b:
            wp, bp = ctx.w_direct, ctx.b_direct
            w_dir = wp is not None and (cout_p, cin_p) == (O, I)
            b_dir = want_b and bp is not None and cout_p == O
            dwbuf = wp.grad.view(O, I) if w_dir else torch.zeros((cout_p, cin_p), dtype=torch.float32, device=x.device)
            dbbuf = None
            if want_b:
                dbbuf = bp.grad if b_dir else torch.zeros((cout_p,), dtype=torch.float32, device=x.device)
            k.linear_wgrad(dy, y, act, x.contiguous(), dwbuf, dbbuf)
            if w_dir:
                _notify_grad_ready(wp)
            elif want_w:
                dw = dwbuf[:O, :I].reshape(ctx.wshape)
            if b_dir:
                _notify_grad_ready(bp)
            elif want_b:
                db = dbbuf[:O]
        return dx, dw, db, None, None


def _pad_bias(bias, n):
    if bias is None:
        return None
    b = bias.detach().float()
    if b.numel() != n:
        b = torch.nn.functional.pad(b, (0, n - b.numel()))
    return b.contiguous()


class _TailGrads(object):
    """Where the weight / bias gradients of a chain's layers go: straight into the flat gradient buffer when the parameter lives
    there and its pack is unpadded (train.FlatParameters), else into a zeroed temporary that is sliced for autograd."""

    def __init__(self, device):
        self.device, self.items = device, []

    def target(self, weight, bias, pack, want_w, want_b):
        cout_p, _, _, cin_p = pack.ohwi.shape
        O, I = pack.shape_oihw[0], pack.shape_oihw[1]
        w_dir = want_w and _direct_grad_ok(weight) and (cout_p, cin_p) == (O, I)
        b_dir = want_b and bias is not None and _direct_grad_ok(bias) and cout_p == O
        dw = weight.grad.view(O, I) if w_dir else torch.zeros((cout_p, cin_p), dtype=torch.float32, device=self.device)
        db = None
        if want_b and bias is not None:
            db = bias.grad if b_dir else torch.zeros((cout_p,), dtype=torch.float32, device=self.device)
        self.items.append((weight, bias, pack, want_w, want_b, w_dir, b_dir, dw, db))
        return dw, db

    def results(self):
        """-> [(dw or None, db or None)] per target() call, after the gradient launch; notifies the data-parallel bookkeeping."""
        out = []
        for weight, bias, pack, want_w, want_b, w_dir, b_dir, dw, db in self.items:
            O, I = pack.shape_oihw[0], pack.shape_oihw[1]
            gw = gb = None
            if w_dir:
                _notify_grad_ready(weight)
            elif want_w:
                gw = dw[:O, :I].reshape(tuple(weight.shape))
            if b_dir:
                _notify_grad_ready(bias)
            elif want_b and bias is not None:
                gb = db[:O]
            out.append((gw, gb))
        return out


class EyeTailLossFn(torch.autograd.Function):
    """EyeNet's tail AND its losses as one autograd node (round 4): features [2*B*T, 512] (left clips, then right) -> fc -> head-pose
    concatenation -> fc_common -> GRU cell over T -> gaze / pupil heads (eye_net.py:109-146) -> the four masked loss terms and
    their weighted sum (eve.py:234-265, 286-325).  Every step is one launch of the float32 tail kernels, called directly: the
    per-layer path builds the same chain out of eight LinearFn / GRUScanFn / EyeLossesFn nodes, and autograd's glue between them
    (bias pads, cat, pad, slices, zeros, the sum of the two heads' input gradients) is ~50 tiny ATen launches per step --
    0.2 ms of a 4.4 ms step at batch 8.  Here fc writes into the concatenation (row stride 132), the padded heads take their
    real bias length, the second head's data gradient accumulates onto the first one's, fc's data gradient reads the leading
    128 columns of the concatenation's, and the nine weight / bias gradients go straight into the flat gradient buffer in ONE
    launch (the GRU's h_prev is hs shifted by one step inside the kernel).
    Only for the product train step: one GRU cell, head-pose input, no initial state, every tail parameter living in
    train.FlatParameters (gradients are written in place, nothing is returned for them) -- eye_net.EyeNet.loss_terms_sequence
    checks that and falls back to the per-layer path otherwise.  Only d(full_loss) is propagated."""

    @staticmethod
    def forward(ctx, feats, h_left, h_right, targets, c_ang, c_l1, net, packs, B, T):
        k = default_kernels()
        cnn, cell = net.cnn_layers, net.rnn_cells[0]
        p_fc, p0, p2, p_ih, p_hh, pg0, pg2, pp0, pp2 = packs
        M, dev = feats.shape[0], feats.device
        BT = B * T
        assert M == 2 * BT and feats.dtype == torch.float32
        f32 = lambda t: t.detach()
        new = lambda n: torch.empty((M, n), dtype=torch.float32, device=dev)
        feats = feats.contiguous()
        cin0 = p0.ohwi.shape[3]                                     # 130 padded to 132
        c0 = new(cin0)
        k.linear_fwd_ex(feats, 512, p_fc.ihwo.view(512, 128), f32(cnn.fc.bias), ACT_NONE, c0)
        k.tail_head_pose(h_left.detach().float().contiguous(), h_right.detach().float().contiguous(), c0, 128)
        a1 = k.linear_fwd(c0, p0.ihwo.view(cin0, 128), f32(net.fc_common[0].bias), ACT_SELU)
        a2 = k.linear_fwd(a1, p2.ihwo.view(128, 128), f32(net.fc_common[2].bias), ACT_NONE)
        gi = k.linear_fwd(a2, p_ih.ihwo.view(128, 384), f32(cell.bias_ih), ACT_NONE)
        hs, gates, hn_pre = k.gru_scan_fwd(gi.view(2 * B, T, 384), p_hh.ihwo.view(128, 384), f32(cell.bias_hh), None)
        hs2 = hs.view(M, 128)
        g1 = k.linear_fwd(hs2, pg0.ihwo.view(128, 128), f32(net.fc_to_gaze[0].bias), ACT_SELU)
        g2 = k.linear_fwd_ex(g1, 128, pg2.ihwo.view(128, 4), None, ACT_TANH, new(4))
        p1 = k.linear_fwd(hs2, pp0.ihwo.view(128, 128), f32(net.fc_to_pupil[0].bias), ACT_SELU)
        p2_ = k.linear_fwd_ex(p1, 128, pp2.ihwo.view(128, 4), f32(net.fc_to_pupil[2].bias), ACT_RELU, new(4))
        gaze, pupil = k.tail_outputs_fwd(g2, p2_)
        tg_l, tg_r, vg_l, vg_r, tp_l, tp_r, vp_l, vp_r = targets
        terms, dg, dp = k.eye_losses((gaze[:BT].view(B, T, 2), gaze[BT:].view(B, T, 2)), (tg_l, tg_r), (vg_l, vg_r),
                                     (pupil[:BT].view(B, T), pupil[BT:].view(B, T)), (tp_l, tp_r), (vp_l, vp_r), c_ang, c_l1)
        ctx.net, ctx.packs, ctx.coeffs, ctx.BT = net, packs, (c_ang, c_l1), (B, T)
        ctx.params = EyeTailLossFn.tail_parameters(net)
        for p_ in ctx.params:
            _note_use(p_, True)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(feats, c0, a1, a2, hs, gates, hn_pre, g1, g2, p1, p2_, dg[0], dg[1], dp[0], dp[1])
        ctx.mark_non_differentiable(gaze, pupil, hs)
        return tuple(terms.unbind(0)) + (gaze, pupil, hs)

    @staticmethod
    def tail_parameters(net):
        cnn, cell = net.cnn_layers, net.rnn_cells[0]
        return [cnn.fc.weight, cnn.fc.bias, net.fc_common[0].weight, net.fc_common[0].bias, net.fc_common[2].weight,
                net.fc_common[2].bias, cell.weight_ih, cell.bias_ih, cell.weight_hh, cell.bias_hh, net.fc_to_gaze[0].weight,
                net.fc_to_gaze[0].bias, net.fc_to_gaze[2].weight, net.fc_to_pupil[0].weight, net.fc_to_pupil[0].bias,
                net.fc_to_pupil[2].weight, net.fc_to_pupil[2].bias]

    @staticmethod
    def backward(ctx, g_ang_l, g_l1_l, g_ang_r, g_l1_r, g_full, *unused):
        if any(g is not None for g in (g_ang_l, g_l1_l, g_ang_r, g_l1_r)):
            raise NotImplementedError('EyeTailLossFn propagates d(full_loss) only: use losses.eyenet_loss_terms for the separate terms')
        if g_full is None:
            return (None,) * 10
        k = default_kernels()
        feats, c0, a1, a2, hs, gates, hn_pre, g1, g2, p1, p2_, dg_l, dg_r, dp_l, dp_r = ctx.saved_tensors
        p_fc, p0, p2, p_ih, p_hh, pg0, pg2, pp0, pp2 = ctx.packs
        (w_fc, b_fc, w0, b0, w2, b2, w_ih, b_ih, w_hh, b_hh, wg0, bg0, wg2, wp0, bp0, wp2, bp2) = ctx.params
        B, T = ctx.BT
        M, dev = feats.shape[0], feats.device
        cin0 = p0.ohwi.shape[3]
        hs2 = hs.view(M, 128)
        g_full = g_full.detach().float().contiguous()
        d_g2, d_p2 = k.tail_outputs_bwd((dg_l, dg_r), (dp_l, dp_r), g_full, ctx.coeffs[0], ctx.coeffs[1])
        d_g1 = k.linear_dgrad(d_g2, g2, ACT_TANH, pg2.ohwi.view(4, 128))
        d_hs = k.linear_dgrad(d_g1, g1, ACT_SELU, pg0.ohwi.view(128, 128))
        d_p1 = k.linear_dgrad(d_p2, p2_, ACT_RELU, pp2.ohwi.view(4, 128))
        k.linear_dgrad_ex(d_p1, 128, p1, ACT_SELU, pp0.ohwi.view(128, 128), d_hs, accumulate=True)
        dgi, dgh, _ = k.gru_scan_bwd(d_hs.view(2 * B, T, 128), p_hh.ohwi.view(384, 128), None, hs, gates, hn_pre, False)
        dgi2, dgh2 = dgi.view(M, 384), dgh.view(M, 384)
        d_a2 = k.linear_dgrad(dgi2, None, ACT_NONE, p_ih.ohwi.view(384, 128))
        d_a1 = k.linear_dgrad(d_a2, None, ACT_NONE, p2.ohwi.view(128, 128))
        d_c0 = k.linear_dgrad(d_a1, a1, ACT_SELU, p0.ohwi.view(128, cin0))
        d_feats = None
        if ctx.needs_input_grad[0]:
            d_feats = torch.empty((M, 512), dtype=torch.float32, device=dev)
            k.linear_dgrad_ex(d_c0, 128, None, ACT_NONE, p_fc.ohwi.view(128, 512), d_feats)
        gw = lambda p_: p_.grad.view(p_.shape[0], -1) if p_.dim() == 2 else p_.grad
        k.linear_wgrad_batch([
            dict(dY=d_c0, X=feats, dW=gw(w_fc), db=gw(b_fc)),
            dict(dY=d_a1, Y=a1, act=ACT_SELU, X=c0, K1=w0.shape[1], dW=gw(w0), db=gw(b0)),
            dict(dY=d_a2, X=a1, dW=gw(w2), db=gw(b2)),
            dict(dY=dgi2, X=a2, dW=gw(w_ih), db=gw(b_ih)),
            dict(dY=dgh2, X=hs2, x_shift_T=T, dW=gw(w_hh), db=gw(b_hh)),
            dict(dY=d_g1, Y=g1, act=ACT_SELU, X=hs2, dW=gw(wg0), db=gw(bg0)),
            dict(dY=d_g2, Y=g2, act=ACT_TANH, X=g1, dW=gw(wg2)),
            dict(dY=d_p1, Y=p1, act=ACT_SELU, X=hs2, dW=gw(wp0), db=gw(bp0)),
            dict(dY=d_p2, Y=p2_, act=ACT_RELU, X=p1, dW=gw(wp2), db=gw(bp2))])
        for p_ in ctx.params:
            _notify_grad_ready(p_)
        return (d_feats,) + (None,) * 9


def linear(x2d, weight, bias, pack, act=ACT_NONE):
    """x2d: [M, Cin_padded] -> [M, Cout_padded].  float32 (the EyeNet tail) goes through the small-tile FMA kernels,
    anything else through the implicit-GEMM kernel as a 1x1 conv."""
    M, C = x2d.shape
    if x2d.dtype == torch.float32 and pack.ohwi.dtype == torch.float32 and pack.ihwo is not None \
            and dispatch_flag(default_kernels(), 'small_linear', 1):
        return LinearFn.apply(x2d, weight, bias, pack, act)
    y = Conv2dFn.apply(x2d.view(M, 1, 1, C), weight, bias, pack, 1, 0, act)
    return y.view(M, y.shape[-1])


def _wgrad_pixel_pairs(k, x, dy, ks, stride, pad, dwp, db):
    """Weight gradient of a layer with an 8-channel side on a >= 2 M-pixel tensor (RefineNet's first and last convolution:
    `initial.0`, 4 -> 16 padded to 8, and `final.2`, 16 -> 1 padded to 8; refine_net.py:96-131) over PIXEL PAIRS: [W][8] is byte
    for byte [W / 2][16], the grouped layer's filter gradient is a shape the band-resident kernel serves (it wants multiples of
    16 channels; the 8-channel layers fell to the gather kernel at 1.3 TB/s), and the true gradient is the scatter-sum of the
    grouped one through the same index map that builds grouped filters (adjoint of pixel_group_weights).  Accumulates into dwp
    [Cout][ks][ks][Cin] (and db); False when it does not apply."""
    N, H, W, ci = x.shape
    co = dy.shape[3]
    wg = W // 2
    if (x.dtype not in HALF_DTYPES or stride != 1 or pad != (ks - 1) // 2 or ks not in (1, 3) or min(ci, co) != 8 or max(ci, co) > 32 or
            W % 2 or wg < 32 or wg > 128 or (wg & (wg - 1)) or N * H * wg < (1 << 20) or tuple(dy.shape[:3]) != (N, H, W) or
            not hasattr(k, 'dispatch_config')):
        return False
    dwg = torch.zeros((2 * co, ks, ks, 2 * ci), dtype=torch.float32, device=x.device)
    dbg = torch.zeros((2 * co,), dtype=torch.float32, device=x.device) if db is not None else None
    k.conv2d_wgrad(x.view(N, H, wg, 2 * ci), dy.view(N, H, wg, 2 * co), ks, ks, 1, pad, dwg, db=dbg)
    idx = _pixel_group_index(co, ci, ks, 2, False, x.device)
    flat = torch.zeros((co * ks * ks * ci + 1,), dtype=torch.float32, device=x.device)
    flat.index_add_(0, idx.reshape(-1), dwg.reshape(-1))
    dwp += flat[:-1].view(co, ks, ks, ci)
    if db is not None:
        db += dbg.view(2, co).sum(0)
    return True


def _wgrad_into(k, x, dy, weight, pack, stride, pad, db=None):
    """Weight gradient of a conv: straight into the flat gradient buffer when the parameter lives there (returns
    None), else a fresh OIHW-shaped tensor for autograd.  `db` (float32 [Cout_padded], accumulated) also receives the
    bias gradient from the same kernel."""
    cout_p, KH, KW, cin_p = pack.ohwi.shape
    O, I = pack.shape_oihw[0], pack.shape_oihw[1]
    if _direct_grad_ok(weight) and (cout_p, cin_p) == (O, I):
        k.conv2d_wgrad(x, dy, KH, KW, stride, pad, weight.grad.permute(0, 2, 3, 1), algo=pack.algo, db=db)
        _notify_grad_ready(weight)
        return None
    dwp = torch.zeros((cout_p, KH, KW, cin_p), dtype=torch.float32, device=x.device)
    k.conv2d_wgrad(x, dy, KH, KW, stride, pad, dwp, algo=pack.algo, db=db)
    return dwp[:O, :, :, :I].permute(0, 3, 1, 2)


def _in_fwd(k, x, res, act, eps, want_mask=False):
    """-> (y, mean_rstd[, sign mask or None]).  The mask (one byte per 16-byte vector of y) only comes out of the
    register-resident kernel; planes too large for it take the multi-pass kernels and the backward reads y."""
    fused = k.instnorm_fwd_fused(x, None, None, res, act, eps, want_mask=True) if want_mask else \
        k.instnorm_fwd_fused(x, None, None, res, act, eps)
    if fused is not None:
        return fused
    mr = k.instnorm_stats(x, eps)
    y = k.instnorm_act_fwd(x, mr, None, None, res, act)
    return (y, mr, None) if want_mask else (y, mr)


def _in_bwd(k, dy, y, x, mr, act, want_dres, dy2=None):
    """y: the forward's output, or its sign mask (uint8) when the forward produced one."""
    mask = y if (y is not None and y.dtype == torch.uint8) else None
    out = k.instnorm_bwd_fused(dy, None if mask is not None else y, x, mr, None, act, want_dres, dy2=dy2, mask=mask) \
        if mask is not None else k.instnorm_bwd_fused(dy, y, x, mr, None, act, want_dres, dy2=dy2)
    if out is None:
        assert mask is None, 'a sign mask exists only where the fused kernel fits'
        if dy2 is not None:
            dy = k.add(dy, dy2)
        out = k.instnorm_act_bwd(dy, y, x, mr, None, act, want_dres)
    return out[0], out[1]


def _block_forward(k, x, packs, stride, eps):
    """torchvision BasicBlock with norm_layer = InstanceNorm2d (no affine), as eye_net.py:48-50 builds it:
        out = relu(IN(conv1(x)));  out = IN(conv2(out));  y = relu(out + identity),
        identity = x  or  IN(conv1x1/s(x)).   Returns y and the tensors the backward needs."""
    p1, p2, pd = packs
    a = k.conv2d_fwd(x, p1.ohwi, None, stride, 1, ACT_NONE, algo=p1.algo)
    an, mr1 = _in_fwd(k, a, None, ACT_RELU, eps)
    b = k.conv2d_fwd(an, p2.ohwi, None, 1, 1, ACT_NONE, algo=p2.algo)
    if pd is not None:
        d = k.conv2d_fwd(x, pd.ohwi, None, stride, 0, ACT_NONE, algo=pd.algo)
        idn, mrd = _in_fwd(k, d, None, ACT_NONE, eps)
    else:
        d = mrd = None
        idn = x
    y, mr2, ymask = _in_fwd(k, b, idn, ACT_RELU, eps, want_mask=True)
    # the backward needs y only for relu'(y): the sign mask (1/16 of the bytes) when the kernel produced one
    return y, (x, a, mr1, an, b, mr2, ymask if ymask is not None else y, d, mrd)


def _block_backward(k, dy, dy2, saved, weights, packs, stride, need_w):
    """Backward of _block_forward in dependency order.  The incoming gradient may arrive as two summands
    (dy + dy2: the previous fork) and the gradient of the block input is RETURNED as two summands (g, dx1) --
    residual branch and conv1 branch -- so the sum is folded into whichever kernel reads it next.
    Returns (g, dx1, dw1, dw2, dwd); weight gradients are None when they went straight into the flat buffer."""
    x, a, mr1, an, b, mr2, y, d, mrd = saved
    p1, p2, pd = packs
    w1, w2, wd = weights
    hw = (x.shape[1], x.shape[2])
    # y = relu(IN(b) + identity): db and the residual-branch gradient g = dy * relu'(y)
    db, g = _in_bwd(k, dy, y, b, mr2, ACT_RELU, True, dy2=dy2)
    dw2 = _wgrad_into(k, an, db, w2, p2, 1, 1) if need_w[1] else None
    dan = k.conv2d_dgrad(db, p2.ihwo, (an.shape[1], an.shape[2]), 1, 1, algo=p2.algo)
    da, _ = _in_bwd(k, dan, None, a, mr1, ACT_RELU, False)              # act' recomputed from a (no affine / residual)
    dw1 = _wgrad_into(k, x, da, w1, p1, stride, 1) if need_w[0] else None
    dwd = None
    dx1 = k.conv2d_dgrad(da, p1.ihwo, hw, stride, 1, algo=p1.algo)
    if pd is not None:
        dd, _ = _in_bwd(k, g, None, d, mrd, ACT_NONE, False)
        dwd = _wgrad_into(k, x, dd, wd, pd, stride, 0) if need_w[2] else None
        # the 1x1 / stride-s branch reaches one pixel in s*s: added onto conv1's data gradient in that kernel's epilogue
        # (no zero-filled full-size tensor, and the next InstanceNorm backward reads one summand instead of two)
        dx1 = k.conv2d_dgrad(dd, pd.ihwo, hw, stride, 0, algo=pd.algo, accumulate_into=dx1)
        g = None
    return (dx1, None, dw1, dw2, dwd) if g is None else (g, dx1, dw1, dw2, dwd)


class ResNetTrunkFn(torch.autograd.Function):
    """conv1/bn1/relu/maxpool (optional, fused stem) and the BasicBlock chain of torchvision's ResNet as ONE
    autograd node, so that the backward runs in dependency order with explicit buffers: every residual fork
    hands its two gradient summands to the kernel that reads them next (the add is never a launch), and the
    weight gradients go straight into the flat gradient buffer.

    apply(x, x8, x_padded, spec, eps, *weights):
      spec = (stem_pack or None, ((p1, p2, pd), stride) per block); weights = [conv1 if stem] + per block (w1, w2[, wd]).
      With a stem, x and x8 are ignored and the input is x_padded (no data gradient); without, x is the block input."""

    @staticmethod
    def forward(ctx, x, x8, x_padded, spec, eps, *weights):
        k = default_kernels()
        stem_pack, blocks = spec
        saved = []
        if stem_pack is not None:
            y, idx, mr = k.stem_fwd_fused(x_padded, stem_pack.ohwi, eps)
            saved += [x_padded, y, idx, mr]
        else:
            y = x
        for packs, stride in blocks:
            y, sv = _block_forward(k, y, packs, stride, eps)
            saved += list(sv)
        ctx.spec, ctx.weights = spec, weights
        # weight gradients written in place into the flat buffer: count this use, so that a trunk applied several times
        # per step (the per-frame contract) reports each parameter to the data-parallel bookkeeping after its LAST backward
        for i, w in enumerate(weights):
            if _direct_grad_ok(w):
                _note_use(w, ctx.needs_input_grad[5 + i])
        ctx.save_for_backward(*saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        k = default_kernels()
        stem_pack, blocks = ctx.spec
        saved = list(ctx.saved_tensors)
        weights = list(ctx.weights)
        need_w = list(ctx.needs_input_grad[5:])
        grads = [None] * len(weights)
        wpos = len(weights)
        spos = len(saved)
        d_a, d_b = dy.contiguous(), None
        for packs, stride in reversed(blocks):
            nw = 3 if packs[2] is not None else 2
            wpos -= nw
            spos -= 9
            ws = tuple(weights[wpos:wpos + nw]) + ((None,) if nw == 2 else ())
            nd = tuple(need_w[wpos:wpos + nw]) + ((False,) if nw == 2 else ())
            d_a, d_b, dw1, dw2, dwd = _block_backward(k, d_a, d_b, tuple(saved[spos:spos + 9]), ws, packs, stride, nd)
            grads[wpos], grads[wpos + 1] = dw1, dw2
            if nw == 3:
                grads[wpos + 2] = dwd
        dx = None
        if stem_pack is not None:
            x_padded, y, idx, mr = saved[:4]
            if need_w[0]:
                dwp = torch.zeros((64, 7, 8, 4), dtype=torch.float32, device=x_padded.device)
                if getattr(k, 'stem_fused_wgrad_enabled', lambda: False)():
                    # backward + weight gradient in one launch: d(conv1 out) stays in LDS (round 4)
                    k.stem_bwd_wgrad(x_padded, stem_pack.ohwi, mr, d_a, y, idx, dwp, dy_pool2=d_b)
                else:
                    dconv = k.stem_bwd_dx(x_padded, stem_pack.ohwi, mr, d_a, y, idx, dy_pool2=d_b)
                    k.stem_wgrad(x_padded, dconv, dwp)          # straight from the packed patches (no 8-channel copy)
                O, I = stem_pack.shape_oihw[0], stem_pack.shape_oihw[1]
                grads[0] = dwp[:O, :, :7, :I].permute(0, 3, 1, 2)
        elif ctx.needs_input_grad[0]:
            dx = k.add(d_a, d_b) if d_b is not None else d_a
        return (dx, None, None, None, None) + tuple(grads)


def _affine_direct(k, gamma_p, beta_p):
    """True when an affine InstanceNorm's (d gamma, d beta) are ADDED IN PLACE into the trainer's flat gradient buffer."""
    return (gamma_p is not None and beta_p is not None and _direct_grad_ok(gamma_p) and _direct_grad_ok(beta_p) and
            hasattr(k, 'sum_rows_pairs') and gamma_p.grad.is_contiguous() and beta_p.grad.is_contiguous())


def _note_affine_use(k, gamma_p, beta_p, want_gamma, want_beta):
    """Forward-side twin of _affine_grads' in-place route: count this use (see _note_use), so that a parameter pair that
    normalises several times per step -- the per-frame contract: RefineNet.forward() once per time step -- reports to the
    data-parallel bookkeeping after its LAST backward, not its first."""
    if _affine_direct(k, gamma_p, beta_p):
        _note_use(gamma_p, want_gamma)
        _note_use(beta_p, want_beta)


def _affine_grads(k, sums, gamma_p, beta_p):
    """(d gamma, d beta) of an affine InstanceNorm from the backward kernel's per-plane partials `sums` [N, C, 2] = (d beta, d
    gamma): added in place into the flat gradient buffer when both parameters live there (-> (None, None); the autograd
    route returned two strided views and paid one accumulate launch per parameter: ~80 launches per configs[2] step), else two
    fresh vectors for autograd."""
    if _affine_direct(k, gamma_p, beta_p):
        if sums.dim() == 3:
            k.sum_rows_pairs(sums, beta_p.grad, gamma_p.grad)
        else:
            s = k.sum_rows(sums)
            gamma_p.grad.add_(s[:, 1])
            beta_p.grad.add_(s[:, 0])
        _notify_grad_ready(gamma_p)
        _notify_grad_ready(beta_p)
        return None, None
    s = k.sum_rows(sums)                # [C, 2]: N-reduction of the per-plane partials, fixed order
    return s[:, 1], s[:, 0]


class InstNormActFn(torch.autograd.Function):
    """y = act(gamma * IN(x) + beta + res);  gamma/beta/res optional.  eps 1e-5, biased variance."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, act, eps, stats=None):
        """stats: (mean, rstd) [N, C, 2] of x already formed by its producer (ops.conv2d(want_stats=True)): the statistics pass
        of the multi-pass path is skipped."""
        k = default_kernels()
        g = gamma.detach().float().contiguous() if gamma is not None else None
        b = beta.detach().float().contiguous() if beta is not None else None
        fused = k.instnorm_fwd_fused(x, g, b, res, act, eps) if stats is None else None     # one launch when the plane fits in registers
        if fused is not None:
            y, mr = fused
        else:
            mr = stats if stats is not None else k.instnorm_stats(x, eps)
            y = k.instnorm_act_fwd(x, mr, g, b, res, act)
        ctx.act, ctx.has_res, ctx.has_affine = act, res is not None, gamma is not None
        ctx.affine_params = (gamma, beta)
        _note_affine_use(k, gamma, beta, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        # the backward needs y only to evaluate act'; without a residual it recomputes that from x (the register-resident
        # kernel does so only when there is no affine either)
        need_y = act != ACT_NONE and res is not None       # (else both backward kernels recompute act' from x: gamma / beta given)
        ctx.save_for_backward(x, y if need_y else None, mr, g, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        k = default_kernels()
        x, y, mr, g, b = ctx.saved_tensors
        want_dres = ctx.has_res and ctx.needs_input_grad[3]
        out = k.instnorm_bwd_fused(dy.contiguous(), y, x, mr, g, ctx.act, want_dres, beta=b)
        if out is None:
            out = k.instnorm_act_bwd(dy.contiguous(), y, x, mr, g, ctx.act, want_dres, beta=b)
        dx, dres, sums = out
        dgamma = dbeta = None
        if ctx.has_affine:
            dgamma, dbeta = _affine_grads(k, sums, *ctx.affine_params)
        return dx, dgamma, dbeta, dres, None, None, None


def instnorm_act(x, gamma=None, beta=None, res=None, act=ACT_NONE, eps=1e-5, stats=None):
    if stats is not None:
        return InstNormActFn.apply(x, gamma, beta, res, act, eps, stats)
    return InstNormActFn.apply(x, gamma, beta, res, act, eps)


class InstNormActSkipFn(torch.autograd.Function):
    """(act(gamma * IN(x) + beta), x): the input of an identity-skip BasicBlock (refine_net.py:35-67, ic == oc) feeds `layers` AND the
    block's final add.  Returned as the second output of this node, the skip's gradient arrives in the same backward as the
    normalised branch's and the InstanceNorm backward adds it in its epilogue (`dx_add`) -- autograd's gradient-fork add (three
    passes over the block input) is not launched.  The second output aliases x."""

    @staticmethod
    def forward(ctx, x, gamma, beta, act, eps):
        k = default_kernels()
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        fused = k.instnorm_fwd_fused(x, g, b, None, act, eps)
        if fused is not None:
            y, mr = fused
        else:
            mr = k.instnorm_stats(x, eps)
            y = k.instnorm_act_fwd(x, mr, g, b, None, act)
        ctx.act = act
        ctx.affine_params = (gamma, beta)
        _note_affine_use(k, gamma, beta, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        ctx.save_for_backward(x, mr, g, b)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        k = default_kernels()
        x, mr, g, b = ctx.saved_tensors
        if dy is None:
            return dskip, None, None, None, None
        add = dskip.contiguous() if dskip is not None else None
        out = k.instnorm_bwd_fused(dy.contiguous(), None, x, mr, g, ctx.act, False, beta=b, dx_add=add)
        if out is None:
            out = k.instnorm_act_bwd(dy.contiguous(), None, x, mr, g, ctx.act, False, beta=b, dx_add=add)
        dx, _, sums = out
        dgamma, dbeta = _affine_grads(k, sums, *ctx.affine_params)
        return dx, dgamma, dbeta, None, None


class InstNormAct2Fn(torch.autograd.Function):
    """(a, b) = act(gamma_h * IN(cat(xs)) + beta_h) for two affine heads h over the channel-concatenation of 1-2 NHWC sources.
    RefineNet's pre-activation BasicBlock (refine_net.py:46-47,59-60) normalises its input twice -- `layers.0` and
    `skip_layer.0` -- and the decoder's input is a torch.cat (refine_net.py:125-126): InstanceNorm statistics are per channel,
    so each source is normalised on its own, read once for both heads and written straight into its channel range of the
    two outputs; the backward forms the fork's gradient sum in registers (kernels: eve_instnorm_act2_{fwd,bwd})."""

    @staticmethod
    def forward(ctx, act, eps, gamma_a, beta_a, gamma_b, beta_b, *xs):
        k = default_kernels()
        f32 = lambda t: t.detach().float().contiguous()
        ga, ba, gb, bb = f32(gamma_a), f32(beta_a), f32(gamma_b), f32(beta_b)
        ctot = sum(x.shape[-1] for x in xs)
        assert ga.numel() == ba.numel() == gb.numel() == bb.numel() == ctot, \
            'affine parameters must cover the concatenated channels exactly (%d), got %d' % (ctot, ga.numel())
        assert all(x.is_contiguous() and x.dtype == xs[0].dtype and x.shape[:3] == xs[0].shape[:3] for x in xs)
        mrs = [k.instnorm_stats(x, eps) for x in xs]
        out_a, out_b = k.instnorm_act2_fwd(xs, mrs, ga, ba, gb, bb, act)      # both sources, both heads: one launch
        ctx.act, ctx.n = act, len(xs)
        ctx.affine_params = (gamma_a, beta_a, gamma_b, beta_b)
        _note_affine_use(k, gamma_a, beta_a, ctx.needs_input_grad[2], ctx.needs_input_grad[3])
        _note_affine_use(k, gamma_b, beta_b, ctx.needs_input_grad[4], ctx.needs_input_grad[5])
        ctx.save_for_backward(ga, ba, gb, bb, *xs, *mrs)
        return out_a, out_b

    @staticmethod
    def backward(ctx, d_a, d_b):
        k = default_kernels()
        ga, ba, gb, bb = ctx.saved_tensors[:4]
        xs, mrs = ctx.saved_tensors[4:4 + ctx.n], ctx.saved_tensors[4 + ctx.n:]
        if d_a is None or d_b is None:                      # one head unused downstream: its gradient is zero
            ref = d_a if d_a is not None else d_b
            d_a = d_a if d_a is not None else torch.zeros_like(ref)
            d_b = d_b if d_b is not None else torch.zeros_like(ref)
        dxs, sa, sb = k.instnorm_act2_bwd(d_a.contiguous(), d_b.contiguous(), xs, mrs, ga, ba, gb, bb, ctx.act)
        pa, pb = ctx.affine_params[:2], ctx.affine_params[2:]
        dga, dba = _affine_grads(k, sa, *pa)
        dgb, dbb = _affine_grads(k, sb, *pb)
        return (None, None, dga, dba, dgb, dbb) + tuple(dxs)


def instnorm_act2(xs, gamma_a, beta_a, gamma_b, beta_b, act, eps=1e-5):
    return InstNormAct2Fn.apply(act, eps, gamma_a, beta_a, gamma_b, beta_b, *xs)


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return default_kernels().add(a, b)

    @staticmethod
    def backward(ctx, d):
        return d, d


def add(a, b):
    return AddFn.apply(a, b)


class MaxPool3x3s2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y, idx = default_kernels().maxpool3x3s2_fwd(x)
        ctx.in_hw = (x.shape[1], x.shape[2])
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return default_kernels().maxpool3x3s2_bwd(dy.contiguous(), idx, ctx.in_hw)


class InReluMaxPoolFn(torch.autograd.Function):
    """maxpool3x3s2(relu(IN(x))), no affine: the ResNet stem tail in two launches (stats + fused pass)."""

    @staticmethod
    def forward(ctx, x, eps):
        k = default_kernels()
        mr = k.instnorm_stats(x, eps)
        y, idx = k.in_relu_maxpool_fwd(x, mr)
        ctx.save_for_backward(x, mr, y, idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mr, y, idx = ctx.saved_tensors
        return default_kernels().in_relu_maxpool_bwd(dy.contiguous(), y, idx, x, mr), None


class AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.hw = (x.shape[1], x.shape[2])
        return default_kernels().avgpool_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        return default_kernels().avgpool_bwd(dy.contiguous(), ctx.hw)


class AvgPoolF32Fn(torch.autograd.Function):
    """AvgPoolFn followed by a cast to float32 (and the cast back in the backward) as one launch each way; same bits."""
    @staticmethod
    def forward(ctx, x):
        ctx.hw, ctx.src = (x.shape[1], x.shape[2]), x.dtype
        return default_kernels().avgpool_fwd_f32(x)

    @staticmethod
    def backward(ctx, dy):
        return default_kernels().avgpool_bwd_f32(dy.contiguous(), ctx.hw, ctx.src)


class AdaptiveMaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_hw):
        y, idx = default_kernels().adaptive_maxpool_fwd(x, out_hw)
        ctx.in_hw = (x.shape[1], x.shape[2])
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return default_kernels().adaptive_maxpool_bwd(dy.contiguous(), idx, ctx.in_hw), None


class PoolForkFn(torch.autograd.Function):
    """(AdaptiveMaxPool2d(x), x): the encoder output of a RefineNet level feeds the pool to the next level AND the decoder's skip
    connection (refine_net.py:103-126).  As two consumers of one tensor autograd sums their gradients with an add launch (three
    passes over a 70-566 MB tensor per level); here both arrive in ONE backward and the pool's adjoint adds the skip gradient
    in its epilogue.  The second output aliases x."""

    @staticmethod
    def forward(ctx, x, out_hw):
        y, idx = default_kernels().adaptive_maxpool_fwd(x, out_hw)
        ctx.in_hw = (x.shape[1], x.shape[2])
        ctx.save_for_backward(idx)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        (idx,) = ctx.saved_tensors
        k = default_kernels()
        if dy is None:
            return dskip, None
        return k.adaptive_maxpool_bwd(dy.contiguous(), idx, ctx.in_hw, add=dskip.contiguous() if dskip is not None else None), None


class BilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_hw):
        ctx.in_hw = (x.shape[1], x.shape[2])
        return default_kernels().bilinear_fwd(x, out_hw)

    @staticmethod
    def backward(ctx, dy):
        return default_kernels().bilinear_bwd(dy.contiguous(), ctx.in_hw), None


class ToNHWCFn(torch.autograd.Function):
    """float NCHW (the reference's layout at the module boundary) -> NHWC compute dtype, channels
    zero-padded to a 16-byte multiple."""

    @staticmethod
    def forward(ctx, x_nchw, dtype, cpad):
        ctx.C = x_nchw.shape[1]
        return default_kernels().nchw_to_nhwc(x_nchw.contiguous().float(), dtype, cpad)

    @staticmethod
    def backward(ctx, dy):
        return default_kernels().nhwc_to_nchw(dy.contiguous(), ctx.C), None, None


class FromNHWCFn(torch.autograd.Function):
    """NHWC compute dtype -> float NCHW, keeping the first C channels."""

    @staticmethod
    def forward(ctx, x, C):
        ctx.dtype, ctx.cpad = x.dtype, x.shape[3]
        return default_kernels().nhwc_to_nchw(x, C)

    @staticmethod
    def backward(ctx, dy):
        return default_kernels().nchw_to_nhwc(dy.contiguous().float(), ctx.dtype, ctx.cpad), None


class CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return default_kernels().cast(x.contiguous(), dtype)

    @staticmethod
    def backward(ctx, dy):
        return default_kernels().cast(dy.contiguous(), ctx.src), None


def cast(x, dtype):
    return x if x.dtype == dtype else CastFn.apply(x, dtype)


class GRUScanFn(torch.autograd.Function):
    """hs[s, t] = GRUCell step over t for every sequence s, given gi = W_ih x + b_ih for all steps.
    dW_hh / db_hh come from the batched wgrad / bias-grad kernels over all (s, t)."""

    @staticmethod
    def forward(ctx, gi, w_hh, b_hh, h0):
        k = default_kernels()
        whh = w_hh.detach().float().contiguous()
        whh_t = whh.t().contiguous()
        bhh = b_hh.detach().float().contiguous()
        h0c = h0.detach().float().contiguous() if h0 is not None else None
        hs, gates, hn_pre = k.gru_scan_fwd(gi.contiguous(), whh_t, bhh, h0c)
        ctx.has_h0 = h0 is not None
        ctx.save_for_backward(whh, h0c, hs, gates, hn_pre)
        return hs

    @staticmethod
    def backward(ctx, dhs):
        k = default_kernels()
        whh, h0, hs, gates, hn_pre = ctx.saved_tensors
        S, T, H = hs.shape
        dgi, dgh, dh0 = k.gru_scan_bwd(dhs.contiguous().float(), whh, h0, hs, gates, hn_pre,
                                       ctx.has_h0 and ctx.needs_input_grad[3])
        dw, db = _recurrent_param_grads(k, dgh.view(S * T, 3 * H), _shift_states(h0, hs), ctx.needs_input_grad[1],
                                        ctx.needs_input_grad[2])
        return dgi, dw, db, dh0


def _recurrent_param_grads(k, dpre, h_prev, need_w, need_b):
    """dW_hh = dpre^T . h_prev and db_hh = column sums of dpre, over all (sequence, step) rows."""
    R, GH = dpre.shape
    H = h_prev.shape[1]
    dw = torch.zeros((GH, H), dtype=torch.float32, device=dpre.device) if need_w or need_b else None
    db = torch.zeros((GH,), dtype=torch.float32, device=dpre.device) if need_b else None
    if dw is not None:
        k.linear_wgrad(dpre, None, ACT_NONE, h_prev, dw, db)
    return (dw if need_w else None), db


def _shift_states(first, hs):
    S, T, H = hs.shape
    if first is None:
        first = torch.zeros((S, H), dtype=torch.float32, device=hs.device)
    return torch.cat([first.unsqueeze(1), hs[:, :-1]], dim=1).reshape(S * T, H).contiguous()


class RNNScanFn(torch.autograd.Function):
    """hs[s, t] = nn.RNNCell (tanh) over t, given gi = W_ih x + b_ih for all steps (eye_net.py:62-63)."""

    @staticmethod
    def forward(ctx, gi, w_hh, b_hh, h0):
        k = default_kernels()
        whh = w_hh.detach().float().contiguous()
        h0c = h0.detach().float().contiguous() if h0 is not None else None
        hs = k.rnn_scan_fwd(gi.contiguous(), whh.t().contiguous(), b_hh.detach().float().contiguous(), h0c)
        ctx.has_h0 = h0 is not None
        ctx.save_for_backward(whh, h0c, hs)
        return hs

    @staticmethod
    def backward(ctx, dhs):
        k = default_kernels()
        whh, h0, hs = ctx.saved_tensors
        S, T, H = hs.shape
        dpre, dh0 = k.rnn_scan_bwd(dhs.contiguous().float(), whh, hs, ctx.has_h0 and ctx.needs_input_grad[3])
        dw, db = _recurrent_param_grads(k, dpre.view(S * T, H), _shift_states(h0, hs), ctx.needs_input_grad[1],
                                        ctx.needs_input_grad[2])
        return dpre, dw, db, dh0


class LSTMScanFn(torch.autograd.Function):
    """(hs, cs)[s, t] = nn.LSTMCell over t (gate order i, f, g, o), given gi = W_ih x + b_ih (eye_net.py:64-66)."""

    @staticmethod
    def forward(ctx, gi, w_hh, b_hh, h0, c0):
        k = default_kernels()
        whh = w_hh.detach().float().contiguous()
        h0c = h0.detach().float().contiguous() if h0 is not None else None
        c0c = c0.detach().float().contiguous() if c0 is not None else None
        hs, cs, gates = k.lstm_scan_fwd(gi.contiguous(), whh.t().contiguous(), b_hh.detach().float().contiguous(), h0c, c0c)
        ctx.has_0 = h0 is not None
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(whh, h0c, c0c, hs, cs, gates)
        return hs, cs

    @staticmethod
    def backward(ctx, dhs, dcs):
        k = default_kernels()
        whh, h0, c0, hs, cs, gates = ctx.saved_tensors
        S, T, H = hs.shape
        if dhs is None:
            dhs = torch.zeros_like(hs)
        want0 = ctx.has_0 and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        dpre, dh0, dc0 = k.lstm_scan_bwd(dhs.contiguous().float(), dcs.contiguous().float() if dcs is not None else None,
                                         whh, c0, hs, cs, gates, want0)
        dw, db = _recurrent_param_grads(k, dpre.view(S * T, 4 * H), _shift_states(h0, hs), ctx.needs_input_grad[1],
                                        ctx.needs_input_grad[2])
        return dpre, dw, db, dh0, dc0


class CGRUGates1Fn(torch.autograd.Function):
    """(ru, rh) = (sigmoid(g1), sigmoid(g1[..., :C]) * h)   -- common.py:410-411"""

    @staticmethod
    def forward(ctx, g1, h):
        ru, rh = default_kernels().cgru_gates1(g1, h)
        ctx.save_for_backward(ru, h)
        return ru, rh

    @staticmethod
    def backward(ctx, dru, drh):
        ru, h = ctx.saved_tensors
        dg1, dh = default_kernels().cgru_gates1_bwd(drh.contiguous(), dru.contiguous(), ru, h)
        return dg1, dh


def _bias_grad_into(k, dy, bias):
    """Bias gradient of a convolution whose output gradient is `dy` [..., C]: straight into the flat gradient buffer when the
    parameter lives there (returns None), else a fresh float32 vector."""
    if _direct_grad_ok(bias):
        k.bias_grad(dy, bias.grad)
        _notify_grad_ready(bias)
        return None
    db = torch.zeros((dy.shape[-1],), dtype=torch.float32, device=dy.device)
    k.bias_grad(dy, db)
    return db


def _float_banks(weight, pack):
    """(OHWI, IHWO) float32 copies of a convolution's filter bank for the float32 clip scans (csrc/cell_scan_f32.hip): the
    pack's own tensors in the float32 instantiation, otherwise formed from the parameter (295-590 KB)."""
    if pack is not None and pack.ohwi.dtype == torch.float32 and pack.ihwo is not None:
        return pack.ohwi, pack.ihwo
    w = weight.detach().float()
    return w.permute(0, 2, 3, 1).contiguous(), w.permute(1, 2, 3, 0).contiguous()


class CRNNScanFn(torch.autograd.Function):
    """hs[:, t] = CRNNCell(xs[:, t], hs[:, t-1]) = tanh(conv3x3([x_t | h_{t-1}]) + b) for the whole clip in ONE launch, and the
    frame-reversed backward in one more (kernels.crnn_scan_{fwd,bwd}, float32: csrc/cell_scan_f32.hip; common.py:331-352 applied
    per frame by refine_net.py:132-176).  The weight and bias gradients are one batched launch each over all T*B frames.
    xs float32 [B, T, 5, 8, 64] (16-bit callers convert: the bottleneck is 2 560 values per frame)."""

    @staticmethod
    def forward(ctx, xs, w, b, h0, pack):
        k = default_kernels()
        h0c = h0.detach().contiguous() if h0 is not None else None
        w_ohwi, w_ihwo = _float_banks(w, pack)
        xs = xs.contiguous()
        hs, hs_tm = k.crnn_scan_fwd(xs, h0c, w_ohwi, b.detach().float().contiguous())
        ctx.pack = pack
        ctx.params = (w, b)
        ctx.has_h0 = h0 is not None
        ctx.save_for_backward(xs, h0c, hs_tm, w_ihwo)
        return hs

    @staticmethod
    def backward(ctx, dhs):
        k = default_kernels()
        xs, h0, hs_tm, w_ihwo = ctx.saved_tensors
        w, b = ctx.params
        B, T, H, W, C = xs.shape
        need = ctx.needs_input_grad
        dhs_tm = dhs.float().transpose(0, 1).contiguous()
        want_dh0 = bool(ctx.has_h0 and need[3])
        dpre, dxs_tm, dh0 = k.crnn_scan_bwd(dhs_tm, hs_tm, w_ihwo, want_dh0)
        dxs = dxs_tm.transpose(0, 1).contiguous() if need[0] else None
        first = h0 if h0 is not None else torch.zeros_like(hs_tm[0])
        h_prev_all = torch.cat([first.unsqueeze(0), hs_tm[:-1]], dim=0)
        cat1 = torch.cat([xs.transpose(0, 1), h_prev_all], dim=-1).reshape(T * B, H, W, 2 * C)
        gpre = dpre.view(T * B, H, W, C)
        dw = _wgrad_into(k, cat1, gpre, w, ctx.pack, 1, 1) if need[1] else None
        db = _bias_grad_into(k, gpre, b) if need[2] else None
        return dxs, dw, db, dh0, None


def clstm_scan(xs, weight, bias, pack, h0=None, c0=None):
    """CLSTMCell over a clip in one launch, forward only (kernels.clstm_scan_fwd; common.py:355-385): the reference stores
    the (h, c) tuple and never feeds it to the decoder (refine_net.py:168-174), so nothing is differentiated.
    xs [B, T, 5, 8, 64] any dtype -> (hs, cs) float32 [B, T, 5, 8, 64]."""
    k = default_kernels()
    with torch.no_grad():
        w_ohwi, _ = _float_banks(weight, pack)
        f = lambda t: None if t is None else t.detach().float().contiguous()
        return k.clstm_scan_fwd(xs.detach().float().contiguous(), f(h0), f(c0), w_ohwi, bias.detach().float().contiguous())


class CGRUScanFn(torch.autograd.Function):
    """hs[:, t] = CGRUCell(xs[:, t], hs[:, t-1]) for the whole clip in ONE launch (kernels.cgru_scan_fwd: hidden state
    resident in LDS, both gate convolutions and their sigmoid / tanh / blend epilogues fused; common.py:388-415 applied
    per frame by refine_net.py:132-176).  The backward is one persistent launch as well (kernels.cgru_scan_bwd: frames in
    reverse, gate gradients + both data-gradient GEMMs + the carry into the previous state fused; eve_dispatch_config.cgru_scan = 2
    selects the per-frame kernels on time-major tensors); the two weight gradients and bias gradients are ONE batched
    launch each over all T*B frames."""

    @staticmethod
    def forward(ctx, xs, w1, b1, w2, b2, h0, p1, p2):
        k = default_kernels()
        h0c = h0.detach().contiguous() if h0 is not None else None
        hs, hs_tm, ru, rh, og = k.cgru_scan_fwd(xs.contiguous(), h0c, p1.ohwi, b1.detach().float().contiguous(), p2.ohwi,
                                                b2.detach().float().contiguous())
        ctx.packs = (p1, p2)
        ctx.params = (w1, b1, w2, b2)
        ctx.has_h0 = h0 is not None
        ctx.save_for_backward(xs, h0c, hs_tm, ru, rh, og)
        return hs

    @staticmethod
    def backward(ctx, dhs):
        k = default_kernels()
        xs, h0, hs_tm, ru, rh, og = ctx.saved_tensors
        p1, p2 = ctx.packs
        w1, b1, w2, b2 = ctx.params
        B, T, H, W, C = xs.shape
        need = ctx.needs_input_grad
        dhs_tm = dhs.transpose(0, 1).contiguous()                       # [T, B, ...]
        xs_tm = xs.transpose(0, 1).contiguous()
        first = h0 if h0 is not None else torch.zeros_like(xs_tm[0])
        want_dh0 = bool(ctx.has_h0 and need[5])
        if hasattr(k, 'cgru_scan_bwd') and dispatch_flag(k, 'cgru_scan', 1) != 2:
            # the whole frame-reversed recursion in one persistent launch (kernels.cgru_scan_bwd): gate gradients, both
            # data-gradient GEMMs, the carry into the previous state; gradients of the two pre-activations come back for
            # the batched weight / bias gradients below
            dg1_all, dg2_all, dxs_tm, dh0 = k.cgru_scan_bwd(dhs_tm, ru, og, hs_tm, h0, p1.ihwo, p2.ihwo, want_dh0)
            dxs = dxs_tm.transpose(0, 1).contiguous()
        else:
            dcat1_all = torch.empty((T, B, H, W, 2 * C), dtype=xs.dtype, device=xs.device)      # d[x | h] per frame
            dcat2_all = torch.empty((T, B, H, W, 2 * C), dtype=xs.dtype, device=xs.device)      # d[r*h | x] per frame
            dg1_all = torch.empty((T, B, H, W, 2 * C), dtype=xs.dtype, device=xs.device)
            dg2_all = torch.empty((T, B, H, W, C), dtype=xs.dtype, device=xs.device)
            carry = None
            for t in range(T - 1, -1, -1):
                dhn = dhs_tm[t] if carry is None else k.add(dhs_tm[t], carry)
                h_prev = hs_tm[t - 1] if t > 0 else first
                dg2, dru, dh_a = k.cgru_gates2_bwd(dhn, ru[t], h_prev, og[t])
                dcat2 = k.conv2d_dgrad(dg2, p2.ihwo, (H, W), 1, 1, algo=p2.algo)
                dg1, dh_b = k.cgru_gates1_bwd(dcat2[..., :C].contiguous(), dru, ru[t], h_prev)
                dcat1 = k.conv2d_dgrad(dg1, p1.ihwo, (H, W), 1, 1, algo=p1.algo)
                carry = k.add(k.add(dh_a, dh_b), dcat1[..., C:].contiguous())
                dcat1_all[t], dcat2_all[t], dg1_all[t], dg2_all[t] = dcat1, dcat2, dg1, dg2
            # d(xs) = x-halves of the two concatenated-input gradients, for all frames at once
            dxs = (dcat1_all[..., :C] + dcat2_all[..., C:]).transpose(0, 1).contiguous()
            dh0 = carry if want_dh0 else None
        # weight / bias gradients: one launch each over all T*B frames
        h_prev_all = torch.cat([first.unsqueeze(0), hs_tm[:-1]], dim=0)
        cat1 = torch.cat([xs_tm, h_prev_all], dim=-1).view(T * B, H, W, 2 * C)
        cat2 = torch.cat([rh, xs_tm], dim=-1).view(T * B, H, W, 2 * C)
        g1f, g2f = dg1_all.view(T * B, H, W, 2 * C), dg2_all.view(T * B, H, W, C)

        dw1 = _wgrad_into(k, cat1, g1f, w1, p1, 1, 1) if need[1] else None
        db1 = _bias_grad_into(k, g1f, b1) if need[2] else None
        dw2 = _wgrad_into(k, cat2, g2f, w2, p2, 1, 1) if need[3] else None
        db2 = _bias_grad_into(k, g2f, b2) if need[4] else None
        return dxs, dw1, db1, dw2, db2, dh0, None, None


class CGRUGates2Fn(torch.autograd.Function):
    """h' = (1 - u) * tanh(g2) + u * h   -- common.py:413-414.  `ru` only contributes through u."""

    @staticmethod
    def forward(ctx, g2, ru, h):
        o, hnew = default_kernels().cgru_gates2(g2, ru, h)
        ctx.save_for_backward(ru, h, o)
        return hnew

    @staticmethod
    def backward(ctx, dhnew):
        ru, h, o = ctx.saved_tensors
        dg2, dru, dh = default_kernels().cgru_gates2_bwd(dhnew.contiguous(), ru, h, o)
        return dg2, dru, dh


class VectorTermsFn(torch.autograd.Function):
    """The validity-masked [B, T, D <= 3] loss / metric terms of EVE.calculate_losses_and_metrics (eve.py:286-439) in one launch
    (kernels.vector_terms).  apply(spec, *preds): spec = tuple of (kind, target, validity) per prediction; returns one scalar per
    term.  Gradients w.r.t. the predictions for 'mse', 'l1', 'angular' (the caller keeps 'euclidean' terms whose prediction
    requires a gradient on the tensor-expression path)."""

    @staticmethod
    def forward(ctx, spec, *preds):
        k = default_kernels()
        need = [bool(n) for n in ctx.needs_input_grad[1:]]
        items = [(kind, p.detach(), tgt, val) for (kind, tgt, val), p in zip(spec, preds)]
        out, dps = k.vector_terms(items, need)
        ctx.set_materialize_grads(False)
        ctx.n = len(preds)
        ctx.has = [d is not None for d in dps]
        ctx.save_for_backward(*[d for d in dps if d is not None])
        return tuple(out.unbind(0))

    @staticmethod
    def backward(ctx, *grads):
        saved = iter(ctx.saved_tensors)
        res = [None]
        for i in range(ctx.n):
            d = next(saved) if ctx.has[i] else None
            res.append(d * grads[i] if (d is not None and grads[i] is not None) else None)
        return tuple(res)


# ---- gaze geometry / heat-maps / soft-argmax around the two networks (kernels in csrc/gaze_geometry.hip) -------------
class GazeToPoGFn(torch.autograd.Function):
    """(g_out, PoG_mm, PoG_px) of a flat batch of frames: to_screen_coordinates (models/common.py:157-187), after
    apply_offset_augmentation (:190-229) when `kappa` is given.  Differentiable w.r.t. the gaze angles only (the other
    inputs are data in EVE); the kernel returns the 2x2 Jacobians, so the backward is one tiny launch."""

    @staticmethod
    def forward(ctx, g, origin, R, inv_cam, ppm, screen, head_R, kappa):
        g_out, mm, px, jac = default_kernels().gaze_to_pog(g, origin, R, inv_cam, ppm, screen, head_R, kappa)
        ctx.save_for_backward(jac)
        ctx.set_materialize_grads(False)
        return g_out, mm, px

    @staticmethod
    def backward(ctx, dg_out, dmm, dpx):
        jac, = ctx.saved_tensors
        if dg_out is None and dmm is None and dpx is None:
            return (None,) * 8
        f = lambda t: None if t is None else t.contiguous()
        return (default_kernels().gaze_to_pog_bwd(jac, f(dg_out), f(dmm), f(dpx)),) + (None,) * 7


class MakeHeatmapsFn(torch.autograd.Function):
    """Gaussian heat-maps [N,1,H,W] around centres [N,2] (px) -- models/common.py:236-255."""

    @staticmethod
    def forward(ctx, centres_px, sigma, hw, screen):
        ctx.save_for_backward(centres_px)
        ctx.args = (sigma, screen)
        return default_kernels().make_heatmaps(centres_px, sigma, hw, screen)

    @staticmethod
    def backward(ctx, dout):
        centres, = ctx.saved_tensors
        sigma, screen = ctx.args
        return default_kernels().make_heatmaps_bwd(centres, sigma, screen, dout.contiguous()), None, None, None


class SoftArgmaxFn(torch.autograd.Function):
    """PoG (px) [N,2] of heat-maps [N,1,H,W] -- models/common.py:304-333."""

    @staticmethod
    def forward(ctx, heat, screen):
        heat = heat.contiguous()
        px, stats = default_kernels().soft_argmax_fwd(heat, screen)
        ctx.save_for_backward(heat, stats)
        ctx.screen = screen
        return px

    @staticmethod
    def backward(ctx, dpx):
        heat, stats = ctx.saved_tensors
        return default_kernels().soft_argmax_bwd(heat, stats, dpx.contiguous(), ctx.screen), None


# ---- RefineNet output head and the heat-map losses (kernels in csrc/heatmap_loss.hip) ---------------------------------
class HeatmapHeadFn(torch.autograd.Function):
    """heatmap_final [N,1,H,W] float = sigmoid(channel 0 of the last convolution's NHWC logits), evaluated in float
    whatever the compute dtype (refine_net.py:221-223,255): the map feeds a float BCE and a softmax(100 h)."""

    @staticmethod
    def forward(ctx, logits):
        y = default_kernels().heatmap_head_fwd(logits.contiguous())
        ctx.meta = (logits.dtype, logits.shape[3])
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        return default_kernels().heatmap_head_bwd(dy, y, *ctx.meta)


class HeatmapLossFn(torch.autograd.Function):
    """Validity-masked heat-map loss over [B,T,1,H,W] float maps: kind 0 = loss_ce_heatmap_* (cross_entropy.py:27-35),
    kind 1 = loss_mse_heatmap_final (mse.py), with base_loss_with_validity.py:64-73's reduction -- two launches
    forward, one backward, no host sync."""

    @staticmethod
    def forward(ctx, pred, gt, validity, kind):
        loss, w = default_kernels().heatmap_loss_fwd(kind, pred, gt, validity)
        ctx.kind = kind
        ctx.save_for_backward(pred, gt, w)
        return loss

    @staticmethod
    def backward(ctx, g):
        pred, gt, w = ctx.saved_tensors
        return default_kernels().heatmap_loss_bwd(ctx.kind, pred, gt, w, g), None, None, None
