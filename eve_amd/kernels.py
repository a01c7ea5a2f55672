"""Tensor-level launcher over the C ABI of libeve_hip.so.

`HipKernels` takes torch tensors (device memory owned by torch's allocator), checks layout/dtype,
and calls the `extern "C"` entry points of include/eve_hip.h with raw pointers, explicit shapes and
torch's current HIP stream.  PyTorch is plumbing here: allocation and stream only.

Activations are NHWC (`[N, H, W, C]` contiguous) in the compute dtype (float32, bfloat16 or float16).
"""
import ctypes

import torch

from . import _lib
from ._lib import ConvDesc

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SELU, ACT_TANH, ACT_SIGMOID = range(6)
EPI_ACCUMULATE = 0x100          # include/eve_hip.h EVE_EPI_ACCUMULATE: y += act(conv + bias)
DT_F32, DT_BF16, DT_F16 = 0, 1, 2
HALF_DTYPES = (torch.bfloat16, torch.float16)      # the two 16-bit instantiations of the MFMA kernels


def dt_code(dtype):
    if dtype == torch.float32:
        return DT_F32
    if dtype == torch.bfloat16:
        return DT_BF16
    if dtype == torch.float16:
        return DT_F16
    raise TypeError('eve_amd kernels support float32, bfloat16 and float16, got %s' % dtype)


def vec_of(dtype):
    return 4 if dtype == torch.float32 else 8


def pad_channels(c, dtype):
    v = vec_of(dtype)
    return (c + v - 1) // v * v


def conv_out_size(i, k, stride, pad):
    return (i + 2 * pad - k) // stride + 1


class HipKernels(object):
    """One instance per process; loads the library lazily and fails loudly if it is absent."""

    name = 'hip'

    def __init__(self):
        self.lib = _lib.load()
        self.prof = None          # list of (tag, flops, start_event, end_event) while profiling
        self._workspaces = {}     # (device, stream) -> uint8 scratch tensor, allocated once and never replaced

    WORKSPACE_BYTES = int(__import__('os').environ.get('EVE_AMD_WORKSPACE_MB', '128')) << 20

    def workspace(self, device):
        """(pointer, bytes) of this process's device scratch for the library calls that take one (include/eve_hip.h: the
        split-K partial filters of the weight-gradient kernels -- 7 splits x 512 x 4608 floats = 66 MB for ResNet layer 4 --
        and the re-packed filters of the stride-2 data gradient).  Passed PER CALL; allocated once per device by torch (the
        library never allocates) and kept for the life of the process, so a pointer captured into a hipGraph stays valid.
        Launches on one stream use it one after the other; the scratch is keyed by (device, current stream), so work issued
        from a second stream (a warm-up side stream, another trainer) gets its own and cannot corrupt split-K partials."""
        if device.type != 'cuda' or self.WORKSPACE_BYTES <= 0:
            return None, 0
        index = device.index if device.index is not None else torch.cuda.current_device()
        stream = torch.cuda.current_stream(index)
        if torch.cuda.is_current_stream_capturing():
            # a capture stream stands for the stream the graph will replay on; launches of one graph are ordered by the
            # captured dependencies, so they share one scratch whatever stream object carried the capture
            key = (index, 'graph')
        else:
            key = (index, stream.cuda_stream)
        ws = self._workspaces.get(key)
        if ws is None:
            ws = self._workspaces[key] = torch.empty(self.WORKSPACE_BYTES, dtype=torch.uint8, device=device)
        return ctypes.c_void_p(ws.data_ptr()), ws.numel()

    def prepare_graph_workspace(self, device):
        """Allocate the scratch of captured launches BEFORE a capture begins: first touched inside one it would come out of the
        graph's private memory pool and keep that pool's segment alive after the graph is destroyed."""
        if device.type == 'cuda' and self.WORKSPACE_BYTES > 0:
            index = device.index if device.index is not None else torch.cuda.current_device()
            if (index, 'graph') not in self._workspaces:
                self._workspaces[(index, 'graph')] = torch.empty(self.WORKSPACE_BYTES, dtype=torch.uint8, device=device)

    # ------------------------------------------------------------------ kernel selection (eve_dispatch_config)
    def dispatch_config(self):
        c = _lib.DispatchConfig()
        self._ck(self.lib.eve_get_dispatch_config(ctypes.byref(c)))
        return c

    def default_dispatch_config(self):
        c = _lib.DispatchConfig()
        self._ck(self.lib.eve_get_default_dispatch_config(ctypes.byref(c)))
        return c

    def dispatch_override(self, **fields):
        """Context manager: force kernel-selection fields for the calls inside it (tests comparing two kernels on the same
        inputs, tuning sweeps); the previous table is restored on exit."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old = self.dispatch_config()
            new = self.dispatch_config()
            for n, v in fields.items():
                assert hasattr(new, n), 'eve_dispatch_config has no field %r' % n
                setattr(new, n, int(v))
            self._ck(self.lib.eve_set_dispatch_config(ctypes.byref(new)))
            try:
                yield new
            finally:
                self._ck(self.lib.eve_set_dispatch_config(ctypes.byref(old)))
        return ctx()

    # ------------------------------------------------------------------ per-launch timing (bench roofline)
    def start_profile(self):
        """Record a HIP event pair around every conv launch on the stream it is launched on."""
        self.prof = []

    def stop_profile(self):
        """-> {tag: {'launches', 'ms', 'flops'}}; synchronises the device."""
        rec, self.prof = self.prof or [], None
        # an event pair with nothing between it still reads a few us..tens of us on this runtime: calibrate and
        # subtract, so the per-launch figure agrees with rocprofv3's kernel durations
        null = []
        for _ in range(32):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            null.append((e0, e1))
        torch.cuda.synchronize()
        overhead = sorted(a.elapsed_time(b) for a, b in null)[len(null) // 2]
        out, by_kernel = {}, {}
        for tag, flops, nbytes, e0, e1, sym in rec:
            ms = max(e0.elapsed_time(e1) - overhead, 1e-4)
            for table, key in ((out, tag), (by_kernel, sym)):
                d = table.setdefault(key, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0})
                d['launches'] += 1
                d['ms'] += ms
                d['flops'] += flops
                d['bytes'] += nbytes
        out['_event_overhead_ms'] = overhead
        out['_by_kernel'] = by_kernel          # keyed by the kernel symbol the library reports (eve_last_kernel)
        return out

    def _timed(self, tag, flops, fn, tensors=()):
        """tensors: what the launch reads and writes once (its ALGORITHMIC bytes, for HBM-bound kernels)."""
        if self.prof is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        sym = self.lib.eve_last_kernel()
        nbytes = float(sum(t.numel() * t.element_size() for t in tensors if t is not None))
        self.prof.append((tag, flops, nbytes, e0, e1, sym.decode() if sym else ''))
        return r

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _p(t):
        if t is None:
            return None
        if not t.is_cuda:
            raise RuntimeError('eve_amd: tensor is not on the GPU (the hot path has no CPU fallback)')
        if not t.is_contiguous():
            raise RuntimeError('eve_amd: non-contiguous tensor handed to a kernel')
        return ctypes.c_void_p(t.data_ptr())

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _ck(self, status):
        if status != 0:
            _lib.check(status, self.lib)

    @staticmethod
    def _f32(t, what):
        if t is not None and t.dtype != torch.float32:
            raise TypeError('%s must be float32' % what)
        return t

    @staticmethod
    def _desc(dtype, N, IH, IW, Cin, Cout, KH, KW, stride, pad):
        d = ConvDesc()
        d.dtype = dt_code(dtype)
        d.N, d.IH, d.IW, d.Cin = N, IH, IW, Cin
        d.OH, d.OW, d.Cout = conv_out_size(IH, KH, stride, pad), conv_out_size(IW, KW, stride, pad), Cout
        d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
        return d

    # ------------------------------------------------------------------ convolution
    def conv2d_fwd(self, x, w_ohwi, bias, stride, pad, epi_act=ACT_NONE, ss=None, pro_act=ACT_NONE, algo=None,
                   accumulate_into=None):
        """algo = (true Cout, true KH*KW*Cin) when channels are zero-padded (algorithmic FLOP count).
        accumulate_into: a contiguous [N, OH, OW, Cout] tensor the result is ADDED to in the kernel epilogue
        (EVE_EPI_ACCUMULATE; the block's `layers(x) + skip_layer(x)` without an add launch); returned."""
        N, IH, IW, Cin = x.shape
        Cout, KH, KW, Cin2 = w_ohwi.shape
        assert Cin2 == Cin and w_ohwi.dtype == x.dtype
        d = self._desc(x.dtype, N, IH, IW, Cin, Cout, KH, KW, stride, pad)
        if accumulate_into is not None:
            y = accumulate_into
            assert tuple(y.shape) == (N, d.OH, d.OW, Cout) and y.dtype == x.dtype and y.is_contiguous() and ss is None
            epi_act = epi_act | EPI_ACCUMULATE
        else:
            y = torch.empty((N, d.OH, d.OW, Cout), dtype=x.dtype, device=x.device)
        co, kk = algo or (Cout, KH * KW * Cin)
        self._timed('conv_fwd', 2.0 * N * d.OH * d.OW * co * kk, lambda: self._ck(self.lib.eve_conv2d_fwd(
            ctypes.byref(d), self._p(x), self._p(w_ohwi), self._p(self._f32(bias, 'bias')), epi_act,
            self._p(self._f32(ss, 'scale/shift')), pro_act, self._p(y), self._stream())),
            (x, w_ohwi, y) + ((y,) if accumulate_into is not None else ()))
        return y

    def conv2d_fwd_stats(self, x, w_ohwi, bias, stride, pad, epi_act=ACT_NONE, eps=1e-5, algo=None):
        """conv2d_fwd that also returns the InstanceNorm statistics (mean, rstd) [N, Cout, 2] of its output when the dispatched
        kernel can form them in its epilogue (the row-streaming 3x3 kernel walks whole images), else None: -> (y, mean_rstd | None)."""
        N, IH, IW, Cin = x.shape
        Cout, KH, KW, Cin2 = w_ohwi.shape
        assert Cin2 == Cin and w_ohwi.dtype == x.dtype
        d = self._desc(x.dtype, N, IH, IW, Cin, Cout, KH, KW, stride, pad)
        y = torch.empty((N, d.OH, d.OW, Cout), dtype=x.dtype, device=x.device)
        mr = torch.empty((N, Cout, 2), dtype=torch.float32, device=x.device)
        written = ctypes.c_int(0)
        co, kk = algo or (Cout, KH * KW * Cin)
        self._timed('conv_fwd', 2.0 * N * d.OH * d.OW * co * kk, lambda: self._ck(self.lib.eve_conv2d_fwd_stats(
            ctypes.byref(d), self._p(x), self._p(w_ohwi), self._p(self._f32(bias, 'bias')), epi_act, self._p(y), self._p(mr),
            float(eps), ctypes.byref(written), self._stream())), (x, w_ohwi, y))
        return y, (mr if written.value else None)

    def conv2d_dgrad(self, dy, w_ihwo, in_hw, stride, pad, algo=None, accumulate_into=None):
        """Data gradient; with accumulate_into (a contiguous [N, IH, IW, Cin] tensor) it is ADDED to that tensor in
        the kernel epilogue and the same tensor is returned."""
        N, OH, OW, Cout = dy.shape
        Cin, KH, KW, Cout2 = w_ihwo.shape
        assert Cout2 == Cout and w_ihwo.dtype == dy.dtype
        IH, IW = in_hw
        d = self._desc(dy.dtype, N, IH, IW, Cin, Cout, KH, KW, stride, pad)
        assert (d.OH, d.OW) == (OH, OW)
        co, kk = algo or (Cout, KH * KW * Cin)
        # (the stride-2 data gradient re-packs its filters into the scratch; dgrad_acc takes none)
        wsp, wsn = self.workspace(dy.device) if (stride == 2 and dy.dtype in HALF_DTYPES) else (None, 0)
        if accumulate_into is not None:
            dx = accumulate_into
            assert tuple(dx.shape) == (N, IH, IW, Cin) and dx.dtype == dy.dtype and dx.is_contiguous()
            fn = lambda: self.lib.eve_conv2d_dgrad_acc(ctypes.byref(d), self._p(dy), self._p(w_ihwo), self._p(dx), self._stream())
        else:
            dx = torch.empty((N, IH, IW, Cin), dtype=dy.dtype, device=dy.device)
            fn = lambda: self.lib.eve_conv2d_dgrad(ctypes.byref(d), self._p(dy), self._p(w_ihwo), self._p(dx), wsp, wsn, self._stream())
        self._timed('conv_dgrad', 2.0 * N * OH * OW * co * kk, lambda: self._ck(fn()), (dy, w_ihwo, dx))
        return dx

    def conv2d_wgrad(self, x, dy, KH, KW, stride, pad, dw_ohwi, ss=None, pro_act=ACT_NONE, algo=None, db=None):
        """Accumulates into dw_ohwi (float32 [Cout, KH, KW, Cin]) and, when given, the column sums of dy into db
        (float32 [Cout]) in the same pass."""
        N, IH, IW, Cin = x.shape
        Cout = dy.shape[3]
        assert dw_ohwi.shape == (Cout, KH, KW, Cin) and dw_ohwi.dtype == torch.float32
        d = self._desc(x.dtype, N, IH, IW, Cin, Cout, KH, KW, stride, pad)
        assert tuple(dy.shape) == (N, d.OH, d.OW, Cout) and dy.dtype == x.dtype
        co, kk = algo or (Cout, KH * KW * Cin)
        wsp, wsn = self.workspace(x.device) if x.dtype in HALF_DTYPES else (None, 0)
        if db is not None:
            assert ss is None and db.shape == (Cout,) and db.dtype == torch.float32 and db.is_contiguous()
            self._timed('conv_wgrad', 2.0 * N * d.OH * d.OW * co * kk, lambda: self._ck(self.lib.eve_conv2d_wgrad_bias(
                ctypes.byref(d), self._p(x), self._p(dy), self._p(dw_ohwi), self._p(db), wsp, wsn, self._stream())), (x, dy, dw_ohwi))
            return dw_ohwi
        self._timed('conv_wgrad', 2.0 * N * d.OH * d.OW * co * kk, lambda: self._ck(self.lib.eve_conv2d_wgrad(
            ctypes.byref(d), self._p(x), self._p(dy), self._p(self._f32(ss, 'scale/shift')), pro_act,
            self._p(dw_ohwi), wsp, wsn, self._stream())), (x, dy, dw_ohwi))
        return dw_ohwi

    def stem_pack_input(self, src_nchw, out=None, dtype=torch.bfloat16):
        N, C, H, W = src_nchw.shape
        src = self._f32(src_nchw.contiguous(), 'src')
        dst = out if out is not None else torch.empty((N, H + 6, W + 8, 4), dtype=dtype, device=src.device)
        assert tuple(dst.shape) == (N, H + 6, W + 8, 4) and dst.dtype in HALF_DTYPES
        self._ck(self.lib.eve_stem_pack_input(dt_code(dst.dtype), N, C, H, W, self._p(src), self._p(dst), self._stream()))
        return dst

    def stem7x7s2_fwd(self, x_padded, w_ohwi8):
        N, Hp, Wp, four = x_padded.shape
        IH, IW = Hp - 6, Wp - 8
        assert four == 4 and tuple(w_ohwi8.shape) == (64, 7, 7, 8) and w_ohwi8.dtype == x_padded.dtype and x_padded.dtype in HALF_DTYPES
        y = torch.empty((N, IH // 2, IW // 2, 64), dtype=x_padded.dtype, device=x_padded.device)
        flops = 2.0 * N * (IH // 2) * (IW // 2) * 64 * 147
        self._timed('conv_fwd', flops, lambda: self._ck(self.lib.eve_stem7x7s2_fwd(
            dt_code(x_padded.dtype), N, IH, IW, self._p(x_padded), self._p(w_ohwi8), self._p(y), self._stream())))
        return y

    def stem_fwd_fused(self, x_padded, w_ohwi8, eps=1e-5):
        N, Hp, Wp, four = x_padded.shape
        IH, IW = Hp - 6, Wp - 8
        assert four == 4 and tuple(w_ohwi8.shape) == (64, 7, 7, 8) and w_ohwi8.dtype == x_padded.dtype and x_padded.dtype in HALF_DTYPES
        y = torch.empty((N, IH // 4, IW // 4, 64), dtype=x_padded.dtype, device=x_padded.device)
        idx = torch.empty((N, IH // 4, IW // 4, 64), dtype=torch.uint8, device=x_padded.device)
        mr = torch.empty((N, 64, 2), dtype=torch.float32, device=x_padded.device)
        flops = 2.0 * N * (IH // 2) * (IW // 2) * 64 * 147
        self._timed('conv_fwd', flops, lambda: self._ck(self.lib.eve_stem_fwd_fused(
            dt_code(x_padded.dtype), N, IH, IW, self._p(x_padded), self._p(w_ohwi8), eps, self._p(y), self._p(idx), self._p(mr),
            self._stream())))
        return y, idx, mr

    def stem_wgrad(self, x_padded, dconv, dw):
        """dw [64, 7, 8, 4] float32 += weight gradient of conv1 from the packed patches (filter column 7 / channel 3 unused)."""
        N, Hp, Wp, _ = x_padded.shape
        IH, IW = Hp - 6, Wp - 8
        assert tuple(dw.shape) == (64, 7, 8, 4) and dw.dtype == torch.float32 and dw.is_contiguous()
        assert tuple(dconv.shape) == (N, IH // 2, IW // 2, 64) and dconv.dtype == x_padded.dtype
        flops = 2.0 * N * (IH // 2) * (IW // 2) * 64 * 147
        self._timed('conv_wgrad', flops, lambda: self._ck(self.lib.eve_stem_wgrad(
            dt_code(x_padded.dtype), N, IH, IW, self._p(x_padded), self._p(dconv), self._p(dw), self._stream())))

    def stem_bwd_dx(self, x_padded, w_ohwi8, mr, dy_pool, y_pool, idx, dy_pool2=None):
        N, Hp, Wp, _ = x_padded.shape
        IH, IW = Hp - 6, Wp - 8
        dx = torch.empty((N, IH // 2, IW // 2, 64), dtype=x_padded.dtype, device=x_padded.device)
        assert dy_pool.dtype == x_padded.dtype and dy_pool.is_contiguous() and dy_pool.shape == y_pool.shape == idx.shape
        # (the convolution is RE-computed here: no algorithmic FLOPs are credited)
        self._timed('stem_bwd', 0.0, lambda: self._ck(self.lib.eve_stem_bwd_dx(
            dt_code(x_padded.dtype), N, IH, IW, self._p(x_padded), self._p(w_ohwi8), self._p(self._f32(mr, 'mean_rstd')), self._p(dy_pool),
            self._p(dy_pool2), self._p(y_pool), self._p(idx), self._p(dx), self._stream())))
        return dx

    def stem_bwd_wgrad(self, x_padded, w_ohwi8, mr, dy_pool, y_pool, idx, dw, dy_pool2=None):
        """dw [64, 7, 8, 4] float32 += the stem's weight gradient straight from d(pooled output): backward of the fused stem and
        its weight gradient in one launch, d(conv1 out) never written (include/eve_hip.h eve_stem_bwd_wgrad)."""
        N, Hp, Wp, _ = x_padded.shape
        IH, IW = Hp - 6, Wp - 8
        assert tuple(dw.shape) == (64, 7, 8, 4) and dw.dtype == torch.float32 and dw.is_contiguous()
        assert dy_pool.dtype == x_padded.dtype and dy_pool.is_contiguous() and dy_pool.shape == y_pool.shape == idx.shape
        # the weight gradient's FLOPs are algorithmic; the recomputed convolution is not credited
        flops = 2.0 * N * (IH // 2) * (IW // 2) * 64 * 147
        # the call's scratch (the masked, summed gradient + the per-plane constants: 252 MB at N = 1 920), from torch's allocator
        # like any activation: stream-ordered, and part of the graph's pool under capture
        nbytes = int(self.lib.eve_stem_bwd_wgrad_workspace(dt_code(x_padded.dtype), N, IH))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=x_padded.device)
        self._timed('conv_wgrad', flops, lambda: self._ck(self.lib.eve_stem_bwd_wgrad(
            dt_code(x_padded.dtype), N, IH, IW, self._p(x_padded), self._p(w_ohwi8), self._p(self._f32(mr, 'mean_rstd')), self._p(dy_pool),
            self._p(dy_pool2), self._p(y_pool), self._p(idx), self._p(dw), self._p(ws), nbytes, self._stream())),
            (x_padded, dy_pool, dy_pool2, y_pool, idx))

    # ------------------------------------------------------------------ chains of small float32 linear layers (the tail)
    def linear_wgrad_batch(self, problems):
        """All weight / bias gradients of the tail in one launch.  problems: dicts dY [M,N], Y (or None) + act, X [M,K1], X2 (or
        None), dW [N,K] (accumulated), db [N] or None."""
        n = len(problems)
        assert 0 < n <= _lib.WGRAD_BATCH_MAX
        arr = (_lib.WgradProblem * n)()
        for q, pr in zip(arr, problems):
            dY, X, dW = pr['dY'], pr['X'], pr['dW']
            for t in (dY, X, dW, pr.get('Y'), pr.get('X2'), pr.get('db')):
                assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda)
            q.dY, q.X, q.dW = dY.data_ptr(), X.data_ptr(), dW.data_ptr()
            q.Y = pr['Y'].data_ptr() if pr.get('Y') is not None else None
            q.X2 = pr['X2'].data_ptr() if pr.get('X2') is not None else None
            q.db = pr['db'].data_ptr() if pr.get('db') is not None else None
            q.M, q.N, q.K = dY.shape[0], dW.shape[0], dW.shape[1]
            q.K1, q.K2 = pr.get('K1', X.shape[1]), (pr['X2'].shape[1] if pr.get('X2') is not None else 0)
            q.act = pr.get('act', ACT_NONE)
            q.ldY = dY.shape[1] if dY.shape[1] != q.N else 0
            q.ldX = X.shape[1] if X.shape[1] != q.K1 else 0           # the leading K1 columns of a wider X
            q.ldW, q.x_shift_T, q.reserved = 0, int(pr.get('x_shift_T', 0)), 0
            assert dY.shape[1] >= q.N and q.K1 + q.K2 <= q.K and X.shape[0] == q.M and X.shape[1] >= q.K1
            assert not q.x_shift_T or (q.K2 == 0 and q.K1 == q.K and q.M % q.x_shift_T == 0)    # (the kernel reads column k of X for every k < K)
            assert pr.get('Y') is None or pr['Y'].shape == dY.shape
        self._timed('tail', 0.0, lambda: self._ck(self.lib.eve_linear_wgrad_batch(arr, n, self._stream())))

    def stem_fused_wgrad_enabled(self):
        return bool(self.dispatch_config().stem_fused_wgrad)

    # ------------------------------------------------------------------ decoded uint8 frames (input pipeline)
    def frames_u8_to_nchw(self, frames, scale, shift=None):
        """uint8 [N,H,W,C] -> float32 [N,C,H,W] = frames * scale (+ shift)."""
        N, H, W, C = frames.shape
        assert frames.dtype == torch.uint8
        out = torch.empty((N, C, H, W), dtype=torch.float32, device=frames.device)
        self._ck(self.lib.eve_frames_u8_to_nchw(N, H, W, C, self._p(frames), float(scale), float(shift or 0.0),
                                                0 if shift is None else 1, self._p(out), self._stream()))
        return out

    def frames_u8_to_stem(self, frames, scale, shift, out=None, dtype=torch.bfloat16):
        """uint8 [N,H,W,C<=4] -> the stem's packed 16-bit input [N,H+6,W+8,4]."""
        N, H, W, C = frames.shape
        assert frames.dtype == torch.uint8
        dst = out if out is not None else torch.empty((N, H + 6, W + 8, 4), dtype=dtype, device=frames.device)
        assert tuple(dst.shape) == (N, H + 6, W + 8, 4) and dst.dtype in HALF_DTYPES
        self._ck(self.lib.eve_frames_u8_to_stem(dt_code(dst.dtype), N, C, H, W, self._p(frames), float(scale), float(shift), self._p(dst), self._stream()))
        return dst

    # ------------------------------------------------------------------ gaze geometry / heat-maps / soft-argmax
    def _flat32(self, t, shape, what):
        t = t.contiguous()
        assert t.dtype == torch.float32 and tuple(t.shape) == tuple(shape), (what, t.dtype, tuple(t.shape), tuple(shape))
        self._p(t)                                   # device check
        return t

    def gaze_to_pog(self, g, origin, R, inv_cam, ppm, screen, head_R=None, kappa=None):
        """Flat N frames.  Returns (g_out [N,2], pog_mm [N,2], pog_px [N,2], jac [N,6,2])."""
        N = g.shape[0]
        g = self._flat32(g, (N, 2), 'g'); origin = self._flat32(origin, (N, 3), 'origin')
        R = self._flat32(R, (N, 3, 3), 'R'); inv_cam = self._flat32(inv_cam, (N, 4, 4), 'inv_cam')
        ppm = self._flat32(ppm, (N, 2), 'ppm')
        if kappa is not None:
            head_R = self._flat32(head_R, (N, 3, 3), 'head_R'); kappa = self._flat32(kappa, (N, 2), 'kappa')
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=g.device)
        g_out, mm, px, jac = new(N, 2), new(N, 2), new(N, 2), new(N, 6, 2)
        self._ck(self.lib.eve_gaze_to_pog(N, self._p(g), self._p(origin), self._p(R), self._p(inv_cam), self._p(ppm),
                                          self._p(head_R if kappa is not None else None), self._p(kappa), float(screen[0]),
                                          float(screen[1]), self._p(g_out), self._p(mm), self._p(px), self._p(jac), self._stream()))
        return g_out, mm, px, jac

    def gaze_to_pog_bwd(self, jac, dg_out, dmm, dpx):
        N = jac.shape[0]
        f = lambda t: None if t is None else self._flat32(t, (N, 2), 'grad')
        dg_out, dmm, dpx = f(dg_out), f(dmm), f(dpx)
        dg = torch.empty((N, 2), dtype=torch.float32, device=jac.device)
        self._ck(self.lib.eve_gaze_to_pog_bwd(N, self._p(jac), self._p(dg_out), self._p(dmm), self._p(dpx), self._p(dg), self._stream()))
        return dg

    def combined_gaze(self, origin, pog_mm, R, cam):
        N = origin.shape[0]
        origin = self._flat32(origin, (N, 3), 'origin'); pog_mm = self._flat32(pog_mm, (N, 2), 'pog_mm')
        R = self._flat32(R, (N, 3, 3), 'R'); cam = self._flat32(cam, (N, 4, 4), 'cam')
        g = torch.empty((N, 2), dtype=torch.float32, device=origin.device)
        self._ck(self.lib.eve_combined_gaze(N, self._p(origin), self._p(pog_mm), self._p(R), self._p(cam), self._p(g), self._stream()))
        return g

    def make_heatmaps(self, centres_px, sigma, hw, screen, validity=None):
        """centres [N,2] px -> [N,1,H,W] float32 (x validity[n] when given)."""
        N = centres_px.shape[0]
        c = self._flat32(centres_px, (N, 2), 'centres')
        v = None
        if validity is not None:
            v = validity.contiguous()
            v = v.view(torch.uint8) if v.dtype == torch.bool else v.to(torch.uint8)
            assert tuple(v.shape) == (N,)
        out = torch.empty((N, 1, hw[0], hw[1]), dtype=torch.float32, device=c.device)
        for i in range(0, N, 65535):
            n = min(65535, N - i)
            self._ck(self.lib.eve_make_heatmaps(n, hw[0], hw[1], self._p(c[i:i + n]), self._p(None if v is None else v[i:i + n]),
                                                float(sigma), float(screen[0]), float(screen[1]), self._p(out[i:i + n]), self._stream()))
        return out

    def make_heatmaps_bwd(self, centres_px, sigma, screen, dout):
        N, _, H, W = dout.shape
        c = self._flat32(centres_px, (N, 2), 'centres')
        dout = self._flat32(dout, (N, 1, H, W), 'dout')
        dc = torch.empty((N, 2), dtype=torch.float32, device=c.device)
        self._ck(self.lib.eve_make_heatmaps_bwd(N, H, W, self._p(c), float(sigma), float(screen[0]), float(screen[1]), self._p(dout),
                                                self._p(dc), self._stream()))
        return dc

    def soft_argmax_fwd(self, heat, screen):
        """heat [N,1,H,W] float32 -> (pog_px [N,2], stats [N,4])."""
        N, _, H, W = heat.shape
        heat = self._flat32(heat, (N, 1, H, W), 'heat')
        px = torch.empty((N, 2), dtype=torch.float32, device=heat.device)
        stats = torch.empty((N, 4), dtype=torch.float32, device=heat.device)
        self._ck(self.lib.eve_soft_argmax_fwd(N, H, W, self._p(heat), float(screen[0]), float(screen[1]), self._p(px), self._p(stats),
                                              self._stream()))
        return px, stats

    def soft_argmax_bwd(self, heat, stats, dpog, screen):
        N, _, H, W = heat.shape
        dpog = self._flat32(dpog, (N, 2), 'dpog')
        dh = torch.empty_like(heat)
        for i in range(0, N, 65535):
            n = min(65535, N - i)
            self._ck(self.lib.eve_soft_argmax_bwd(n, H, W, self._p(heat[i:i + n]), self._p(stats[i:i + n]), self._p(dpog[i:i + n]),
                                                  float(screen[0]), float(screen[1]), self._p(dh[i:i + n]), self._stream()))
        return dh

    # ------------------------------------------------------------------ RefineNet head + heat-map losses
    def heatmap_head_fwd(self, logits):
        """logits NHWC [N, H, W, Cpad] (channel 0 is the map) -> float [N, 1, H, W] = sigmoid, evaluated in float."""
        N, H, W, Cp = logits.shape
        out = torch.empty((N, 1, H, W), dtype=torch.float32, device=logits.device)
        self._ck(self.lib.eve_heatmap_head_fwd(dt_code(logits.dtype), N * H * W, Cp, self._p(logits), self._p(out),
                                               self._stream()))
        return out

    def heatmap_head_bwd(self, dy, y, dtype, cpad):
        N, _, H, W = y.shape
        dy = dy.contiguous().float()
        dl = torch.empty((N, H, W, cpad), dtype=dtype, device=y.device)
        self._ck(self.lib.eve_heatmap_head_bwd(dt_code(dtype), N * H * W, cpad, self._p(dy), self._p(y), self._p(dl),
                                               self._stream()))
        return dl

    def heatmap_loss_fwd(self, kind, pred, gt, validity):
        """kind 0 = BCE, 1 = MSE; pred, gt float [B, T, ...map...]; validity bool/uint8 [B, T].
        Returns (loss 0-dim, w [B*T] = d loss / d per-frame mean)."""
        B, T = pred.shape[:2]
        HW = pred[0, 0].numel()
        pred, gt = pred.contiguous(), gt.contiguous()
        assert pred.dtype == torch.float32 and gt.dtype == torch.float32 and gt.shape == pred.shape
        v = validity.contiguous()
        v = v.view(torch.uint8) if v.dtype == torch.bool else v.to(torch.uint8)
        assert tuple(v.shape) == (B, T)
        per_map = torch.empty((B * T,), dtype=torch.float32, device=pred.device)
        w = torch.empty((B * T,), dtype=torch.float32, device=pred.device)
        loss = torch.empty((1,), dtype=torch.float32, device=pred.device)
        self._ck(self.lib.eve_heatmap_loss_fwd(kind, B, T, HW, self._p(pred), self._p(gt), self._p(v), self._p(per_map),
                                               self._p(loss), self._p(w), self._stream()))
        return loss.view(()), w

    def heatmap_loss_bwd(self, kind, pred, gt, w, upstream):
        B, T = pred.shape[:2]
        HW = pred[0, 0].numel()
        pred, gt = pred.contiguous(), gt.contiguous()
        up = upstream.detach().float().reshape(1).contiguous()
        dp = torch.empty_like(pred)
        self._ck(self.lib.eve_heatmap_loss_bwd(kind, B * T, HW, self._p(pred), self._p(gt), self._p(w), self._p(up),
                                               self._p(dp), self._stream()))
        return dp

    # ------------------------------------------------------------------ fused train-step losses
    def eye_losses(self, g_pred, g_tgt, g_val, p_pred, p_tgt, p_val, coeff_ang, coeff_l1):
        """Each argument is a (left, right) pair.  Returns terms[5], (dg_l, dg_r), (dp_l, dp_r)."""
        B, T = p_pred[0].shape
        dev = p_pred[0].device

        def pair(ts, dtype, shape):
            out = []
            for t in ts:
                t = t.contiguous()
                if dtype == torch.uint8:
                    t = t.view(torch.uint8) if t.dtype == torch.bool else t.to(torch.uint8)
                assert t.dtype == dtype and tuple(t.shape) == shape and t.is_cuda, (t.dtype, t.shape)
                out.append(t)
            return out, (ctypes.c_void_p * 2)(*[t.data_ptr() for t in out])

        keep = []
        ptrs = []
        for ts, dt, shp in ((g_pred, torch.float32, (B, T, 2)), (g_tgt, torch.float32, (B, T, 2)), (g_val, torch.uint8, (B, T)),
                            (p_pred, torch.float32, (B, T)), (p_tgt, torch.float32, (B, T)), (p_val, torch.uint8, (B, T))):
            k_, a = pair(ts, dt, shp)
            keep.append(k_)
            ptrs.append(a)
        terms = torch.zeros((5,), dtype=torch.float32, device=dev)
        dg = [torch.empty((B, T, 2), dtype=torch.float32, device=dev) for _ in range(2)]
        dp = [torch.empty((B, T), dtype=torch.float32, device=dev) for _ in range(2)]
        dg_a = (ctypes.c_void_p * 2)(*[t.data_ptr() for t in dg])
        dp_a = (ctypes.c_void_p * 2)(*[t.data_ptr() for t in dp])
        self._ck(self.lib.eve_eye_losses(B, T, ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], ptrs[5], float(coeff_ang),
                                         float(coeff_l1), self._p(terms), dg_a, dp_a, self._stream()))
        return terms, dg, dp

    # ------------------------------------------------------------------ small float32 linear layers
    def linear_fwd(self, x, w_in_out, bias, act):
        M, K = x.shape
        N = w_in_out.shape[1]
        assert x.dtype == torch.float32 and w_in_out.dtype == torch.float32 and w_in_out.shape[0] == K
        assert x.is_contiguous() and w_in_out.is_contiguous()
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        self._ck(self.lib.eve_linear_fwd(M, K, N, self._p(x), self._p(w_in_out), self._p(self._f32(bias, 'bias')), act,
                                         self._p(y), self._stream()))
        return y

    def linear_fwd_ex(self, x, k_cols, w_in_out, bias, act, y, n_cols=None):
        """y[:, :N] = act(x[:, :K] @ w_in_out + bias): K leading columns of x, N leading columns of y (both may be wider), a bias
        with fewer than N entries covers the leading columns (the zero-padded heads)."""
        M = x.shape[0]
        K, N = w_in_out.shape
        assert k_cols == K and (n_cols is None or n_cols == N)
        for t in (x, w_in_out, y, bias):
            assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda)
        assert x.shape[1] >= K and y.shape[1] >= N and y.shape[0] == M
        self._ck(self.lib.eve_linear_fwd_ex(M, K, N, self._p(x), x.shape[1], self._p(w_in_out), self._p(bias),
                                            bias.numel() if bias is not None else 0, act, self._p(y), y.shape[1], self._stream()))
        return y

    def linear_dgrad_ex(self, dy, n_cols, y, act, w_out_in, dx, accumulate=False):
        """dx (+)= (dy[:, :N] * act'(y[:, :N])) @ w_out_in  -- dy / y may be wider than N, dx exactly K wide."""
        M = dy.shape[0]
        N, K = w_out_in.shape
        assert n_cols == N and dy.shape[1] >= N and tuple(dx.shape) == (M, K) and (y is None or y.shape == dy.shape)
        for t in (dy, y, w_out_in, dx):
            assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda)
        self._ck(self.lib.eve_linear_dgrad_ex(M, K, N, self._p(dy), dy.shape[1], self._p(y), act, self._p(w_out_in), self._p(dx), K,
                                              int(bool(accumulate)), self._stream()))
        return dx

    def tail_head_pose(self, h_left, h_right, cat, col):
        BT = h_left.numel() // 2
        for t in (h_left, h_right, cat):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
        assert h_right.numel() == 2 * BT and cat.shape[0] == 2 * BT and cat.shape[1] >= col + 4
        self._ck(self.lib.eve_tail_head_pose(BT, self._p(h_left), self._p(h_right), self._p(cat), cat.shape[1], col, self._stream()))

    def tail_outputs_fwd(self, g2, p2):
        M = g2.shape[0]
        assert tuple(g2.shape) == (M, 4) and tuple(p2.shape) == (M, 4) and g2.is_contiguous() and p2.is_contiguous()
        gaze = torch.empty((M, 2), dtype=torch.float32, device=g2.device)
        pupil = torch.empty((M,), dtype=torch.float32, device=g2.device)
        self._ck(self.lib.eve_tail_outputs_fwd(M, self._p(g2), self._p(p2), self._p(gaze), self._p(pupil), self._stream()))
        return gaze, pupil

    def tail_outputs_bwd(self, dg, dp, g_full, coeff_ang, coeff_l1):
        """dg / dp: the (left, right) unit gradients of eye_losses; g_full: device scalar or None -> d_g2, d_p2 [2*B*T, 4]."""
        BT = dp[0].numel()
        for t in (dg[0], dg[1], dp[0], dp[1]):
            assert t.dtype == torch.float32 and t.is_contiguous()
        assert g_full is None or (g_full.dtype == torch.float32 and g_full.numel() == 1 and g_full.is_cuda)
        d_g2 = torch.empty((2 * BT, 4), dtype=torch.float32, device=dp[0].device)
        d_p2 = torch.empty((2 * BT, 4), dtype=torch.float32, device=dp[0].device)
        self._ck(self.lib.eve_tail_outputs_bwd(BT, self._p(dg[0]), self._p(dg[1]), self._p(dp[0]), self._p(dp[1]), self._p(g_full),
                                               float(coeff_ang), float(coeff_l1), self._p(d_g2), self._p(d_p2), self._stream()))
        return d_g2, d_p2

    def linear_dgrad(self, dy, y, act, w_out_in):
        M, N = dy.shape
        K = w_out_in.shape[1]
        assert dy.dtype == torch.float32 and dy.is_contiguous() and tuple(w_out_in.shape) == (N, K)
        dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
        self._ck(self.lib.eve_linear_dgrad(M, K, N, self._p(dy), self._p(y), act, self._p(w_out_in), self._p(dx),
                                           self._stream()))
        return dx

    def linear_wgrad(self, dy, y, act, x, dw_out_in, db):
        """Accumulates into dw_out_in [N, K] (and db [N] when given)."""
        M, N = dy.shape
        K = x.shape[1]
        assert tuple(dw_out_in.shape) == (N, K) and dw_out_in.dtype == torch.float32 and dw_out_in.is_contiguous()
        assert x.is_contiguous() and dy.is_contiguous() and (db is None or (db.numel() == N and db.is_contiguous()))
        self._ck(self.lib.eve_linear_wgrad(M, K, N, self._p(dy), self._p(y), act, self._p(x), self._p(dw_out_in),
                                           self._p(db), self._stream()))

    def bias_grad(self, dy, db):
        C = dy.shape[-1]
        M = dy.numel() // C
        assert db.dtype == torch.float32 and db.numel() == C
        self._ck(self.lib.eve_bias_grad(dt_code(dy.dtype), M, C, self._p(dy), self._p(db), self._stream()))
        return db

    # ------------------------------------------------------------------ instance norm
    def instnorm_stats(self, x, eps=1e-5):
        N, H, W, C = x.shape
        mr = torch.empty((N, C, 2), dtype=torch.float32, device=x.device)
        self._timed('in_stats', 0.0, lambda: self._ck(self.lib.eve_instnorm_stats(
            dt_code(x.dtype), N, H * W, C, self._p(x), eps, self._p(mr), self._stream())), (x,))
        return mr

    def instnorm_act_fwd(self, x, mr, gamma, beta, res, act):
        N, H, W, C = x.shape
        y = torch.empty_like(x)
        self._timed('in_fwd', 0.0, lambda: self._ck(self.lib.eve_instnorm_act_fwd(
            dt_code(x.dtype), N, H * W, C, self._p(x), self._p(mr), self._p(self._f32(gamma, 'gamma')),
            self._p(self._f32(beta, 'beta')), self._p(res), act, self._p(y), self._stream())), (x, res, y))
        return y

    def instnorm_act_bwd(self, dy, y, x, mr, gamma, act, want_dres, beta=None, dx_add=None):
        """y may be None when the forward had no residual (beta is then needed next to gamma)."""
        N, H, W, C = x.shape
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if want_dres else None
        sums = torch.empty((N, C, 2), dtype=torch.float32, device=x.device)
        # two passes over (dy, x[, y]) -- the reductions, then the gradient -- are this kernel's algorithmic traffic
        self._timed('in_bwd', 0.0, lambda: self._ck(self.lib.eve_instnorm_act_bwd(
            dt_code(x.dtype), N, H * W, C, self._p(dy), self._p(y), self._p(x), self._p(mr),
            self._p(self._f32(gamma, 'gamma')), self._p(self._f32(beta, 'beta')), act, self._p(dx), self._p(dres),
            self._p(sums), self._p(dx_add), self._stream())), (dy, x, y, dy, x, y, dx, dres, dx_add))
        return dx, dres, sums

    def instnorm_act2_fwd(self, xs, mrs, gamma_a, beta_a, gamma_b, beta_b, act):
        """Two heads act(gamma_h * xhat + beta_h) of the channel-concatenation of the 1-2 NHWC sources `xs` (statistics
        `mrs` per source); returns (y_a, y_b), [N, H, W, sum C].  gamma_b / beta_b None: one head, y_b None."""
        x, x2 = xs[0], (xs[1] if len(xs) > 1 else None)
        N, H, W, C = x.shape
        C2 = x2.shape[-1] if x2 is not None else 0
        y_a = torch.empty((N, H, W, C + C2), dtype=x.dtype, device=x.device)
        y_b = torch.empty_like(y_a) if gamma_b is not None else None
        self._timed('in_fwd', 0.0, lambda: self._ck(self.lib.eve_instnorm_act2_fwd(
            dt_code(x.dtype), N, H * W, C, self._p(x), self._p(mrs[0]), self._p(self._f32(gamma_a, 'gamma')),
            self._p(self._f32(beta_a, 'beta')), self._p(self._f32(gamma_b, 'gamma')), self._p(self._f32(beta_b, 'beta')), act,
            self._p(y_a), self._p(y_b), C + C2, C2, self._p(x2), self._p(mrs[1] if x2 is not None else None), self._stream())),
            (x, x2, y_a, y_b))
        return y_a, y_b

    def instnorm_act2_bwd(self, dy_a, dy_b, xs, mrs, gamma_a, beta_a, gamma_b, beta_b, act):
        """Backward of instnorm_act2_fwd: dy_a / dy_b [N, H, W, sum C] (dy_b None with one head).
        Returns ([dx per source], sums_a, sums_b) with sums [N, sum C, 2]."""
        x, x2 = xs[0], (xs[1] if len(xs) > 1 else None)
        N, H, W, C = x.shape
        C2 = x2.shape[-1] if x2 is not None else 0
        assert dy_a.shape[-1] == C + C2
        dx = torch.empty_like(x)
        dx2 = torch.empty_like(x2) if x2 is not None else None
        sums_a = torch.empty((N, C + C2, 2), dtype=torch.float32, device=x.device)
        sums_b = torch.empty_like(sums_a) if dy_b is not None else None
        # two passes over (dy_a, dy_b, x), then the gradient: this kernel's algorithmic traffic
        self._timed('in_bwd', 0.0, lambda: self._ck(self.lib.eve_instnorm_act2_bwd(
            dt_code(x.dtype), N, H * W, C, self._p(dy_a), self._p(dy_b), C + C2, self._p(x), self._p(mrs[0]),
            self._p(self._f32(gamma_a, 'gamma')), self._p(self._f32(beta_a, 'beta')), self._p(self._f32(gamma_b, 'gamma')),
            self._p(self._f32(beta_b, 'beta')), act, self._p(dx), self._p(sums_a), self._p(sums_b), C2, self._p(x2),
            self._p(mrs[1] if x2 is not None else None), self._p(dx2), self._stream())),
            (dy_a, dy_b, x, x2, dy_a, dy_b, x, x2, dx, dx2))
        return ([dx] if x2 is None else [dx, dx2]), sums_a, sums_b

    def instnorm_fwd_fused(self, x, gamma, beta, res, act, eps=1e-5, want_mask=False):
        """Single-launch stats + apply; returns (y, mean_rstd) or None when the plane is too large.
        want_mask: also the sign mask of y (one byte per 16-byte vector) -> (y, mean_rstd, mask)."""
        N, H, W, C = x.shape
        y = torch.empty_like(x)
        mr = torch.empty((N, C, 2), dtype=torch.float32, device=x.device)
        mask = torch.empty((x.numel() * x.element_size() // 16,), dtype=torch.uint8, device=x.device) if want_mask else None
        st = self._timed('in_fwd', 0.0, lambda: self.lib.eve_instnorm_fwd_fused(
            dt_code(x.dtype), N, H * W, C, self._p(x), self._p(self._f32(gamma, 'gamma')), self._p(self._f32(beta, 'beta')),
            self._p(res), act, eps, self._p(y), self._p(mr), self._p(mask), self._stream()), (x, res, y, mask))
        if st == -1:
            if self.prof:
                self.prof.pop()          # nothing was launched
            return None
        self._ck(st)
        return (y, mr, mask) if want_mask else (y, mr)

    def instnorm_bwd_fused(self, dy, y, x, mr, gamma, act, want_dres, dy2=None, mask=None, beta=None, dx_add=None):
        """Single-launch backward; returns (dx, dres, sums) or None when the plane is too large.
        dy2: optional second summand of the incoming gradient (added on load).
        beta: with gamma and y = None (no residual), act' is recomputed from x with the forward's scale / shift.
        mask: the forward's sign mask (ReLU only), read instead of y."""
        N, H, W, C = x.shape
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if want_dres else None
        sums = torch.empty((N, C, 2), dtype=torch.float32, device=x.device)
        st = self._timed('in_bwd', 0.0, lambda: self.lib.eve_instnorm_bwd_fused(
            dt_code(x.dtype), N, H * W, C, self._p(dy), self._p(dy2), self._p(None if mask is not None else y), self._p(x), self._p(mr),
            self._p(self._f32(gamma, 'gamma')), self._p(self._f32(beta, 'beta')), act, self._p(dx), self._p(dres), self._p(sums), self._p(mask),
            self._p(dx_add), self._stream()), (dy, dy2, y if mask is None else mask, x, dx, dres, dx_add))
        if st == -1:
            if self.prof:
                self.prof.pop()
            return None
        self._ck(st)
        return dx, dres, sums

    def sum_rows(self, t):
        """[N, ...] float32 -> [...]: the sum over the leading dimension in a fixed order (per-plane partials -> parameter gradient)."""
        assert t.dtype == torch.float32 and t.is_contiguous()
        out = torch.empty(t.shape[1:], dtype=torch.float32, device=t.device)
        self._ck(self.lib.eve_sum_rows(t.shape[0], out.numel(), self._p(t), self._p(out), self._stream()))
        return out

    def sum_rows_pairs(self, sums, out0, out1):
        """sums [N, C, 2] float32: out0 += sums[:, :, 0].sum(0), out1 += sums[:, :, 1].sum(0), fixed order, in place (contiguous
        float32 [C] vectors: an affine InstanceNorm's d beta / d gamma inside the flat gradient buffer)."""
        N, C, two = sums.shape
        assert two == 2 and sums.dtype == torch.float32 and sums.is_contiguous()
        for o in (out0, out1):
            assert o.dtype == torch.float32 and o.is_contiguous() and o.numel() == C
        self._ck(self.lib.eve_sum_rows_pairs(N, C, self._p(sums), self._p(out0), self._p(out1), self._stream()))

    # ------------------------------------------------------------------ element-wise
    def act_bwd(self, dy, y, act):
        dx = torch.empty_like(dy)
        self._ck(self.lib.eve_act_bwd(dt_code(dy.dtype), dy.numel(), self._p(dy), self._p(y), act,
                                      self._p(dx), self._stream()))
        return dx

    def add(self, a, b):
        out = torch.empty_like(a)
        self._ck(self.lib.eve_add(dt_code(a.dtype), a.numel(), self._p(a), self._p(b), self._p(out),
                                  self._stream()))
        return out

    # ------------------------------------------------------------------ pooling / resampling
    def maxpool3x3s2_fwd(self, x):
        N, IH, IW, C = x.shape
        OH, OW = (IH - 1) // 2 + 1, (IW - 1) // 2 + 1
        y = torch.empty((N, OH, OW, C), dtype=x.dtype, device=x.device)
        idx = torch.empty((N, OH, OW, C), dtype=torch.uint8, device=x.device)
        self._ck(self.lib.eve_maxpool3x3s2_fwd(dt_code(x.dtype), N, IH, IW, C, self._p(x), self._p(y),
                                               self._p(idx), self._stream()))
        return y, idx

    def maxpool3x3s2_bwd(self, dy, idx, in_hw):
        N, OH, OW, C = dy.shape
        IH, IW = in_hw
        dx = torch.empty((N, IH, IW, C), dtype=dy.dtype, device=dy.device)
        self._ck(self.lib.eve_maxpool3x3s2_bwd(dt_code(dy.dtype), N, IH, IW, C, self._p(dy), self._p(idx),
                                               self._p(dx), self._stream()))
        return dx

    def in_relu_maxpool_fwd(self, x, mr):
        N, IH, IW, C = x.shape
        OH, OW = (IH - 1) // 2 + 1, (IW - 1) // 2 + 1
        y = torch.empty((N, OH, OW, C), dtype=x.dtype, device=x.device)
        idx = torch.empty((N, OH, OW, C), dtype=torch.uint8, device=x.device)
        self._ck(self.lib.eve_in_relu_maxpool_fwd(dt_code(x.dtype), N, IH, IW, C, self._p(x), self._p(mr),
                                                  self._p(y), self._p(idx), self._stream()))
        return y, idx

    def in_relu_maxpool_bwd(self, dy_pool, y_pool, idx, x, mr):
        N, IH, IW, C = x.shape
        dx = torch.empty_like(x)
        self._ck(self.lib.eve_in_relu_maxpool_bwd(dt_code(x.dtype), N, IH, IW, C, self._p(dy_pool), self._p(y_pool),
                                                  self._p(idx), self._p(x), self._p(mr), self._p(dx),
                                                  self._stream()))
        return dx

    def avgpool_fwd(self, x):
        N, H, W, C = x.shape
        y = torch.empty((N, C), dtype=x.dtype, device=x.device)
        self._ck(self.lib.eve_avgpool_fwd(dt_code(x.dtype), N, H * W, C, self._p(x), self._p(y),
                                          self._stream()))
        return y

    def avgpool_fwd_f32(self, x):
        """avgpool_fwd with a float32 result holding the x.dtype-rounded means (= avgpool_fwd + cast, one launch)."""
        N, H, W, C = x.shape
        y = torch.empty((N, C), dtype=torch.float32, device=x.device)
        self._ck(self.lib.eve_avgpool_fwd_f32(dt_code(x.dtype), N, H * W, C, self._p(x), self._p(y), self._stream()))
        return y

    def avgpool_bwd_f32(self, dy, hw, dtype):
        """dy float32 [N, C] -> dx [N, H, W, C] in `dtype` (= cast + avgpool_bwd, one launch)."""
        N, C = dy.shape
        assert dy.dtype == torch.float32 and dy.is_contiguous()
        dx = torch.empty((N, hw[0], hw[1], C), dtype=dtype, device=dy.device)
        self._ck(self.lib.eve_avgpool_bwd_f32(dt_code(dtype), N, hw[0] * hw[1], C, self._p(dy), self._p(dx), self._stream()))
        return dx

    def avgpool_bwd(self, dy, hw):
        N, C = dy.shape
        dx = torch.empty((N, hw[0], hw[1], C), dtype=dy.dtype, device=dy.device)
        self._ck(self.lib.eve_avgpool_bwd(dt_code(dy.dtype), N, hw[0] * hw[1], C, self._p(dy), self._p(dx),
                                          self._stream()))
        return dx

    def adaptive_maxpool_fwd(self, x, out_hw):
        N, IH, IW, C = x.shape
        OH, OW = out_hw
        y = torch.empty((N, OH, OW, C), dtype=x.dtype, device=x.device)
        idx = torch.empty((N, OH, OW, C), dtype=torch.int32, device=x.device)
        self._ck(self.lib.eve_adaptive_maxpool_fwd(dt_code(x.dtype), N, IH, IW, OH, OW, C, self._p(x),
                                                   self._p(y), self._p(idx), self._stream()))
        return y, idx

    def adaptive_maxpool_bwd(self, dy, idx, in_hw, add=None):
        """add: a second gradient of the pool's input ([N, IH, IW, C]), summed in the kernel epilogue."""
        N, OH, OW, C = dy.shape
        IH, IW = in_hw
        dx = torch.empty((N, IH, IW, C), dtype=dy.dtype, device=dy.device)
        assert add is None or (tuple(add.shape) == (N, IH, IW, C) and add.dtype == dy.dtype)
        self._ck(self.lib.eve_adaptive_maxpool_bwd(dt_code(dy.dtype), N, IH, IW, OH, OW, C, self._p(dy),
                                                   self._p(idx), self._p(add), self._p(dx), self._stream()))
        return dx

    def bilinear_fwd(self, x, out_hw):
        N, IH, IW, C = x.shape
        OH, OW = out_hw
        y = torch.empty((N, OH, OW, C), dtype=x.dtype, device=x.device)
        self._ck(self.lib.eve_bilinear_fwd(dt_code(x.dtype), N, IH, IW, OH, OW, C, self._p(x), self._p(y),
                                           self._stream()))
        return y

    def bilinear_bwd(self, dy, in_hw):
        N, OH, OW, C = dy.shape
        IH, IW = in_hw
        dx = torch.empty((N, IH, IW, C), dtype=dy.dtype, device=dy.device)
        self._ck(self.lib.eve_bilinear_bwd(dt_code(dy.dtype), N, IH, IW, OH, OW, C, self._p(dy), self._p(dx),
                                           self._stream()))
        return dx

    # ------------------------------------------------------------------ layout / dtype
    def nchw_to_nhwc(self, src, dtype, cpad=None, out=None):
        N, C, H, W = src.shape
        cpad = cpad or pad_channels(C, dtype)
        src = src.contiguous()
        dst = out if out is not None else torch.empty((N, H, W, cpad), dtype=dtype, device=src.device)
        assert tuple(dst.shape) == (N, H, W, cpad) and dst.dtype == dtype
        self._ck(self.lib.eve_nchw_to_nhwc(dt_code(dtype), N, C, H, W, cpad, self._p(self._f32(src, 'src')),
                                           self._p(dst), self._stream()))
        return dst

    def nhwc_to_nchw(self, src, C):
        N, H, W, cpad = src.shape
        dst = torch.empty((N, C, H, W), dtype=torch.float32, device=src.device)
        self._ck(self.lib.eve_nhwc_to_nchw(dt_code(src.dtype), N, C, H, W, cpad, self._p(src), self._p(dst),
                                           self._stream()))
        return dst

    def cast(self, src, dtype):
        if src.dtype == dtype:
            return src
        dst = torch.empty(src.shape, dtype=dtype, device=src.device)
        self._ck(self.lib.eve_cast(dt_code(src.dtype), dt_code(dtype), src.numel(), self._p(src), self._p(dst),
                                   self._stream()))
        return dst

    def pack_weights(self, w_ohwi_f32, dtype, want_ihwo=True):
        Cout, KH, KW, Cin = w_ohwi_f32.shape
        ohwi = torch.empty((Cout, KH, KW, Cin), dtype=dtype, device=w_ohwi_f32.device)
        ihwo = torch.empty((Cin, KH, KW, Cout), dtype=dtype, device=w_ohwi_f32.device) if want_ihwo else None
        self._ck(self.lib.eve_pack_weights(dt_code(dtype), Cout, KH * KW, Cin,
                                           self._p(self._f32(w_ohwi_f32, 'weights')), self._p(ohwi),
                                           self._p(ihwo), self._stream()))
        return ohwi, ihwo

    def pack_weights_batch(self, ws_ohwi_f32, dtype, want_ihwo, padded=None):
        """pack_weights for a list of OHWI float32 weights (and per-weight want_ihwo flags) in ONE launch per
        EVE_PACK_BATCH_MAX weights.  padded: per weight None or (Cout_padded, Cin_padded) -- the packed copies are that wide, the
        extra channels zero (written by the same launch).  Returns [(ohwi, ihwo)]."""
        out, items, keep = [], [], []
        padded = padded if padded is not None else [None] * len(ws_ohwi_f32)
        for w, wi, pd in zip(ws_ohwi_f32, want_ihwo, padded):
            sCout, KH, KW, sCin = w.shape
            Cout, Cin = pd if pd is not None else (sCout, sCin)
            assert Cout >= sCout and Cin >= sCin
            w = self._f32(w, 'weights')
            self._p(w)                                 # (raises for a tensor that is not on the GPU)
            ohwi = torch.empty((Cout, KH, KW, Cin), dtype=dtype, device=w.device)
            ihwo = torch.empty((Cin, KH, KW, Cout), dtype=dtype, device=w.device) if wi else None
            keep.append(w)
            items.append(_lib.PackItem(w.data_ptr(), ohwi.data_ptr(), ihwo.data_ptr() if wi else None, Cout, KH * KW, Cin, sCout, sCin))
            out.append((ohwi, ihwo))
        for i in range(0, len(items), _lib.PACK_BATCH_MAX):
            chunk = items[i:i + _lib.PACK_BATCH_MAX]
            arr = (_lib.PackItem * len(chunk))(*chunk)
            self._ck(self.lib.eve_pack_weights_batch(dt_code(dtype), len(chunk), arr, self._stream()))
        return out

    # ------------------------------------------------------------------ recurrent
    def gru_scan_fwd(self, gi, whh_t, bhh, h0):
        S, T, H3 = gi.shape
        H = H3 // 3
        dev = gi.device
        hs = torch.empty((S, T, H), dtype=torch.float32, device=dev)
        gates = torch.empty((S, T, H3), dtype=torch.float32, device=dev)
        hn_pre = torch.empty((S, T, H), dtype=torch.float32, device=dev)
        self._ck(self.lib.eve_gru_scan_fwd(S, T, H, self._p(self._f32(gi, 'gi')), self._p(whh_t), self._p(bhh),
                                           self._p(h0), self._p(hs), self._p(gates), self._p(hn_pre),
                                           self._stream()))
        return hs, gates, hn_pre

    def gru_scan_bwd(self, dhs, whh, h0, hs, gates, hn_pre, want_dh0):
        S, T, H = dhs.shape
        dev = dhs.device
        dgi = torch.empty((S, T, 3 * H), dtype=torch.float32, device=dev)
        dgh = torch.empty((S, T, 3 * H), dtype=torch.float32, device=dev)
        dh0 = torch.empty((S, H), dtype=torch.float32, device=dev) if want_dh0 else None
        self._ck(self.lib.eve_gru_scan_bwd(S, T, H, self._p(dhs), self._p(whh), self._p(h0), self._p(hs),
                                           self._p(gates), self._p(hn_pre), self._p(dgi), self._p(dgh),
                                           self._p(dh0), self._stream()))
        return dgi, dgh, dh0

    def rnn_scan_fwd(self, gi, whh_t, bhh, h0):
        S, T, H = gi.shape
        hs = torch.empty((S, T, H), dtype=torch.float32, device=gi.device)
        self._ck(self.lib.eve_rnn_scan_fwd(S, T, H, self._p(self._f32(gi, 'gi')), self._p(whh_t), self._p(bhh), self._p(h0),
                                           self._p(hs), self._stream()))
        return hs

    def rnn_scan_bwd(self, dhs, whh, hs, want_dh0):
        S, T, H = dhs.shape
        dpre = torch.empty((S, T, H), dtype=torch.float32, device=dhs.device)
        dh0 = torch.empty((S, H), dtype=torch.float32, device=dhs.device) if want_dh0 else None
        self._ck(self.lib.eve_rnn_scan_bwd(S, T, H, self._p(dhs), self._p(whh), self._p(hs), self._p(dpre), self._p(dh0),
                                           self._stream()))
        return dpre, dh0

    def lstm_scan_fwd(self, gi, whh_t, bhh, h0, c0):
        S, T, H4 = gi.shape
        H = H4 // 4
        dev = gi.device
        hs = torch.empty((S, T, H), dtype=torch.float32, device=dev)
        cs = torch.empty((S, T, H), dtype=torch.float32, device=dev)
        gates = torch.empty((S, T, H4), dtype=torch.float32, device=dev)
        self._ck(self.lib.eve_lstm_scan_fwd(S, T, H, self._p(self._f32(gi, 'gi')), self._p(whh_t), self._p(bhh), self._p(h0),
                                            self._p(c0), self._p(hs), self._p(cs), self._p(gates), self._stream()))
        return hs, cs, gates

    def lstm_scan_bwd(self, dhs, dcs, whh, c0, hs, cs, gates, want_d0):
        S, T, H = dhs.shape
        dev = dhs.device
        dpre = torch.empty((S, T, 4 * H), dtype=torch.float32, device=dev)
        dh0 = torch.empty((S, H), dtype=torch.float32, device=dev) if want_d0 else None
        dc0 = torch.empty((S, H), dtype=torch.float32, device=dev) if want_d0 else None
        self._ck(self.lib.eve_lstm_scan_bwd(S, T, H, self._p(dhs), self._p(dcs), self._p(whh), self._p(c0), self._p(hs),
                                            self._p(cs), self._p(gates), self._p(dpre), self._p(dh0), self._p(dc0),
                                            self._stream()))
        return dpre, dh0, dc0

    def cgru_scan_fwd(self, xs, h0, w1_ohwi, b1, w2_ohwi, b2):
        """CGRUCell over T in one launch.  xs [B, T, 5, 8, 64] bf16 / fp16 / float32 -> hs [B, T, 5, 8, 64] and, time-major
        [T, B, 5, 8, .] for the backward: hs_tm, ru, rh, og.  (float32: csrc/cell_scan_f32.hip.)"""
        B, T, H, W, C = xs.shape
        assert (H, W, C) == (5, 8, 64) and xs.dtype in HALF_DTYPES + (torch.float32,) and xs.is_contiguous()
        assert w1_ohwi.dtype == w2_ohwi.dtype == xs.dtype and w1_ohwi.is_contiguous() and w2_ohwi.is_contiguous()
        assert tuple(w1_ohwi.shape) == (128, 3, 3, 128) and tuple(w2_ohwi.shape) == (64, 3, 3, 128)
        dev = xs.device
        hdt = xs.dtype
        hs = torch.empty((B, T, H, W, C), dtype=hdt, device=dev)
        hs_tm = torch.empty((T, B, H, W, C), dtype=hdt, device=dev)
        ru = torch.empty((T, B, H, W, 2 * C), dtype=hdt, device=dev)
        rh = torch.empty((T, B, H, W, C), dtype=hdt, device=dev)
        og = torch.empty((T, B, H, W, C), dtype=hdt, device=dev)
        self._ck(self.lib.eve_cgru_scan_fwd(dt_code(hdt), B, T, self._p(xs), self._p(h0), self._p(w1_ohwi), self._p(self._f32(b1, 'b1')),
                                            self._p(w2_ohwi), self._p(self._f32(b2, 'b2')), self._p(hs), self._p(hs_tm),
                                            self._p(ru), self._p(rh), self._p(og), self._stream()))
        return hs, hs_tm, ru, rh, og

    def cgru_scan_bwd(self, dhs_tm, ru, og, hs_tm, h0, w1_ihwo, w2_ihwo, want_dh0=False):
        """Backward of cgru_scan_fwd in one launch.  Time-major [T, B, 5, 8, .] inputs (16-bit or float32); returns (dg1_all
        [T,B,5,8,128], dg2_all [T,B,5,8,64], dxs_tm [T,B,5,8,64], dh0 [B,5,8,64] or None)."""
        T, B, H, W, C = dhs_tm.shape
        assert (H, W, C) == (5, 8, 64) and dhs_tm.dtype in HALF_DTYPES + (torch.float32,) and dhs_tm.is_contiguous()
        assert w1_ihwo.dtype == w2_ihwo.dtype == dhs_tm.dtype and w1_ihwo.is_contiguous() and w2_ihwo.is_contiguous()
        assert all(t.is_contiguous() and t.dtype == dhs_tm.dtype for t in (ru, og, hs_tm))
        assert tuple(ru.shape) == (T, B, H, W, 2 * C) and tuple(og.shape) == tuple(hs_tm.shape) == (T, B, H, W, C)
        assert tuple(w1_ihwo.shape) == (128, 3, 3, 128) and tuple(w2_ihwo.shape) == (128, 3, 3, 64)
        dev = dhs_tm.device
        hdt = dhs_tm.dtype
        dg1 = torch.empty((T, B, H, W, 2 * C), dtype=hdt, device=dev)
        dg2 = torch.empty((T, B, H, W, C), dtype=hdt, device=dev)
        dxs = torch.empty((T, B, H, W, C), dtype=hdt, device=dev)
        dh0 = torch.empty((B, H, W, C), dtype=hdt, device=dev) if want_dh0 else None
        self._ck(self.lib.eve_cgru_scan_bwd(dt_code(hdt), B, T, self._p(dhs_tm), self._p(ru), self._p(og), self._p(hs_tm), self._p(h0),
                                            self._p(w1_ihwo), self._p(w2_ihwo), self._p(dg1), self._p(dg2), self._p(dxs),
                                            self._p(dh0), self._stream()))
        return dg1, dg2, dxs, dh0

    def crnn_scan_fwd(self, xs, h0, w_ohwi, bias):
        """CRNNCell over T in one launch (float32, csrc/cell_scan_f32.hip).  xs [B, T, 5, 8, 64] -> (hs [B, T, 5, 8, 64], hs_tm
        [T, B, 5, 8, 64])."""
        B, T, H, W, C = xs.shape
        assert (H, W, C) == (5, 8, 64) and xs.dtype == torch.float32 and xs.is_contiguous()
        assert tuple(w_ohwi.shape) == (64, 3, 3, 128) and w_ohwi.dtype == torch.float32 and w_ohwi.is_contiguous()
        assert h0 is None or (h0.dtype == torch.float32 and h0.is_contiguous() and tuple(h0.shape) == (B, H, W, C))
        hs = torch.empty((B, T, H, W, C), dtype=torch.float32, device=xs.device)
        hs_tm = torch.empty((T, B, H, W, C), dtype=torch.float32, device=xs.device)
        self._ck(self.lib.eve_crnn_scan_fwd(B, T, self._p(xs), self._p(h0), self._p(w_ohwi), self._p(self._f32(bias, 'bias')),
                                            self._p(hs), self._p(hs_tm), self._stream()))
        return hs, hs_tm

    def crnn_scan_bwd(self, dhs_tm, hs_tm, w_ihwo, want_dh0=False):
        """Backward of crnn_scan_fwd in one launch: (dpre_all, dxs_tm [T, B, 5, 8, 64], dh0 [B, 5, 8, 64] or None)."""
        T, B, H, W, C = dhs_tm.shape
        assert (H, W, C) == (5, 8, 64) and dhs_tm.dtype == hs_tm.dtype == torch.float32
        assert dhs_tm.is_contiguous() and hs_tm.is_contiguous() and tuple(hs_tm.shape) == (T, B, H, W, C)
        assert tuple(w_ihwo.shape) == (128, 3, 3, 64) and w_ihwo.dtype == torch.float32 and w_ihwo.is_contiguous()
        dpre = torch.empty((T, B, H, W, C), dtype=torch.float32, device=dhs_tm.device)
        dxs = torch.empty((T, B, H, W, C), dtype=torch.float32, device=dhs_tm.device)
        dh0 = torch.empty((B, H, W, C), dtype=torch.float32, device=dhs_tm.device) if want_dh0 else None
        self._ck(self.lib.eve_crnn_scan_bwd(B, T, self._p(dhs_tm), self._p(hs_tm), self._p(w_ihwo), self._p(dpre), self._p(dxs),
                                            self._p(dh0), self._stream()))
        return dpre, dxs, dh0

    def clstm_scan_fwd(self, xs, h0, c0, w_ohwi, bias):
        """CLSTMCell over T in one launch (float32, forward only).  xs [B, T, 5, 8, 64] -> (hs, cs) [B, T, 5, 8, 64]."""
        B, T, H, W, C = xs.shape
        assert (H, W, C) == (5, 8, 64) and xs.dtype == torch.float32 and xs.is_contiguous()
        assert tuple(w_ohwi.shape) == (256, 3, 3, 128) and w_ohwi.dtype == torch.float32 and w_ohwi.is_contiguous()
        for s0 in (h0, c0):
            assert s0 is None or (s0.dtype == torch.float32 and s0.is_contiguous() and tuple(s0.shape) == (B, H, W, C))
        hs = torch.empty((B, T, H, W, C), dtype=torch.float32, device=xs.device)
        cs = torch.empty((B, T, H, W, C), dtype=torch.float32, device=xs.device)
        self._ck(self.lib.eve_clstm_scan_fwd(B, T, self._p(xs), self._p(h0), self._p(c0), self._p(w_ohwi),
                                             self._p(self._f32(bias, 'bias')), self._p(hs), self._p(cs), self._stream()))
        return hs, cs

    def cgru_gates1(self, g1, h):
        C = h.shape[-1]
        P = h.numel() // C
        ru = torch.empty_like(g1)
        rh = torch.empty_like(h)
        self._ck(self.lib.eve_cgru_gates1(dt_code(h.dtype), P, C, self._p(g1), self._p(h), self._p(ru),
                                          self._p(rh), self._stream()))
        return ru, rh

    def cgru_gates2(self, g2, ru, h):
        C = h.shape[-1]
        P = h.numel() // C
        o = torch.empty_like(h)
        hnew = torch.empty_like(h)
        self._ck(self.lib.eve_cgru_gates2(dt_code(h.dtype), P, C, self._p(g2), self._p(ru), self._p(h),
                                          self._p(o), self._p(hnew), self._stream()))
        return o, hnew

    def cgru_gates2_bwd(self, dhnew, ru, h, o):
        C = h.shape[-1]
        P = h.numel() // C
        dg2 = torch.empty_like(h)
        dru = torch.empty_like(ru)
        dh = torch.empty_like(h)
        self._ck(self.lib.eve_cgru_gates2_bwd(dt_code(h.dtype), P, C, self._p(dhnew), self._p(ru), self._p(h),
                                              self._p(o), self._p(dg2), self._p(dru), self._p(dh),
                                              self._stream()))
        return dg2, dru, dh

    def cgru_gates1_bwd(self, drh, dru, ru, h):
        C = h.shape[-1]
        P = h.numel() // C
        dg1 = torch.empty_like(ru)
        dh = torch.empty_like(h)
        self._ck(self.lib.eve_cgru_gates1_bwd(dt_code(h.dtype), P, C, self._p(drh), self._p(dru), self._p(ru),
                                              self._p(h), self._p(dg1), self._p(dh), self._stream()))
        return dg1, dh

    def clstm_gates_fwd(self, gates, c_prev):
        C = c_prev.shape[-1]
        P = c_prev.numel() // C
        h = torch.empty_like(c_prev)
        c = torch.empty_like(c_prev)
        self._ck(self.lib.eve_clstm_gates_fwd(dt_code(c_prev.dtype), P, C, self._p(gates), self._p(c_prev),
                                              self._p(h), self._p(c), self._stream()))
        return h, c

    # ------------------------------------------------------------------ optimiser
    # ------------------------------------------------------------------ masked [B, T, D] loss / metric terms, batched
    VEC_KINDS = {'mse': 0, 'euclidean': 1, 'l1': 2, 'angular': 3}

    def vector_terms(self, items, want_grad):
        """items: list of (kind, pred [B, T(, D)], target, validity [B, T]) float32 / bool on the GPU; want_grad: per item, whether
        d term / d pred is wanted (not for 'euclidean').  ONE launch per 32 terms -> (out [n] float32, [dpred or None])."""
        B, T = items[0][3].shape
        dev = items[0][1].device
        out = torch.empty((len(items),), dtype=torch.float32, device=dev)
        keep, dps, recs = [], [], []
        for (kind, pred, tgt, val), wg in zip(items, want_grad):
            D = 1 if pred.dim() == 2 else pred.shape[2]
            pred, tgt = pred.contiguous(), tgt.contiguous()
            val = val.contiguous()
            val = val.view(torch.uint8) if val.dtype == torch.bool else val.to(torch.uint8)
            assert pred.dtype == tgt.dtype == torch.float32 and tuple(pred.shape) == tuple(tgt.shape) and tuple(val.shape) == (B, T)
            assert tuple(pred.shape[:2]) == (B, T) and 1 <= D <= 3 and pred.dim() <= 3
            dp = torch.empty_like(pred) if wg else None
            keep.append((pred, tgt, val))
            dps.append(dp)
            recs.append(_lib.VecTerm(self._p(pred), self._p(tgt), self._p(val), self._p(dp), D, self.VEC_KINDS[kind]))
        for i in range(0, len(recs), _lib.VEC_TERMS_MAX):
            chunk = recs[i:i + _lib.VEC_TERMS_MAX]
            arr = (_lib.VecTerm * len(chunk))(*chunk)
            self._ck(self.lib.eve_vector_terms(arr, len(chunk), B, T, ctypes.c_void_p(out.data_ptr() + 4 * i), self._stream()))
        return out, dps

    # ------------------------------------------------------------------ stream gates (parallel.GradSync)
    def gate_signal(self, flags, index):
        """One-thread kernel on the current stream (capturable): release, then flags[index] += 1."""
        assert flags.dtype == torch.int32 and flags.is_contiguous() and 0 <= index < flags.numel()
        self._ck(self.lib.eve_gate_signal(ctypes.c_void_p(flags.data_ptr() + 4 * index), self._stream()))

    def gate_wait(self, flags, index, value, timeouts, value_index=None, poison=None, max_polls=0):
        """One-wave kernel on the current stream that returns once flags[index] has reached `value` -- or flags[value_index],
        read on the device, when value_index is given (a capturable wait: the target is not baked into the node).  Bounded poll
        (max_polls; 0 = eve_dispatch_config.gate_wait_polls, seconds): a gate that never opens increments timeouts[0] and writes
        +inf to poison[0] (a float32 tensor inside the last all-reduced gradient bucket: adam_step(poison=...) then skips the
        update on every rank) instead of hanging the device."""
        assert flags.dtype == timeouts.dtype == torch.int32 and 0 <= index < flags.numel()
        assert poison is None or (poison.dtype == torch.float32 and poison.numel() >= 1)
        ref = None if value_index is None else ctypes.c_void_p(flags.data_ptr() + 4 * value_index)
        self._ck(self.lib.eve_gate_wait(ctypes.c_void_p(flags.data_ptr() + 4 * index), int(value) & 0xffffffff, ref, self._p(timeouts),
                                        self._p(poison), int(max_polls), self._stream()))

    SUMSQ_WORKSPACE = 1024          # include/eve_hip.h EVE_SUMSQ_WORKSPACE

    def sumsq(self, g, out, workspace=None):
        """out[0] += sum g^2, in a fixed summation order (bit-reproducible).  workspace: float32 [1024] scratch."""
        if workspace is None:
            workspace = torch.empty((self.SUMSQ_WORKSPACE,), dtype=torch.float32, device=g.device)
        assert workspace.numel() >= self.SUMSQ_WORKSPACE and workspace.dtype == torch.float32
        self._ck(self.lib.eve_sumsq(g.numel(), self._p(self._f32(g, 'g')), self._p(out), self._p(workspace), self._stream()))
        return out

    ADAM_GUARD_WORDS = 12        # include/eve_hip.h eve_adam_guard: 4 ints, then floats (loss_scale at word 4), skipped_gate (int) at word 9

    @staticmethod
    def new_adam_guard(device, loss_scale=1.0, step=0):
        """Device-resident optimiser state (eve_adam_guard) as an int32 tensor of 12 words; .view(torch.float32)[4] is the loss scale."""
        g = torch.zeros(HipKernels.ADAM_GUARD_WORDS, dtype=torch.int32, device=device)
        g[0] = int(step)
        g.view(torch.float32)[4] = float(loss_scale)
        return g

    def adam_step(self, p, g, m, v, sumsq, max_norm, gscale, lr, beta1, beta2, eps, weight_decay, step,
                  guard=None, check_finite=False, lr_dev=None, poison=None):
        """poison: float32 device word (gate_wait writes +inf there on a time-out); non-zero skips the step on the device."""
        if guard is not None:
            assert guard.dtype == torch.int32 and guard.numel() >= self.ADAM_GUARD_WORDS and guard.is_contiguous()
        assert poison is None or (guard is not None and poison.dtype == torch.float32)
        self._ck(self.lib.eve_adam_step(p.numel(), self._p(p), self._p(g), self._p(m), self._p(v),
                                        self._p(sumsq), max_norm, gscale, lr, beta1, beta2, eps,
                                        weight_decay, step, self._p(guard), 1 if check_finite else 0, self._p(lr_dev), self._p(poison),
                                        self._stream()))


def dispatch_flag(k, name, default):
    """Field `name` of k's kernel-selection table (include/eve_hip.h eve_dispatch_config: resolved once when the library is
    loaded, overridden only through eve_set_dispatch_config); `default` for a test stand-in that has no table."""
    if hasattr(k, 'dispatch_config'):
        return int(getattr(k.dispatch_config(), name))
    return default


_default = None


def default_kernels():
    """The process-wide HipKernels; raises (no fallback) if libeve_hip.so is not built."""
    global _default
    if _default is None:
        _default = HipKernels()
    return _default


def set_default_kernels(k):
    """Test hook: tests/ installs a torch-CPU stand-in to exercise the HOST logic without a GPU."""
    global _default
    _default = k
