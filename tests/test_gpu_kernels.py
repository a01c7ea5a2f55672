"""GPU: every HIP kernel, called through the C ABI (eve_amd.kernels.HipKernels -> libeve_hip.so),
against an independent ATen restatement of its contract (tests/fake_kernels.py run on the CPU).

float32 kernels must agree to float32 round-off (the f32 MFMA is an exact fmaf chain);
bfloat16 / float16 kernels (the same templates over the 16-bit format) are compared at that format's resolution with
inputs pre-rounded to it on both sides.
"""
import numpy as np
import pytest
import torch

from fake_kernels import FakeKernels

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16, torch.float16]
DT_IDS = ['f32', 'bf16', 'fp16']
HALVES = pytest.mark.parametrize('hdt', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])


@pytest.fixture(scope='module')
def hip():
    from eve_amd.kernels import HipKernels
    assert torch.cuda.is_available(), 'GPU suite needs a GPU'
    return HipKernels()


@pytest.fixture(scope='module')
def ref():
    return FakeKernels()


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def dev(t):
    return None if t is None else t.cuda()


def close(got, want, dtype, what, scale=None):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, '%s: shape %s vs %s' % (what, got.shape, want.shape)
    assert torch.isfinite(got).all(), '%s: non-finite values' % what
    s = float(want.abs().max()) if scale is None else scale
    tol = {torch.float32: 3e-5, torch.bfloat16: 1.6e-2, torch.float16: 2e-3}[dtype] * max(s, 1e-6)
    err = float((got - want).abs().max())
    assert err <= tol, '%s: max|diff| %.3e > tol %.3e (scale %.3e)' % (what, err, tol, s)
    # and in the mean: both sides round the same float32 result to bf16, so only a few elements may sit one ulp apart;
    # a defect of relative size 1e-2 anywhere in the kernel fails this
    rel = float((got - want).norm()) / max(float(want.norm()), 1e-30)
    assert rel <= {torch.float32: 2e-5, torch.bfloat16: 3e-3, torch.float16: 4e-4}[dtype] or err <= 1e-6 * max(s, 1e-6), \
        '%s: relative L2 %.3e' % (what, rel)


CONV_CASES = [
    # N, IH, IW, Cin, Cout, K, stride, pad
    (3, 32, 32, 64, 64, 3, 1, 1),       # layer1
    (2, 32, 32, 64, 128, 3, 2, 1),      # layerX.0 conv1 (stride 2)
    (2, 32, 32, 64, 128, 1, 2, 0),      # down-sample 1x1 / 2
    (5, 4, 4, 512, 512, 3, 1, 1),       # layer4 (16 pixels per image, M = 80: ragged tile)
    (2, 128, 128, 8, 64, 7, 2, 3),      # stem (Cin 3 padded)
    (3, 5, 8, 128, 128, 3, 1, 1),       # conv-GRU gates on the 5x8 map
    (2, 9, 16, 512, 128, 3, 1, 1),      # RefineNet decoder (concatenated input)
    (1, 72, 128, 16, 16, 3, 1, 1),      # RefineNet outer level
    (2, 72, 128, 16, 8, 1, 1, 0),       # RefineNet final 1x1 (Cout padded)
    (37, 1, 1, 512, 128, 1, 1, 0),      # Linear 512 -> 128
    (37, 1, 1, 128, 384, 1, 1, 0),      # GRU input GEMM
    (4, 16, 16, 128, 128, 3, 1, 1),     # layer2 (LDS-DMA kernel, 2x2 waves, several M tiles)
    (3, 8, 8, 256, 256, 3, 1, 1),       # layer3
    (7, 32, 32, 64, 64, 3, 1, 1),       # layer1, M = 7168 (28 tiles of 256, 4x1 waves)
    (2, 16, 16, 128, 256, 3, 2, 1),     # stride-2 forward through the DMA kernel, generic dgrad
    (3, 9, 16, 128, 64, 1, 1, 0),       # 1x1, Cout 64
    (2, 36, 64, 32, 32, 3, 1, 1),       # RefineNet 36x64 level: weights-stationary kernel, one 32-channel slice
    (2, 18, 32, 64, 64, 3, 1, 1),       # RefineNet 18x32 level: ragged last band (18 rows, 8 per tile)
    (5, 8, 8, 64, 48, 3, 1, 1),         # several whole images per tile, Cout not a multiple of 16
    (300, 32, 32, 64, 64, 3, 1, 1),     # more tiles than persistent workgroups
    (530, 16, 16, 32, 128, 3, 1, 1),    # persistent 2x2-wave kernel (1060 tiles), one slice, two channel blocks... of one
    # stride-2 3x3 layers with enough tiles (>= 48) for the eight-wave kernels: conv3x3s2_wg8_kernel forward (parity planes of the
    # input as rotating halo stages) and conv3x3_wg8_kernel<.., NT = 2 | 4> data gradient (window convolution over dy), ragged
    # image counts (the last tile holds fewer images than its 2 / 4 / 16)
    (101, 32, 32, 64, 128, 3, 2, 1),    # layer 2.0
    (197, 16, 16, 128, 256, 3, 2, 1),   # layer 3.0
    (395, 8, 8, 256, 512, 3, 2, 1),     # layer 4.0
    (229, 16, 16, 256, 256, 3, 1, 1),   # layer 3 on 256 x 256 patches (configs[4]): conv3x3_wg8_kernel<2, 4, 16>, one image per tile
    (3, 64, 64, 64, 64, 3, 1, 1),       # layer 1 on 256 x 256 patches (configs[4]): conv3x3_ws64_kernel<.., 64>, 8 row bands per image
    (131, 32, 32, 128, 128, 3, 1, 1),   # layer 2 on 256 x 256 patches (configs[4]): conv3x3_wg8_kernel<4, 2, 32, 9, 2>, half-image bands (262 tiles)
]


def conv_case_id(c):
    return 'N%d_%dx%d_c%d-%d_k%d_s%d' % c[:7]


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
@pytest.mark.parametrize('case', CONV_CASES, ids=conv_case_id)
def test_conv_fwd_dgrad_wgrad(hip, ref, dtype, case):
    N, IH, IW, Cin, Cout, K, stride, pad = case
    if dtype != torch.float32 and (Cin % 8 or Cout % 8):
        pytest.skip('16-bit formats need 8-channel vectors')
    x = rnd((N, IH, IW, Cin), dtype, 1)
    w = rnd((Cout, K, K, Cin), dtype, 2, scale=(2.0 / (K * K * Cin)) ** 0.5)
    bias = rnd((Cout,), torch.float32, 3)
    want = ref.conv2d_fwd(x, w, bias, stride, pad)
    got = hip.conv2d_fwd(dev(x), dev(w), dev(bias), stride, pad)
    close(got, want, dtype, 'conv fwd')
    dy = rnd(tuple(want.shape), dtype, 4)
    w_ihwo = w.permute(3, 1, 2, 0).contiguous()
    want_dx = ref.conv2d_dgrad(dy, w_ihwo, (IH, IW), stride, pad)
    got_dx = hip.conv2d_dgrad(dev(dy), dev(w_ihwo), (IH, IW), stride, pad)
    close(got_dx, want_dx, dtype, 'conv dgrad')
    # the same gradient accumulated onto an existing tensor (residual join fused into the epilogue)
    base = rnd((N, IH, IW, Cin), dtype, 5)
    got_acc = hip.conv2d_dgrad(dev(dy), dev(w_ihwo), (IH, IW), stride, pad, accumulate_into=dev(base).clone())
    close(got_acc, (want_dx.float() + base.float()).to(dtype), dtype, 'conv dgrad accumulate')
    want_dw = ref.conv2d_wgrad(x, dy, K, K, stride, pad, torch.zeros((Cout, K, K, Cin)))
    got_dw = hip.conv2d_wgrad(dev(x), dev(dy), K, K, stride, pad,
                              torch.zeros((Cout, K, K, Cin), device='cuda'))
    close(got_dw, want_dw, dtype, 'conv wgrad')
    want_db = ref.bias_grad(dy, torch.zeros(Cout))
    got_db = hip.bias_grad(dev(dy), torch.zeros(Cout, device='cuda'))
    close(got_db, want_db, dtype, 'bias grad')
    # weight and bias gradient from one pass over dy (accumulating onto existing values)
    dw0, db0 = rnd((Cout, K, K, Cin), torch.float32, 6), rnd((Cout,), torch.float32, 7)
    got_dw2, got_db2 = dev(dw0).clone(), dev(db0).clone()
    hip.conv2d_wgrad(dev(x), dev(dy), K, K, stride, pad, got_dw2, db=got_db2)
    close(got_dw2, want_dw + dw0, dtype, 'conv wgrad (+bias)')
    close(got_db2, want_db + db0, dtype, 'bias grad fused into wgrad')


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
def test_conv_small_cout_and_epilogues(hip, ref, dtype):
    if dtype != torch.float32:
        pytest.skip('tail runs in float32')
    for Cout, act in ((2, 4), (1, 1), (4, 3), (6, 5), (12, 2)):
        x = rnd((19, 1, 1, 128), dtype, 5)
        w = rnd((Cout, 1, 1, 128), dtype, 6, scale=0.1)
        b = rnd((Cout,), torch.float32, 7)
        close(hip.conv2d_fwd(dev(x), dev(w), dev(b), 1, 0, epi_act=act),
              ref.conv2d_fwd(x, w, b, 1, 0, epi_act=act), dtype, 'conv Cout=%d act=%d' % (Cout, act))


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
def test_conv_prologue_fused_instnorm(hip, ref, dtype):
    N, H, W, Cin, Cout = 3, 16, 16, 64, 128
    x = rnd((N, H, W, Cin), dtype, 8)
    w = rnd((Cout, 3, 3, Cin), dtype, 9, scale=0.05)
    ss = torch.stack([1.0 + 0.2 * rnd((N, Cin), torch.float32, 10), 0.3 * rnd((N, Cin), torch.float32, 11)], -1)
    for act in (0, 1, 2):
        # the staged operand is rounded to the compute dtype on both sides
        close(hip.conv2d_fwd(dev(x), dev(w), None, 1, 1, ss=dev(ss), pro_act=act),
              ref.conv2d_fwd(x, w, None, 1, 1, ss=ss, pro_act=act), dtype, 'conv prologue act=%d' % act)
    dy = rnd((N, H, W, Cout), dtype, 12)
    close(hip.conv2d_wgrad(dev(x), dev(dy), 3, 3, 1, 1, torch.zeros((Cout, 3, 3, Cin), device='cuda'),
                           ss=dev(ss), pro_act=1),
          ref.conv2d_wgrad(x, dy, 3, 3, 1, 1, torch.zeros((Cout, 3, 3, Cin)), ss=ss, pro_act=1),
          dtype, 'wgrad prologue')


@HALVES
def test_stem_conv_dedicated_kernel(hip, ref, hdt):
    src = rnd((3, 3, 128, 128), torch.float32, 60)
    xp_w = ref.stem_pack_input(src, dtype=hdt)
    xp_g = hip.stem_pack_input(dev(src), dtype=hdt)
    assert torch.equal(xp_g.cpu().view(torch.int16), xp_w.view(torch.int16))
    w = rnd((64, 7, 7, 8), hdt, 61, scale=0.08)
    w[..., 3:] = 0
    close(hip.stem7x7s2_fwd(xp_g, dev(w)), ref.stem7x7s2_fwd(xp_w, w), hdt, 'stem 7x7/2')
    # and against the generic implicit-GEMM path on the 8-channel NHWC input
    x8 = hip.nchw_to_nhwc(dev(src), hdt, 8)
    close(hip.stem7x7s2_fwd(xp_g, dev(w)), hip.conv2d_fwd(x8, dev(w), None, 2, 3), hdt, 'stem vs generic')


@HALVES
@pytest.mark.parametrize('N', [3, 19])
def test_stem_fused_forward_matches_unfused(hip, N, hdt):
    """conv1 -> IN -> ReLU -> maxpool in one launch == the three-kernel path (which the oracle tests pin)."""
    src = rnd((N, 3, 128, 128), torch.float32, 62) + 0.3
    w = rnd((64, 7, 7, 8), hdt, 63, scale=0.08)
    w[..., 3:] = 0
    xp = hip.stem_pack_input(dev(src), dtype=hdt)
    y_f, idx_f, mr_f = hip.stem_fwd_fused(xp, dev(w))
    conv = hip.stem7x7s2_fwd(xp, dev(w))
    mr_u = hip.instnorm_stats(conv, 1e-5)
    y_u, idx_u = hip.in_relu_maxpool_fwd(conv, mr_u)
    # statistics: the fused kernel sums the fp32 accumulators, the unfused path the bf16-rounded conv output
    assert (mr_f[..., 0] - mr_u[..., 0]).abs().max().item() < 2e-3
    assert ((mr_f[..., 1] - mr_u[..., 1]).abs() / mr_u[..., 1]).max().item() < 2e-3
    d = (y_f.float() - y_u.float()).abs()
    assert d.max().item() < 0.06 and d.mean().item() < 2e-3, (d.max().item(), d.mean().item())
    # the arg-max may differ only where two window candidates tie after bf16 rounding of the unfused conv output
    assert (idx_f != idx_u).float().mean().item() < 0.02
    assert int(idx_f.max()) <= 8
    # exact semantics against an fp32 reference built from the SAME statistics and the fp32 conv
    x32 = torch.nn.functional.conv2d(xp.float().permute(0, 3, 1, 2)[:, :, :, 1:], w.float().permute(0, 3, 1, 2)[:, :4].to('cuda'),
                                     stride=2)                      # padded input: rows 0.., columns 1..
    x32 = x32[:, :, :64, :64]
    z = torch.relu((x32 - mr_f[:, :, 0, None, None]) * mr_f[:, :, 1, None, None])
    ref_y = torch.nn.functional.max_pool2d(z, 3, 2, 1).permute(0, 2, 3, 1)
    d = (y_f.float() - ref_y).abs()
    assert d.max().item() < 0.05 and d.mean().item() < 4e-3, (d.max().item(), d.mean().item())


@HALVES
@pytest.mark.parametrize('N', [3, 19, 1100])
def test_stem_forward_lean_pooling_equals_the_table_coded_kernel(hip, N, hdt):
    """Round 6: stem_fwd_pairs_kernel pools with dense arg-max codes 8 - (kh * 3 + kw) in the key bits, one key per odd column, h + 6
    for the next window's top row and straight even / odd row paths; stem_fwd_fused_kernel (one wave per image, stem_fwd_pairs = 0)
    keeps the round-1 table-coded form.  Same inputs: the SAME pooling arithmetic -- arg-max bit for bit, pooled values equal
    wherever the two kernels' statistics round alike (they sum in different float orders); N = 1 100 > 1 024 image slots: a second
    turn.  (test_stem_fused_forward_matches_unfused ties the shipped kernel to the unfused path and to the float reference.)"""
    src = rnd((N, 3, 128, 128), torch.float32, 62) + 0.3
    w = rnd((64, 7, 7, 8), hdt, 63, scale=0.08)
    w[..., 3:] = 0
    xp = hip.stem_pack_input(dev(src), dtype=hdt)
    with hip.dispatch_override(stem_fwd_pairs=0, stem_split=0):
        y1, i1, m1 = hip.stem_fwd_fused(xp, dev(w))
        assert hip.lib.eve_last_kernel().decode().startswith('stem_fwd_fused_kernel')
    y2, i2, m2 = hip.stem_fwd_fused(xp, dev(w))
    assert hip.lib.eve_last_kernel().decode().startswith('stem_fwd_pairs_kernel')
    assert torch.equal(i1, i2)
    assert float((m1[..., 0] - m2[..., 0]).abs().max()) <= 2e-6 * max(1.0, float(m1[..., 0].abs().max()))
    assert float(((m1[..., 1] - m2[..., 1]) / m1[..., 1]).abs().max()) <= 2e-5
    # the normalised values may differ by an ulp of the storage format where the two statistics differ in their last float bits
    d = (y1.float() - y2.float()).abs()
    assert float((d > 0).float().mean()) < 2e-3 and float(d.max()) <= 0.05, (float((d > 0).float().mean()), float(d.max()))


@HALVES
@pytest.mark.parametrize('N', [2, 17])
def test_stem_fused_backward_matches_unfused(hip, N, hdt):
    """d(conv1 out) by recomputation == the dense IN+ReLU+maxpool backward on the stored conv output."""
    src = rnd((N, 3, 128, 128), torch.float32, 64) + 0.2
    w = rnd((64, 7, 7, 8), hdt, 65, scale=0.08)
    w[..., 3:] = 0
    xp = hip.stem_pack_input(dev(src), dtype=hdt)
    y, idx, mr = hip.stem_fwd_fused(xp, dev(w))
    dy = dev(rnd(tuple(y.shape), hdt, 66))
    dx_f = hip.stem_bwd_dx(xp, dev(w), mr, dy, y, idx)
    conv = hip.stem7x7s2_fwd(xp, dev(w))
    dx_u = hip.in_relu_maxpool_bwd(dy, y, idx, conv, mr)
    a, b = dx_f.float(), dx_u.float()
    rel = ((a - b).norm() / b.norm()).item()
    assert rel < 6e-3, rel                       # the unfused path sees the bf16-rounded conv output
    assert (a - b).abs().max().item() < 0.05 * b.abs().max().item()


@HALVES
@pytest.mark.parametrize('N,two', [(2, False), (17, True), (1100, True)], ids=['N2', 'N17-two-summands', 'N1100-second-turn'])
def test_stem_backward_and_weight_gradient_in_one_launch(hip, N, two, hdt):
    """eve_stem_bwd_wgrad (d(conv1 out) stays in LDS: wave pairs, 32 channels each, weight-gradient MFMAs on the recomputed
    rows) == eve_stem_bwd_dx followed by eve_stem_wgrad on the stored tensor -- same rounding of d(conv1 out) to the storage
    format, float summation order aside -- and == torch's conv2d_weight on that tensor; accumulates onto existing values; the
    gradient delivered as two summands; N = 1 100 > 1 024 image slots: a second turn with idle wave pairs in it.  The plane sums and
    the masked gradient come from stem_grad_prep_kernel (the call's scratch; required since ABI v9)."""
    src = rnd((N, 3, 128, 128), torch.float32, 64) + 0.2
    w = rnd((64, 7, 7, 8), hdt, 65, scale=0.08)
    w[..., 3:] = 0
    xp = hip.stem_pack_input(dev(src), dtype=hdt)
    y, idx, mr = hip.stem_fwd_fused(xp, dev(w))
    dy = dev(rnd(tuple(y.shape), hdt, 66))
    dy2 = dev(rnd(tuple(y.shape), hdt, 69)) if two else None
    dconv = hip.stem_bwd_dx(xp, dev(w), mr, dy, y, idx, dy_pool2=dy2)
    want = torch.zeros((64, 7, 8, 4), device='cuda')
    hip.stem_wgrad(xp, dconv, want)
    base = dev(rnd((64, 7, 8, 4), torch.float32, 70))
    got = base.clone()
    hip.stem_bwd_wgrad(xp, dev(w), mr, dy, y, idx, got, dy_pool2=dy2)
    assert hip.lib.eve_last_kernel().decode().startswith('stem_bwd_wgrad_kernel')
    got = (got - base)[:, :, :7, :3]
    assert torch.isfinite(got).all()
    rel = ((got - want[:, :, :7, :3]).norm() / want[:, :, :7, :3].norm()).item()
    assert rel < 2e-5 * max(1.0, (N / 16) ** 0.5), rel             # same products, float sums in another order
    x8 = hip.nchw_to_nhwc(dev(src), hdt, 8)
    ref = torch.nn.grad.conv2d_weight(x8.float().permute(0, 3, 1, 2)[:, :3], (64, 3, 7, 7), dconv.float().permute(0, 3, 1, 2),
                                      stride=2, padding=3).permute(0, 2, 3, 1)
    assert ((got - ref).norm() / ref.norm()).item() < 2e-3


@pytest.mark.parametrize('shape', [(1920, 512, 128), (37, 130, 128), (60, 128, 384), (1920, 128, 4), (5, 7, 3)],
                         ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('act', [0, 1, 3, 4], ids=['none', 'relu', 'selu', 'tanh'])
def test_small_linear_kernels(hip, ref, shape, act):
    """float32 nn.Linear of the EyeNet tail: forward, data gradient and weight/bias gradient."""
    M, K, N = shape
    x = rnd((M, K), torch.float32, 70)
    w = rnd((N, K), torch.float32, 71, scale=(1.0 / K) ** 0.5)          # [out, in]
    b = rnd((N,), torch.float32, 72)
    wt = w.t().contiguous()
    y_ref = ref.linear_fwd(x, wt, b, act)
    y = hip.linear_fwd(dev(x), dev(wt), dev(b), act)
    assert torch.allclose(y.cpu(), y_ref, rtol=2e-5, atol=2e-5)
    dy = rnd((M, N), torch.float32, 73)
    dx_ref = ref.linear_dgrad(dy, y_ref, act, w)
    dx = hip.linear_dgrad(dev(dy), y, act, dev(w))
    assert torch.allclose(dx.cpu(), dx_ref, rtol=1e-4, atol=2e-5)
    dw_ref, db_ref = torch.ones((N, K)), torch.ones((N,))
    ref.linear_wgrad(dy, y_ref, act, x, dw_ref, db_ref)
    dw, db = torch.ones((N, K), device='cuda'), torch.ones((N,), device='cuda')
    hip.linear_wgrad(dev(dy), y, act, dev(x), dw, db)
    assert torch.allclose(dw.cpu(), dw_ref, rtol=2e-4, atol=2e-4 * (M ** 0.5))
    assert torch.allclose(db.cpu(), db_ref, rtol=2e-4, atol=2e-4 * (M ** 0.5))


@HALVES
def test_stem_wgrad_from_packed_patches(hip, hdt):
    """conv1's weight gradient read from the 4-channel packed patches (7x8-tap view) == the generic kernel on NHWC8."""
    N = 5
    src = rnd((N, 3, 128, 128), torch.float32, 67)
    xp = hip.stem_pack_input(dev(src), dtype=hdt)
    x8 = hip.nchw_to_nhwc(dev(src), hdt, 8)
    dconv = dev(rnd((N, 64, 64, 64), hdt, 68))
    dw = torch.zeros((64, 7, 8, 4), device='cuda')
    hip.stem_wgrad(xp, dconv, dw)
    want = hip.conv2d_wgrad(x8, dconv, 7, 7, 2, 3, torch.zeros((64, 7, 7, 8), device='cuda'))
    got = dw[:, :, :7, :3]
    assert ((got - want[..., :3]).norm() / want[..., :3].norm()).item() < 2e-3
    ref = torch.nn.grad.conv2d_weight(x8.float().permute(0, 3, 1, 2)[:, :3], (64, 3, 7, 7), dconv.float().permute(0, 3, 1, 2),
                                      stride=2, padding=3).permute(0, 2, 3, 1)
    assert ((got - ref).norm() / ref.norm()).item() < 2e-3


def test_stem_wgrad_takes_images_in_chunks_below_2_gib():
    """256 x 256 patches (BASELINE configs[4]): d(conv1 out) of 1 030 frames is 2.01 GiB, beyond the transposing-read kernel's 32-bit
    buffer offsets -- eve_stem_wgrad takes the images in chunks (round 4; as one launch it fell to the first-generation kernel).
    The chunked call == the sum of two calls on the halves (each a single launch), and no first-generation kernel ran."""
    hdt = torch.float16
    N = 1030
    g = torch.Generator(device='cuda').manual_seed(5)
    xp = torch.zeros((N, 262, 264, 4), dtype=hdt, device='cuda')
    xp[:, 3:259, 4:260, :3] = torch.randn((N, 256, 256, 3), generator=g, device='cuda').to(hdt)
    dconv = (0.05 * torch.randn((N, 128, 128, 64), generator=g, device='cuda')).to(hdt)
    from eve_amd.kernels import default_kernels
    k = default_kernels()
    dw = torch.zeros((64, 7, 8, 4), device='cuda')
    k.stem_wgrad(xp, dconv, dw)
    assert 'wgrad_tr_kernel' in k.lib.eve_last_kernel().decode(), k.lib.eve_last_kernel().decode()
    ref = torch.zeros((64, 7, 8, 4), device='cuda')
    k.stem_wgrad(xp[:515], dconv[:515], ref)
    k.stem_wgrad(xp[515:], dconv[515:], ref)
    assert float((dw - ref).norm() / ref.norm()) < 1e-5
    # ... and a float reference on a few images of the tail of the batch
    sub = slice(N - 3, N)
    want = torch.nn.grad.conv2d_weight(xp[sub, 3:259, 4:260, :3].float().permute(0, 3, 1, 2), (64, 3, 7, 7),
                                       dconv[sub].float().permute(0, 3, 1, 2), stride=2, padding=3).permute(0, 2, 3, 1)
    got = torch.zeros((64, 7, 8, 4), device='cuda')
    k.stem_wgrad(xp[sub].contiguous(), dconv[sub].contiguous(), got)
    assert float((got[:, :, :7, :3] - want).norm() / want.norm()) < 2e-3


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
@pytest.mark.parametrize('case', [(2, 72, 128, 16, 32, 1), (2, 36, 64, 32, 64, 1), (2, 18, 32, 128, 64, 1), (3, 9, 16, 256, 128, 1),
                                  (2, 36, 64, 64, 64, 3), (2, 8, 8, 256, 256, 3)], ids=lambda c: 'x'.join(map(str, c)))
def test_conv_forward_accumulates_into_its_output(hip, ref, dtype, case):
    """EVE_EPI_ACCUMULATE: y += conv(x, w) + bias in the kernel epilogue (RefineNet's `layers(x) + skip_layer(x)`), for the
    1x1 skip convolutions of every level and a 3x3; against base + the same kernel's plain result (one extra rounding)."""
    N, H, W, Cin, Cout, K = case
    x = rnd((N, H, W, Cin), dtype, 51)
    w = rnd((Cout, K, K, Cin), dtype, 52, scale=0.1)
    b = rnd((Cout,), torch.float32, 53, scale=0.1)
    base = rnd((N, H, W, Cout), dtype, 54)
    plain = hip.conv2d_fwd(dev(x), dev(w), dev(b), 1, K // 2)
    got = hip.conv2d_fwd(dev(x), dev(w), dev(b), 1, K // 2, accumulate_into=dev(base).clone())
    want = ref.conv2d_fwd(x.float(), w.float(), b, 1, K // 2).float() + base.float()
    close(got, want.to(dtype), dtype, 'accumulating conv forward', scale=float(want.abs().max()))
    # and exactly plain + base up to the rounding of `plain`
    close(got, (plain.float().cpu() + base.float()).to(dtype), dtype, 'accumulating conv vs plain + base', scale=float(want.abs().max()))


STREAM_1X1_CASES = [(16, 32), (32, 16), (16, 64), (64, 16), (32, 64), (64, 32), (32, 128), (128, 32), (64, 128), (128, 64),
                    (16, 16), (32, 32), (64, 64)]


@pytest.mark.parametrize('hdt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
@pytest.mark.parametrize('cin,cout', STREAM_1X1_CASES, ids=lambda v: str(v))
def test_streaming_1x1_convolution(hip, ref, hdt, cin, cout):
    """Round 4: 1x1 / stride 1 convolutions between 16..128 channels on conv1x1_stream_kernel (csrc/conv_1x1.h: filter in MFMA A
    registers, pixels straight from global memory as B operands, no LDS): forward with bias (+ ReLU), forward accumulating into
    its output, data gradient and accumulating data gradient, against the float reference and against the gather kernel it
    replaces on the same inputs (conv1x1_stream = 0) -- a ragged pixel count (M = 3 x 77 x 71 = 16 401: not a multiple of the
    16-pixel tile nor of the 32 / 64-pixel batch), so the buffer bounds carry the tail."""
    N, H, W = 3, 77, 71
    x = rnd((N, H, W, cin), hdt, 61)
    w = rnd((cout, 1, 1, cin), hdt, 62, scale=(2.0 / cin) ** 0.5)
    b = rnd((cout,), torch.float32, 63, scale=0.2)
    base = rnd((N, H, W, cout), hdt, 64)
    dy = rnd((N, H, W, cout), hdt, 65)
    base_dx = rnd((N, H, W, cin), hdt, 66)
    w_ihwo = w.permute(3, 1, 2, 0).contiguous()

    def run():
        out = {}
        out['fwd'] = hip.conv2d_fwd(dev(x), dev(w), dev(b), 1, 0)
        used = hip.lib.eve_last_kernel().decode()
        out['fwd_relu'] = hip.conv2d_fwd(dev(x), dev(w), dev(b), 1, 0, 1)
        out['fwd_acc'] = hip.conv2d_fwd(dev(x), dev(w), dev(b), 1, 0, accumulate_into=dev(base).clone())
        out['dgrad'] = hip.conv2d_dgrad(dev(dy), dev(w_ihwo), (H, W), 1, 0)
        out['dgrad_acc'] = hip.conv2d_dgrad(dev(dy), dev(w_ihwo), (H, W), 1, 0, accumulate_into=dev(base_dx).clone())
        return out, used
    got, used = run()
    assert used.startswith('conv1x1_stream_kernel<') and (', %d, %d, false>' % (cin, cout)) in used, used
    with hip.dispatch_override(conv1x1_stream=0):
        old, used_old = run()
    assert 'conv1x1_stream' not in used_old
    want = ref.conv2d_fwd(x.float(), w.float(), b, 1, 0).float()
    wdx = ref.conv2d_dgrad(dy.float(), w_ihwo.float(), (H, W), 1, 0).float()
    wants = {'fwd': want, 'fwd_relu': want.clamp(min=0), 'fwd_acc': want + base.float(), 'dgrad': wdx, 'dgrad_acc': wdx + base_dx.float()}
    for name in wants:
        close(got[name], wants[name].to(hdt), hdt, 'streaming 1x1 ' + name, scale=float(wants[name].abs().max()))
        close(got[name], old[name], hdt, 'streaming 1x1 vs gather kernel: ' + name, scale=float(wants[name].abs().max()))
    # nothing is written past the last pixel: a guard plane behind the output stays as it was
    ybuf = torch.full((N * H * W + 64, cout), 7.0, dtype=hdt, device='cuda')
    view = ybuf[:N * H * W].view(N, H, W, cout)
    view.copy_(dev(base))
    hip.conv2d_fwd(dev(x), dev(w), dev(b), 1, 0, accumulate_into=view)
    assert bool((ybuf[N * H * W:] == 7.0).all())
    assert torch.equal(view, got['fwd_acc'])


@pytest.mark.parametrize('hdt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
@pytest.mark.parametrize('N,cout', [(131, 128), (75, 256)], ids=['131x128', '75x256'])
def test_eight_wave_convolution_on_half_image_bands(hip, ref, hdt, N, cout):
    """conv3x3_wg8_kernel<4, 2, 32, 9, 2> (csrc/conv_wg8.h, BANDS = 2): 32 x 32 planes of 128-channel layers (ResNet layer 2 on
    256 x 256 patches) as half-image bands of 16 rows -- the halo rows between the bands come from the other band, the outer ones
    are the image border; forward with bias (+ ReLU) and data gradient against the float reference and against the four-wave
    kernel on the same inputs (conv_wg8 = 0); an odd image count, one and two channel tiles."""
    x = rnd((N, 32, 32, 128), hdt, 95)
    w = rnd((cout, 3, 3, 128), hdt, 96, scale=(2.0 / 1152) ** 0.5)
    b = rnd((cout,), torch.float32, 97, scale=0.2)

    def run():
        out = {'fwd': hip.conv2d_fwd(dev(x), dev(w), dev(b), 1, 1)}
        used = hip.lib.eve_last_kernel().decode()
        out['fwd_relu'] = hip.conv2d_fwd(dev(x), dev(w), None, 1, 1, 1)
        if cout == 128:
            dy = rnd((N, 32, 32, 128), hdt, 98)
            out['dgrad'] = hip.conv2d_dgrad(dev(dy), dev(w.permute(3, 1, 2, 0).contiguous()), (32, 32), 1, 1)
            assert hip.lib.eve_last_kernel().decode() == used
        return out, used
    got, used = run()
    assert used.startswith('conv3x3_wg8_kernel<') and used.endswith(', 4, 2, 32, 9, 2>'), used
    with hip.dispatch_override(conv_wg8=0):
        old, used_old = run()
    assert 'wg8' not in used_old
    wants = {'fwd': ref.conv2d_fwd(x.float(), w.float(), b, 1, 1).float(),
             'fwd_relu': ref.conv2d_fwd(x.float(), w.float(), None, 1, 1).float().clamp(min=0)}
    if cout == 128:
        dy = rnd((N, 32, 32, 128), hdt, 98)
        wants['dgrad'] = ref.conv2d_dgrad(dy.float(), w.permute(3, 1, 2, 0).contiguous().float(), (32, 32), 1, 1).float()
    for name in wants:
        close(got[name], wants[name].to(hdt), hdt, 'banded eight-wave 3x3 ' + name, scale=float(wants[name].abs().max()))
        close(got[name], old[name], hdt, 'banded eight-wave 3x3 vs the four-wave kernel: ' + name, scale=float(wants[name].abs().max()))


@pytest.mark.parametrize('hdt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
@pytest.mark.parametrize('N,W', [(1, 32), (37, 64), (300, 32), (1, 64), (67, 64)], ids=lambda v: str(v))
def test_filter_resident_streaming_convolution_of_layer_one(hip, ref, hdt, N, W):
    """conv3x3_ws64_kernel (csrc/conv_ws64.h: the 64 -> 64 filter bank resident in LDS, half-image tiles of 32 x 32 planes or
    8-row bands of 64 x 64 planes streaming under it): forward with bias (+ ReLU) and data gradient against the float reference
    and against the kernel it replaces on the same inputs (conv_ws64 = 0); one image (fewer tiles than workgroups, a stream of
    one tile), 37 / 67 images of 64 x 64 (296 / 536 tiles on 256 workgroups: streams of one, two and three tiles, both
    accumulator sets, the first and the last band's missing halo rows), borders included."""
    x = rnd((N, W, W, 64), hdt, 91)
    w = rnd((64, 3, 3, 64), hdt, 92, scale=(2.0 / 576) ** 0.5)
    b = rnd((64,), torch.float32, 93, scale=0.2)
    dy = rnd((N, W, W, 64), hdt, 94)
    w_ihwo = w.permute(3, 1, 2, 0).contiguous()

    def run():
        out = {}
        out['fwd'] = hip.conv2d_fwd(dev(x), dev(w), dev(b), 1, 1)
        used = hip.lib.eve_last_kernel().decode()
        out['fwd_relu'] = hip.conv2d_fwd(dev(x), dev(w), None, 1, 1, 1)
        out['dgrad'] = hip.conv2d_dgrad(dev(dy), dev(w_ihwo), (W, W), 1, 1)
        return out, used, hip.lib.eve_last_kernel().decode()
    got, used, used_d = run()
    for u in (used, used_d):
        assert u.startswith('conv3x3_ws64_kernel<') and u.endswith(', %d>' % W), u
    with hip.dispatch_override(conv_ws64=0):
        old, used_old, _ = run()
    assert 'ws64' not in used_old
    want = ref.conv2d_fwd(x.float(), w.float(), b, 1, 1).float()
    wants = {'fwd': want, 'fwd_relu': ref.conv2d_fwd(x.float(), w.float(), None, 1, 1).float().clamp(min=0),
             'dgrad': ref.conv2d_dgrad(dy.float(), w_ihwo.float(), (W, W), 1, 1).float()}
    for name in wants:
        close(got[name], wants[name].to(hdt), hdt, 'filter-resident 3x3 ' + name, scale=float(wants[name].abs().max()))
        close(got[name], old[name], hdt, 'filter-resident 3x3 vs the halo kernel: ' + name, scale=float(wants[name].abs().max()))
    # the same launch twice: no accumulator or stage state leaks between streams
    assert torch.equal(hip.conv2d_fwd(dev(x), dev(w), dev(b), 1, 1), got['fwd'])


@pytest.mark.parametrize('hdt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
@pytest.mark.parametrize('geom', [(8, 72, 128), (29, 37, 64)], ids=['72x128', '37x64'])
@pytest.mark.parametrize('cin,cout', [(16, 16), (16, 32), (32, 16), (32, 32), (16, 64), (64, 16), (32, 64), (32, 128)],
                         ids=lambda v: str(v))
def test_row_streaming_3x3_convolution(hip, ref, hdt, geom, cin, cout):
    """Round 4: 3x3 / stride 1 / pad 1 between 16..64 channels on 64 / 128-wide images on conv3x3_stream_kernel (csrc/conv_3x3s.h:
    one workgroup walks an image with a four-row LDS ring, exact channel counts instead of pixel groups, filter in MFMA A
    registers): forward with bias (+ ReLU), accumulating forward, data gradient (mirrored taps on the transposed filter) and
    accumulating data gradient -- against the float reference and against the kernels it replaces on the same inputs
    (conv3x3_stream = 0); an odd image height, borders included."""
    N, H, W = geom
    x = rnd((N, H, W, cin), hdt, 81)
    w = rnd((cout, 3, 3, cin), hdt, 82, scale=(2.0 / (9 * cin)) ** 0.5)
    b = rnd((cout,), torch.float32, 83, scale=0.2)
    base = rnd((N, H, W, cout), hdt, 84)
    dy = rnd((N, H, W, cout), hdt, 85)
    base_dx = rnd((N, H, W, cin), hdt, 86)
    w_ihwo = w.permute(3, 1, 2, 0).contiguous()

    def run():
        out = {}
        out['fwd'] = hip.conv2d_fwd(dev(x), dev(w), dev(b), 1, 1)
        used = hip.lib.eve_last_kernel().decode()
        out['fwd_relu'] = hip.conv2d_fwd(dev(x), dev(w), dev(b), 1, 1, 1)
        out['fwd_acc'] = hip.conv2d_fwd(dev(x), dev(w), dev(b), 1, 1, accumulate_into=dev(base).clone())
        out['dgrad'] = hip.conv2d_dgrad(dev(dy), dev(w_ihwo), (H, W), 1, 1)
        used_d = hip.lib.eve_last_kernel().decode()
        out['dgrad_acc'] = hip.conv2d_dgrad(dev(dy), dev(w_ihwo), (H, W), 1, 1, accumulate_into=dev(base_dx).clone())
        return out, used, used_d
    got, used, used_d = run()
    assert used.startswith('conv3x3_stream_kernel<') and (', %d, %d, false>' % (cin, cout)) in used, used
    from eve_amd.ops import STREAM_3X3
    if (cout, cin) in STREAM_3X3:          # (the data gradient of 32 -> 64 / 128 is a 64 / 128 -> 32 layer: not on this kernel)
        assert used_d.startswith('conv3x3_stream_kernel<') and (', %d, %d, false>' % (cout, cin)) in used_d, used_d
    with hip.dispatch_override(conv3x3_stream=0):
        old, used_old, _ = run()
    assert 'conv3x3_stream' not in used_old
    want = ref.conv2d_fwd(x.float(), w.float(), b, 1, 1).float()
    wdx = ref.conv2d_dgrad(dy.float(), w_ihwo.float(), (H, W), 1, 1).float()
    wants = {'fwd': want, 'fwd_relu': want.clamp(min=0), 'fwd_acc': want + base.float(), 'dgrad': wdx, 'dgrad_acc': wdx + base_dx.float()}
    for name in wants:
        close(got[name], wants[name].to(hdt), hdt, 'row-streaming 3x3 ' + name, scale=float(wants[name].abs().max()))
        close(got[name], old[name], hdt, 'row-streaming 3x3 vs the kernel it replaces: ' + name, scale=float(wants[name].abs().max()))


@pytest.mark.parametrize('ks,cin,cout', [(3, 8, 16), (1, 16, 8)], ids=['initial.0', 'final.2'])
def test_eight_channel_weight_gradient_over_pixel_pairs(hip, ref, ks, cin, cout):
    """RefineNet's first / last convolution (4 -> 16 and 16 -> 1, channels padded to 8) at 72 x 128: ops._wgrad_pixel_pairs forms
    the weight (+ bias) gradient of the pixel-paired layer on the band-resident kernel and scatter-sums it back -- against the
    gather kernel on the unpaired tensors and the float reference."""
    from eve_amd import ops
    dt = torch.bfloat16
    N, H, W = 232, 72, 128                      # 2.1 M pixels: above the band-resident kernel's floor after pairing
    g = torch.Generator(device='cuda').manual_seed(9)
    x = torch.randn((N, H, W, cin), generator=g, device='cuda').to(dt)
    dy = (0.1 * torch.randn((N, H, W, cout), generator=g, device='cuda')).to(dt)
    dwp = torch.zeros((cout, ks, ks, cin), device='cuda')
    db = torch.zeros((cout,), device='cuda')
    assert ops._wgrad_pixel_pairs(hip, x, dy, ks, 1, ks // 2, dwp, db)
    assert 'wgrad_halo_kernel' in hip.lib.eve_last_kernel().decode() or 'index' in hip.lib.eve_last_kernel().decode()
    old_dw = torch.zeros_like(dwp)
    old_db = torch.zeros_like(db)
    hip.conv2d_wgrad(x, dy, ks, ks, 1, ks // 2, old_dw, db=old_db)
    assert float((dwp - old_dw).norm() / old_dw.norm()) < 2e-4
    assert float((db - old_db).norm() / old_db.norm()) < 2e-4
    sub = slice(0, 3)
    want = torch.nn.grad.conv2d_weight(x[sub].float().permute(0, 3, 1, 2), (cout, cin, ks, ks), dy[sub].float().permute(0, 3, 1, 2),
                                       stride=1, padding=ks // 2).permute(0, 2, 3, 1)
    got = torch.zeros((cout, ks, ks, cin), device='cuda')
    hip.conv2d_wgrad(x[sub].contiguous(), dy[sub].contiguous(), ks, ks, 1, ks // 2, got)
    assert float((got - want).norm() / want.norm()) < 3e-3


def test_narrow_output_3x3_runs_pixel_paired(hip, ref):
    """The outermost decoder's first 3x3 (64 -> 16 channels at 72x128): ops.PackedWeight pairs pixels (128 -> 32 over a 64-wide row,
    ops.PAIR_NARROW_OUT) so that the halo kernel takes it instead of the first-generation gather kernel; through ops.conv2d
    against the float reference, forward and both gradients."""
    from eve_amd import ops
    dt = torch.bfloat16
    N, H, W, cin, cout = 2, 72, 128, 64, 16
    x = dev(rnd((N, H, W, cin), dt, 71)).requires_grad_(True)
    wt = torch.nn.Parameter(dev(rnd((cout, cin, 3, 3), torch.float32, 72, scale=(2.0 / (9 * cin)) ** 0.5)))
    bias = torch.nn.Parameter(dev(rnd((cout,), torch.float32, 73, scale=0.1)))
    pack = ops.PackedWeight(wt, dt)
    assert pack.pair_fwd is not None and pack.pair_fwd[0] == 2 and tuple(pack.pair_fwd[1].shape) == (32, 3, 3, 128)
    y = ops.conv2d(x, wt, bias, pack, stride=1, pad=1)
    used = hip.lib.eve_last_kernel().decode()
    assert 'halo' in used, used
    dy = dev(rnd((N, H, W, cout), dt, 74))
    y.backward(dy)
    w_ohwi = wt.detach().permute(0, 2, 3, 1).to(dt)
    want = ref.conv2d_fwd(x.detach().cpu(), w_ohwi.cpu(), bias.detach().cpu(), 1, 1)
    close(y, want, dt, 'paired 64 -> 16 forward')
    want_dx = ref.conv2d_dgrad(dy.cpu(), w_ohwi.permute(3, 1, 2, 0).contiguous().cpu(), (H, W), 1, 1)
    close(x.grad, want_dx, dt, 'paired 64 -> 16 data gradient')
    want_dw = ref.conv2d_wgrad(x.detach().cpu(), dy.cpu(), 3, 3, 1, 1, torch.zeros((cout, 3, 3, cin)))
    close(wt.grad.permute(0, 2, 3, 1), want_dw, dt, 'paired 64 -> 16 weight gradient')


def test_conv_rejects_bad_shapes(hip):
    x = torch.zeros((1, 8, 8, 6), device='cuda')
    w = torch.zeros((8, 3, 3, 6), device='cuda')
    with pytest.raises(RuntimeError, match='multiple of the 16-byte vector'):
        hip.conv2d_fwd(x, w, None, 1, 1)
    with pytest.raises(RuntimeError, match='not on the GPU'):
        hip.conv2d_fwd(torch.zeros((1, 8, 8, 8)), torch.zeros((8, 3, 3, 8)), None, 1, 1)


PLANE_CASES = [(3, 32, 32, 64), (2, 4, 4, 512), (2, 5, 8, 64), (2, 72, 128, 16), (2, 9, 16, 256), (1, 7, 5, 8),
               (2, 16, 16, 128), (3, 8, 8, 256), (2, 18, 32, 64), (2, 36, 64, 32), (2, 64, 64, 64),
               (11, 32, 32, 64),         # (planes of 4 097 .. 8 192 vectors are dealt to two workgroups, 8 blocks apart)
               (9, 64, 64, 64), (3, 32, 32, 128)]    # the trunk on 256x256 patches: four / two parts of 8 192 vectors


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
@pytest.mark.parametrize('shape', PLANE_CASES, ids=lambda s: 'x'.join(map(str, s)))
def test_instnorm_fwd_bwd(hip, ref, dtype, shape):
    N, H, W, C = shape
    x = (rnd(shape, torch.float32, 13) * 1.7 + 0.4 * rnd((N, 1, 1, C), torch.float32, 14)).to(dtype)
    mr_w = ref.instnorm_stats(x)
    mr_g = hip.instnorm_stats(dev(x))
    close(mr_g[..., 0], mr_w[..., 0], torch.float32, 'mean', scale=1.0)
    close(mr_g[..., 1], mr_w[..., 1], torch.float32, 'rstd', scale=float(mr_w[..., 1].max()) * 3)
    gamma = 1 + 0.2 * rnd((C,), torch.float32, 15)
    beta = 0.1 * rnd((C,), torch.float32, 16)
    res = rnd(shape, dtype, 17)
    dy = rnd(shape, dtype, 18)
    for (g, b, r, act) in ((None, None, None, 1), (None, None, res, 1), (None, None, None, 0),
                           (gamma, beta, None, 1), (gamma, beta, None, 2)):
        y_w = ref.instnorm_act_fwd(x, mr_w, g, b, r, act)
        y_g = hip.instnorm_act_fwd(dev(x), dev(mr_w), dev(g), dev(b), dev(r), act)
        close(y_g, y_w, dtype, 'instnorm_act fwd act=%d' % act)
        dx_w, dres_w, s_w = ref.instnorm_act_bwd(dy, y_w, x, mr_w, g, act, r is not None)
        dx_g, dres_g, s_g = hip.instnorm_act_bwd(dev(dy), dev(y_w), dev(x), dev(mr_w), dev(g), act, r is not None)
        close(dx_g, dx_w, dtype, 'instnorm_act bwd dx act=%d' % act, scale=float(dx_w.abs().max()) + 0.05)
        close(s_g, s_w, torch.float32, 'instnorm_act bwd sums', scale=float(s_w.abs().max()) * 4)
        if r is None:
            sk2 = dev(rnd(shape, dtype, 19))
            da, _, sa = hip.instnorm_act_bwd(dev(dy), None if g is not None else dev(y_w), dev(x), dev(mr_w), dev(g), act, False, beta=dev(b), dx_add=sk2)
            dn, _, sn = hip.instnorm_act_bwd(dev(dy), None if g is not None else dev(y_w), dev(x), dev(mr_w), dev(g), act, False, beta=dev(b))
            assert torch.equal(da, hip.add(dn, sk2)) and torch.equal(sa, sn)
        if r is not None:
            close(dres_g, dres_w, dtype, 'instnorm_act bwd dres')
        fused = hip.instnorm_fwd_fused(dev(x), dev(g), dev(b), dev(r), act)
        nvec = H * W * C // (4 if dtype == torch.float32 else 8)
        # 8 vectors per thread x 1 024 threads; beyond that 16-bit planes WITHOUT affine parameters are dealt by channels to several
        # workgroups (round 4, in_big_planes: down to one 16-byte vector per pixel and <= 9 216 vectors per part)
        cv = C // (4 if dtype == torch.float32 else 8)
        part = nvec
        while part > 8192 and cv >= 2 and dtype != torch.float32 and g is None:
            part, cv = part // 2, cv // 2
        assert (fused is None) == (part > 8192 and not (part <= 9216 and dtype != torch.float32 and g is None))
        if fused is not None:
            close(fused[0], y_w, dtype, 'fused instnorm fwd act=%d' % act)
            close(fused[1][..., 0], mr_w[..., 0], torch.float32, 'fused mean', scale=1.0)
            close(fused[1][..., 1], mr_w[..., 1], torch.float32, 'fused rstd', scale=float(mr_w[..., 1].max()) * 3)
            fb = hip.instnorm_bwd_fused(dev(dy), dev(y_w), dev(x), dev(mr_w), dev(g), act, r is not None)
            close(fb[0], dx_w, dtype, 'fused instnorm bwd dx act=%d' % act, scale=float(dx_w.abs().max()) + 0.05)
            close(fb[2], s_w, torch.float32, 'fused instnorm bwd sums', scale=float(s_w.abs().max()) * 4)
            if r is not None:
                close(fb[1], dres_w, dtype, 'fused instnorm bwd dres')
            if g is not None and r is None:      # the fork's other gradient added in the epilogue (ops.InstNormActSkipFn): == a separate add
                sk = dev(rnd(shape, dtype, 19))
                fa = hip.instnorm_bwd_fused(dev(dy), None, dev(x), fused[1], dev(g), act, False, beta=dev(b), dx_add=sk)
                fn = hip.instnorm_bwd_fused(dev(dy), None, dev(x), fused[1], dev(g), act, False, beta=dev(b))
                assert torch.equal(fa[0], hip.add(fn[0], sk)) and torch.equal(fa[2], fn[2])
            if g is not None and r is None:      # affine, no residual: act' recomputed from x with the forward's scale / shift (round 4)
                fr = hip.instnorm_bwd_fused(dev(dy), None, dev(x), fused[1], dev(g), act, False, beta=dev(b))
                fyy = hip.instnorm_bwd_fused(dev(dy), fused[0], dev(x), fused[1], dev(g), act, False)
                if dtype != torch.float16:      # (float16 can round a tiny positive pre-activation to 0; bf16 / float32 cannot)
                    assert torch.equal(fr[0], fyy[0]) and torch.equal(fr[2], fyy[2])
                close(fr[0], dx_w, dtype, 'fused instnorm bwd from x act=%d' % act, scale=float(dx_w.abs().max()) + 0.05)
            if act == 1:       # ReLU: the forward's sign mask (one byte per 16-byte vector) replaces y in the backward
                y_m, _, mask = hip.instnorm_fwd_fused(dev(x), dev(g), dev(b), dev(r), act, want_mask=True)
                assert torch.equal(y_m, fused[0])
                vec = 16 // x.element_size()
                want_bits = ((y_m.cpu().reshape(-1, vec) > 0).to(torch.int32) << torch.arange(vec, dtype=torch.int32)).sum(1)
                assert torch.equal(mask.cpu().to(torch.int32), want_bits)
                fm = hip.instnorm_bwd_fused(dev(dy), None, dev(x), dev(mr_w), dev(g), act, r is not None, mask=mask)
                fy = hip.instnorm_bwd_fused(dev(dy), y_m, dev(x), dev(mr_w), dev(g), act, r is not None)
                assert torch.equal(fm[0], fy[0]) and torch.equal(fm[2], fy[2])          # same arithmetic, bit for bit
                if r is not None:
                    assert torch.equal(fm[1], fy[1])
            if r is None and g is None and act != 0:       # act' recomputed from x instead of reading y
                fb2 = hip.instnorm_bwd_fused(dev(dy), None, dev(x), dev(mr_w), None, act, False)
                close(fb2[0], dx_w, dtype, 'fused instnorm bwd dx without y', scale=float(dx_w.abs().max()) + 0.05)
        if r is None and act != 0:                         # act' recomputed from x (with the affine scale / shift)
            dx2, _, s2 = hip.instnorm_act_bwd(dev(dy), None, dev(x), dev(mr_w), dev(g), act, False, beta=dev(b))
            close(dx2, dx_w, dtype, 'instnorm_act bwd dx without y', scale=float(dx_w.abs().max()) + 0.05)
            close(s2, s_w, torch.float32, 'instnorm_act bwd sums without y', scale=float(s_w.abs().max()) * 4)


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
@pytest.mark.parametrize('shape,c2', [((2, 72, 128, 32), 32), ((11, 36, 64, 64), 64), ((3, 18, 32, 128), 64), ((2, 72, 128, 16), 0),
                                      ((9, 7, 5, 8), 16)], ids=['72x128x32+32', '11x36x64x64+64', '18x32x128+64', '72x128x16', '9x7x5x8+16'])
def test_instnorm_two_heads_over_concatenated_sources(hip, ref, dtype, shape, c2):
    """eve_instnorm_act2_{fwd,bwd}: two affine heads (RefineNet `layers.0` / `skip_layer.0`) over the channel-concatenation
    of one or two sources, each source normalised on its own into its channel range (one launch; the sources of an image run
    8 workgroups apart).  Forward: every head equals the single-head kernel on the materialised concatenation's slices bit
    for bit.  Backward: the fork's summed gradient against the float32 restatement (sum formed before rounding), per-head
    sums against it too; one head only equals the single-head kernel."""
    N, H, W, C1 = shape
    ctot = C1 + c2
    srcs = [(rnd(shape, torch.float32, 31) * 1.3 + 0.3 * rnd((N, 1, 1, C1), torch.float32, 32)).to(dtype)]
    if c2:
        srcs.append((rnd((N, H, W, c2), torch.float32, 33) * 0.8 - 0.2).to(dtype))
    ga, ba = 1 + 0.2 * rnd((ctot,), torch.float32, 34), 0.1 * rnd((ctot,), torch.float32, 35)
    gb, bb = 1 - 0.3 * rnd((ctot,), torch.float32, 36), 0.2 * rnd((ctot,), torch.float32, 37)
    d_a, d_b = rnd((N, H, W, ctot), dtype, 38), rnd((N, H, W, ctot), dtype, 39)
    dsrcs = [dev(x) for x in srcs]
    mrs = [hip.instnorm_stats(x) for x in dsrcs]
    for act in (1, 2):
        y_a, y_b = hip.instnorm_act2_fwd(dsrcs, mrs, dev(ga), dev(ba), dev(gb), dev(bb), act)
        off = 0
        for x, mr in zip(dsrcs, mrs):
            sl = slice(off, off + x.shape[-1])
            g = lambda t: dev(t[sl].contiguous())
            assert torch.equal(y_a[..., sl], hip.instnorm_act_fwd(x, mr, g(ga), g(ba), None, act))
            assert torch.equal(y_b[..., sl], hip.instnorm_act_fwd(x, mr, g(gb), g(bb), None, act))
            off += x.shape[-1]
        y1, none = hip.instnorm_act2_fwd(dsrcs, mrs, dev(ga), dev(ba), None, None, act)
        assert none is None and torch.equal(y1, y_a)
        dxs, s_a, s_b = hip.instnorm_act2_bwd(dev(d_a), dev(d_b), dsrcs, mrs, dev(ga), dev(ba), dev(gb), dev(bb), act)
        want_dx, want_a, want_b = ref.instnorm_act2_bwd(d_a.float(), d_b.float(), [x.float() for x in srcs], [m.cpu() for m in mrs],
                                                        ga, ba, gb, bb, act)
        for dx, want in zip(dxs, want_dx):
            close(dx, want.to(dtype), dtype, 'two-head instnorm bwd dx act=%d' % act, scale=float(want.abs().max()) + 0.05)
        close(s_a, want_a, torch.float32, 'two-head sums a', scale=float(want_a.abs().max()) * 4)
        close(s_b, want_b, torch.float32, 'two-head sums b', scale=float(want_b.abs().max()) * 4)
        # one head only (dy_b = None): the single-head kernel's gradient of each source
        dx1, s1, none = hip.instnorm_act2_bwd(dev(d_a), None, dsrcs, mrs, dev(ga), dev(ba), None, None, act)
        assert none is None
        off = 0
        for x, mr, got in zip(dsrcs, mrs, dx1):
            sl = slice(off, off + x.shape[-1])
            g = lambda t: dev(t[sl].contiguous())
            dx_single, _, s_single = hip.instnorm_act_bwd(dev(d_a[..., sl].contiguous()), None, x, mr, g(ga), act, False, beta=g(ba))
            close(got, dx_single.cpu(), dtype, 'one-head bwd', scale=float(dx_single.abs().max()) + 0.05)
            close(s1[:, sl], s_single.cpu(), torch.float32, 'one-head sums', scale=float(s_single.abs().max()) * 4)
            off += x.shape[-1]


@pytest.mark.parametrize('shape', [(960, 32, 2), (7, 1024, 2), (1, 8, 2), (1920, 130)], ids=lambda s: 'x'.join(map(str, s)))
def test_sum_rows_is_the_fixed_order_batch_reduction(hip, shape):
    t = rnd(shape, torch.float32, 61)
    got = hip.sum_rows(dev(t))
    want = t.double().sum(dim=0)
    assert got.shape == want.shape
    assert float((got.cpu().double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    assert torch.equal(got, hip.sum_rows(dev(t)))          # same order every time


def test_channel_split_instnorm_equals_the_single_workgroup_kernel(tmp_path):
    """Planes of 4 097 .. 8 192 vectors run as two 512-thread workgroups, half the channels each (EVE_IN_SPLIT=0: one
    1 024-thread workgroup).  Same arithmetic per channel, only the order of the plane reductions differs: forward with
    residual + sign mask, backward with the two-summand gradient, the mask or y, dres and the per-plane sums -- masks and
    dres bit-equal, everything else within one rounding of the reduction order.  (The switch is read once per process.)"""
    import os
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from eve_amd.kernels import HipKernels
k = HipKernels()
g = torch.Generator().manual_seed(5)
outs = []
for dt in (torch.float32, torch.bfloat16):
    for (N, H, W, C) in ((3, 16, 16, 128), (11, 32, 32, 64), (2, 18, 32, 64)):
        x, r, dy, dy2 = (torch.randn((N, H, W, C), generator=g).to(dt).cuda() for _ in range(4))
        f = k.instnorm_fwd_fused(x, None, None, r, 1, want_mask=True)
        if f is None:
            outs.append(None)
            continue
        y, mr, mask = f
        b = k.instnorm_bwd_fused(dy, None, x, mr, None, 1, True, dy2=dy2, mask=mask)
        b2 = k.instnorm_bwd_fused(dy, y, x, mr, None, 1, True, dy2=dy2)
        outs.append([t.float().cpu() for t in (y, mr, mask, b[0], b[1], b[2], b2[0], b2[1])])
torch.save(outs, sys.argv[1])
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ('1', '0'):
        path = os.path.join(str(tmp_path), 'in%s.pt' % mode)
        p = subprocess.run([sys.executable, '-c', code, path], env=dict(os.environ, EVE_IN_SPLIT=mode), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:]
        res[mode] = torch.load(path)
    compared = 0
    for a, b in zip(res['1'], res['0']):
        assert (a is None) == (b is None)
        if a is None:
            continue
        compared += 1
        y1, mr1, m1, dx1, dres1, s1, dxy1, dresy1 = a
        y0, mr0, m0, dx0, dres0, s0, dxy0, dresy0 = b
        assert float((mr1 - mr0).abs().max()) <= 1e-6 * float(mr0.abs().max())
        ulp = 2.0 ** -7                                      # bf16 outputs may move by one rounding where a statistic did
        assert float((y1 - y0).abs().max()) <= ulp * float(y0.abs().max())
        assert float((m1 != m0).float().mean()) <= 1e-4      # a sign can only flip where y is within a rounding of 0
        assert torch.equal(dres1, dres0) or float((dres1 != dres0).float().mean()) <= 1e-4
        assert torch.equal(dx1, dxy1) and torch.equal(dres1, dresy1)          # mask or y: the same backward
        assert float((dx1 - dx0).abs().max()) <= ulp * float(dx0.abs().max())
        assert float((s1 - s0).abs().max()) <= 1e-5 * float(s0.abs().max())
    assert compared >= 4


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
def test_elementwise(hip, ref, dtype):
    for n in (8 * 1000, 8 * 1000 + 3):
        a, b = rnd((n,), dtype, 19), rnd((n,), dtype, 20)
        close(hip.add(dev(a), dev(b)), ref.add(a, b), dtype, 'add')
        for act in range(6):
            y = torch.tanh(a.float()).to(dtype) if act in (4, 5) else a
            close(hip.act_bwd(dev(b), dev(y), act), ref.act_bwd(b, y, act), dtype, 'act_bwd %d' % act)


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
def test_pooling_and_resize(hip, ref, dtype):
    x = torch.relu(rnd((2, 64, 64, 64), dtype, 21))          # post-ReLU: many exact ties at 0
    y_w, idx_w = ref.maxpool3x3s2_fwd(x)
    y_g, idx_g = hip.maxpool3x3s2_fwd(dev(x))
    close(y_g, y_w, dtype, 'maxpool fwd')
    dy = rnd(tuple(y_w.shape), dtype, 22)
    close(hip.maxpool3x3s2_bwd(dev(dy), idx_g, (64, 64)), ref.maxpool3x3s2_bwd(dy, idx_w, (64, 64)),
          dtype, 'maxpool bwd')
    # stem tail fused: IN -> ReLU -> max-pool against the three separate reference steps.  The reference runs in
    # float32 on the same (dtype-rounded) input: rounding the normalised tensor to bf16 first would create
    # window ties whose arg-max choice (and gradient routing) is arbitrary.
    xs = (rnd((2, 64, 64, 64), torch.float32, 50) * 1.3 + 0.2).to(dtype)
    mr = ref.instnorm_stats(xs)
    yp_w, idx_w = ref.in_relu_maxpool_fwd(xs.float(), mr)
    yp_g, idx_g = hip.in_relu_maxpool_fwd(dev(xs), dev(mr))
    close(yp_g, yp_w, dtype, 'in_relu_maxpool fwd')
    dyp = rnd(tuple(yp_w.shape), dtype, 51)
    close(hip.in_relu_maxpool_bwd(dev(dyp), yp_g, idx_g, dev(xs), dev(mr)),
          ref.in_relu_maxpool_bwd(dyp.float(), yp_w, idx_w, xs.float(), mr), dtype, 'in_relu_maxpool bwd')
    x = rnd((3, 4, 4, 512), dtype, 23)
    close(hip.avgpool_fwd(dev(x)), ref.avgpool_fwd(x), dtype, 'avgpool fwd')
    dy = rnd((3, 512), dtype, 24)
    close(hip.avgpool_bwd(dev(dy), (4, 4)), ref.avgpool_bwd(dy, (4, 4)), dtype, 'avgpool bwd')
    for (ih, iw, oh, ow, c) in ((72, 128, 36, 64, 32), (9, 16, 5, 8, 256), (18, 32, 9, 16, 128)):
        x = rnd((2, ih, iw, c), dtype, 25)
        y_w, idx_w = ref.adaptive_maxpool_fwd(x, (oh, ow))
        y_g, idx_g = hip.adaptive_maxpool_fwd(dev(x), (oh, ow))
        close(y_g, y_w, dtype, 'adaptive maxpool fwd')
        assert (idx_g.cpu().long() == idx_w).all(), 'adaptive maxpool indices'
        dy = rnd(tuple(y_w.shape), dtype, 26)
        close(hip.adaptive_maxpool_bwd(dev(dy), idx_g, (ih, iw)), ref.adaptive_maxpool_bwd(dy, idx_w, (ih, iw)),
              dtype, 'adaptive maxpool bwd')
        # ... with a second gradient of the pool's input summed in the epilogue (ops.PoolForkFn): exactly the separate add
        skipg = rnd((2, ih, iw, c), dtype, 30)
        fused = hip.adaptive_maxpool_bwd(dev(dy), idx_g, (ih, iw), add=dev(skipg))
        assert torch.equal(fused, hip.add(hip.adaptive_maxpool_bwd(dev(dy), idx_g, (ih, iw)), dev(skipg)))
        xs = rnd((2, oh, ow, c), dtype, 27)
        close(hip.bilinear_fwd(dev(xs), (ih, iw)), ref.bilinear_fwd(xs, (ih, iw)), dtype, 'bilinear fwd')
        dyu = rnd((2, ih, iw, c), dtype, 28)
        close(hip.bilinear_bwd(dev(dyu), (oh, ow)), ref.bilinear_bwd(dyu, (oh, ow)), dtype, 'bilinear bwd')


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
def test_layout_and_pack(hip, ref, dtype):
    src = rnd((3, 3, 16, 24), torch.float32, 29)
    cpad = 4 if dtype == torch.float32 else 8
    got = hip.nchw_to_nhwc(dev(src), dtype, cpad)
    close(got, ref.nchw_to_nhwc(src, dtype, cpad), dtype, 'nchw->nhwc')
    close(hip.nhwc_to_nchw(got, 3), ref.nhwc_to_nchw(got.cpu(), 3), dtype, 'nhwc->nchw')
    w = rnd((16, 3, 3, 8), torch.float32, 30)
    a_g, b_g = hip.pack_weights(dev(w), dtype)
    a_w, b_w = ref.pack_weights(w, dtype)
    close(a_g, a_w, dtype, 'pack ohwi')
    close(b_g, b_w, dtype, 'pack ihwo')
    close(hip.cast(dev(w), dtype), w.to(dtype), dtype, 'cast')
    # the batched (one launch, tile-transposing) variant on ragged shapes, with and without the IHWO copy
    ws = [rnd(sh, torch.float32, 90 + i) for i, sh in enumerate([(16, 3, 3, 8), (40, 1, 1, 72), (64, 7, 7, 8), (4, 1, 1, 128), (130, 3, 3, 33)])]
    want_ihwo = [True, True, False, True, True]
    res = hip.pack_weights_batch([dev(w_) for w_ in ws], dtype, want_ihwo)
    for w_, wi, (a, b) in zip(ws, want_ihwo, res):
        assert torch.equal(a.cpu(), w_.to(dtype))
        assert (b is None) == (not wi)
        if wi:
            assert torch.equal(b.cpu(), w_.permute(3, 1, 2, 0).contiguous().to(dtype))

    # ABI v10: a narrower source, the packed copies' extra output / input channels written as zeros by the same launch
    padded = [(16, 8), (40, 72), (64, 8), (4, 128), (132, 36)]
    srcs = [w_[:co - (2 if i in (0, 4) else 0), :, :, :ci - (3 if i in (0, 4) else 0)].contiguous()
            for i, (w_, (co, ci)) in enumerate(zip(ws, [(16, 8), (40, 72), (64, 8), (4, 128), (130, 33)]))]
    res = hip.pack_weights_batch([dev(w_) for w_ in srcs], dtype, want_ihwo, padded=padded)
    for w_, wi, (co, ci), (a, b) in zip(srcs, want_ihwo, padded, res):
        want = torch.nn.functional.pad(w_, (0, ci - w_.shape[3], 0, 0, 0, 0, 0, co - w_.shape[0])).to(dtype)
        assert tuple(a.shape) == (co, w_.shape[1], w_.shape[2], ci) and torch.equal(a.cpu(), want)
        if wi:
            assert torch.equal(b.cpu(), want.permute(3, 1, 2, 0).contiguous())


@HALVES
def test_avgpool_with_a_float32_pooled_side_equals_pool_plus_cast(hip, hdt):
    """eve_avgpool_{fwd,bwd}_f32 (ABI v10: the trunk -> tail hand-over in one launch each way) == pool in the 16-bit format + cast,
    bit for bit; float32 input: the plain kernels."""
    x = rnd((7, 4, 4, 512), torch.float32, 310)
    xh = dev(x).to(hdt)
    y = hip.avgpool_fwd_f32(xh)
    assert y.dtype == torch.float32 and torch.equal(y, hip.avgpool_fwd(xh).float())
    dy = dev(rnd((7, 512), torch.float32, 311))
    dx = hip.avgpool_bwd_f32(dy, (4, 4), hdt)
    assert dx.dtype == hdt and torch.equal(dx, hip.avgpool_bwd(dy.to(hdt), (4, 4)))
    xf = dev(x)
    assert torch.equal(hip.avgpool_fwd_f32(xf), hip.avgpool_fwd(xf))
    assert torch.equal(hip.avgpool_bwd_f32(dy, (4, 4), torch.float32), hip.avgpool_bwd(dy, (4, 4)))


def test_strided_accumulating_linear_kernels(hip, ref):
    """eve_linear_{fwd,dgrad}_ex: a column range of a wider output, a short bias, the data gradient added onto an existing one, and
    the act'(y) prologue read with the wide row stride -- what ops.EyeTailLossFn launches (odd sizes: the masked scalar path)."""
    for M, K, N, ld in ((45, 128, 4, 8), (33, 512, 128, 132), (19, 6, 10, 12)):
        x = rnd((M, K), torch.float32, 320)
        w = rnd((N, K), torch.float32, 321, scale=K ** -0.5)
        b = rnd((N - 1,), torch.float32, 322)
        wide = torch.full((M, ld), 7.0, device='cuda')
        hip.linear_fwd_ex(dev(x), K, dev(w.t().contiguous()), dev(b), 3, wide)
        bfull = torch.cat([b, torch.zeros(1)])
        want = ref.linear_fwd(x, w.t().contiguous(), bfull, 3)
        assert torch.allclose(wide[:, :N].cpu(), want, rtol=2e-5, atol=2e-5) and bool((wide[:, N:] == 7.0).all())
        dyw = dev(rnd((M, ld), torch.float32, 323))
        base = dev(rnd((M, K), torch.float32, 324))
        dx = base.clone()
        hip.linear_dgrad_ex(dyw, N, wide, 3, dev(w), dx, accumulate=True)
        dx_want = ref.linear_dgrad(dyw[:, :N].cpu().contiguous(), want, 3, w) + base.cpu()
        assert torch.allclose(dx.cpu(), dx_want, rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize('H', [128, 96], ids=['h128_register_resident', 'h96_generic'])
def test_gru_scan(hip, ref, H):
    S, T = 5, 7
    gi = rnd((S, T, 3 * H), torch.float32, 31)
    whh = rnd((3 * H, H), torch.float32, 32, scale=H ** -0.5)
    bhh = rnd((3 * H,), torch.float32, 33, scale=0.1)
    for h0 in (None, rnd((S, H), torch.float32, 34, scale=0.5)):
        hs_w, g_w, hn_w = ref.gru_scan_fwd(gi, whh.t().contiguous(), bhh, h0)
        hs_g, g_g, hn_g = hip.gru_scan_fwd(dev(gi), dev(whh.t().contiguous()), dev(bhh), dev(h0))
        close(hs_g, hs_w, torch.float32, 'gru hs')
        close(g_g, g_w, torch.float32, 'gru gates')
        close(hn_g, hn_w, torch.float32, 'gru hn_pre')
        dhs = rnd((S, T, H), torch.float32, 35)
        w_ = ref.gru_scan_bwd(dhs, whh, h0, hs_w, g_w, hn_w, h0 is not None)
        g_ = hip.gru_scan_bwd(dev(dhs), dev(whh), dev(h0), dev(hs_w), dev(g_w), dev(hn_w), h0 is not None)
        close(g_[0], w_[0], torch.float32, 'gru dgi')
        close(g_[1], w_[1], torch.float32, 'gru dgh')
        if h0 is not None:
            close(g_[2], w_[2], torch.float32, 'gru dh0')


@pytest.mark.parametrize('H', [128, 48])
def test_rnn_and_lstm_scans(hip, ref, H):
    """nn.RNNCell / nn.LSTMCell over T (the other recurrent variants of eye_net.py:60-67): forward and backward."""
    S, T = 5, 6
    for G, name in ((1, 'rnn'), (4, 'lstm')):
        gi = rnd((S, T, G * H), torch.float32, 36)
        whh = rnd((G * H, H), torch.float32, 37, scale=H ** -0.5)
        bhh = rnd((G * H,), torch.float32, 38, scale=0.1)
        for with0 in (False, True):
            h0 = rnd((S, H), torch.float32, 39, scale=0.5) if with0 else None
            c0 = rnd((S, H), torch.float32, 40, scale=0.5) if with0 else None
            dhs = rnd((S, T, H), torch.float32, 41)
            if G == 1:
                hs_w = ref.rnn_scan_fwd(gi, whh.t().contiguous(), bhh, h0)
                hs_g = hip.rnn_scan_fwd(dev(gi), dev(whh.t().contiguous()), dev(bhh), dev(h0))
                close(hs_g, hs_w, torch.float32, 'rnn hs')
                w_ = ref.rnn_scan_bwd(dhs, whh, hs_w, with0)
                g_ = hip.rnn_scan_bwd(dev(dhs), dev(whh), dev(hs_w), with0)
            else:
                hs_w, cs_w, g_w = ref.lstm_scan_fwd(gi, whh.t().contiguous(), bhh, h0, c0)
                hs_g, cs_g, g_g = hip.lstm_scan_fwd(dev(gi), dev(whh.t().contiguous()), dev(bhh), dev(h0), dev(c0))
                close(hs_g, hs_w, torch.float32, 'lstm hs')
                close(cs_g, cs_w, torch.float32, 'lstm cs')
                close(g_g, g_w, torch.float32, 'lstm gates')
                dcs = rnd((S, T, H), torch.float32, 42) if with0 else None
                w_ = ref.lstm_scan_bwd(dhs, dcs, whh, c0, hs_w, cs_w, g_w, with0)
                g_ = hip.lstm_scan_bwd(dev(dhs), dev(dcs), dev(whh), dev(c0), dev(hs_w), dev(cs_w), dev(g_w), with0)
            for a, b in zip(g_, w_):
                if b is not None:
                    close(a, b, torch.float32, name + ' backward')


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
def test_cgru_gates(hip, ref, dtype):
    P, C = (3, 5, 8), 64
    g1, g2 = rnd(P + (2 * C,), dtype, 36), rnd(P + (C,), dtype, 37)
    h = rnd(P + (C,), dtype, 38, scale=0.5)
    ru_w, rh_w = ref.cgru_gates1(g1, h)
    ru_g, rh_g = hip.cgru_gates1(dev(g1), dev(h))
    close(ru_g, ru_w, dtype, 'gates1 ru')
    close(rh_g, rh_w, dtype, 'gates1 rh')
    o_w, hn_w = ref.cgru_gates2(g2, ru_w, h)
    o_g, hn_g = hip.cgru_gates2(dev(g2), dev(ru_w), dev(h))
    close(o_g, o_w, dtype, 'gates2 o')
    close(hn_g, hn_w, dtype, 'gates2 hnew')
    d = rnd(P + (C,), dtype, 39)
    for a, b, nm in zip(hip.cgru_gates2_bwd(dev(d), dev(ru_w), dev(h), dev(o_w)),
                        ref.cgru_gates2_bwd(d, ru_w, h, o_w), ('dg2', 'dru', 'dh')):
        close(a, b, dtype, 'gates2 bwd ' + nm)
    dru = rnd(P + (2 * C,), dtype, 40)
    for a, b, nm in zip(hip.cgru_gates1_bwd(dev(d), dev(dru), dev(ru_w), dev(h)),
                        ref.cgru_gates1_bwd(d, dru, ru_w, h), ('dg1', 'dh')):
        close(a, b, dtype, 'gates1 bwd ' + nm)


def test_adam_and_sumsq(hip, ref):
    n = 100003
    p, g = rnd((n,), torch.float32, 41), rnd((n,), torch.float32, 42, scale=3.0)
    m, v = 0.1 * rnd((n,), torch.float32, 43), rnd((n,), torch.float32, 44).abs()
    ss_w = ref.sumsq(g, torch.zeros(1))
    ss_g = hip.sumsq(dev(g), torch.zeros(1, device='cuda'))
    # fixed summation order: bit-reproducible (data-parallel replicas must clip by the identical factor)
    assert all(torch.equal(hip.sumsq(dev(g), torch.zeros(1, device='cuda')), ss_g) for _ in range(5))
    close(ss_g, ss_w, torch.float32, 'sumsq', scale=float(ss_w) * 4)
    pw, mw, vw = p.clone(), m.clone(), v.clone()
    ref.adam_step(pw, g, mw, vw, ss_w, 5.0, 1.0, 0.016, 0.9, 0.999, 1e-8, 0.005, 3)
    pg, mg, vg = dev(p.clone()), dev(m.clone()), dev(v.clone())
    hip.adam_step(pg, dev(g), mg, vg, ss_g, 5.0, 1.0, 0.016, 0.9, 0.999, 1e-8, 0.005, 3)
    close(pg, pw, torch.float32, 'adam p')
    close(mg, mw, torch.float32, 'adam m')
    close(vg, vw, torch.float32, 'adam v')
    # ---- the device-resident guard (eve_adam_guard): same update with the step counter on the device; a non-finite norm
    # skips the step (weights, moments, counter untouched), two skips in a row halve the loss scale; without check_finite the
    # overflow is NOT hidden ----
    guard_w, guard_g = ref.new_adam_guard('cpu', loss_scale=4.0, step=2), hip.new_adam_guard('cuda', loss_scale=4.0, step=2)
    pw2, mw2, vw2 = p.clone(), m.clone(), v.clone()
    pg2, mg2, vg2 = dev(p.clone()), dev(m.clone()), dev(v.clone())
    g4, ss4 = 4.0 * g, 16.0 * ss_w                               # what a loss scale of 4 leaves in the buffer
    ref.adam_step(pw2, g4, mw2, vw2, ss4, 5.0, 1.0, 0.016, 0.9, 0.999, 1e-8, 0.005, 0, guard=guard_w, check_finite=True)
    hip.adam_step(pg2, dev(g4), mg2, vg2, dev(ss4), 5.0, 1.0, 0.016, 0.9, 0.999, 1e-8, 0.005, 0, guard=guard_g, check_finite=True)
    close(pg2, pw, torch.float32, 'guarded adam p == plain step 3')
    close(pg2, pw2, torch.float32, 'guarded adam p')
    assert guard_g.cpu().tolist()[:4] == [3, 0, 0, 1] == guard_w.tolist()[:4]
    before = (pg2.clone(), mg2.clone(), vg2.clone())
    inf = torch.full((1,), float('inf'), device='cuda')
    for k_, want in ((1, [3, 1, 1, 0, 4.0]), (2, [3, 2, 0, 0, 2.0])):
        hip.adam_step(pg2, dev(g4), mg2, vg2, inf, 5.0, 1.0, 0.016, 0.9, 0.999, 1e-8, 0.005, 0, guard=guard_g, check_finite=True)
        ref.adam_step(pw2, g4, mw2, vw2, inf.cpu(), 5.0, 1.0, 0.016, 0.9, 0.999, 1e-8, 0.005, 0, guard=guard_w, check_finite=True)
        gl = guard_g.cpu()
        assert gl.tolist()[:4] == want[:4] == guard_w.tolist()[:4] and float(gl.view(torch.float32)[4]) == want[4]
        assert all(torch.equal(a, b) for a, b in zip(before, (pg2, mg2, vg2))), 'a skipped step must not touch weights or moments'
    gn = dev(g4).clone()
    gn[7] = float('nan')                                 # (a NaN NORM alone only switches the clip off; a NaN gradient poisons)
    hip.adam_step(pg2, gn, mg2, vg2, inf, 5.0, 1.0, 0.016, 0.9, 0.999, 1e-8, 0.005, 0, guard=guard_g, check_finite=False)
    assert not torch.isfinite(pg2[7]) and int(guard_g[0]) == 4, 'bf16 / fp32 runs must keep failing visibly'


def test_uint8_frame_normalisation_is_bit_identical_to_the_reference(hip):
    """eve_frames_u8_to_nchw / _to_stem against the fixture produced by the reference's own preprocess_frames /
    preprocess_screen_frames (tests/golden/make_golden_frames.py): every uint8 value, both scalings, bit for bit."""
    import os
    import numpy as np
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'frames.npz'))
    for key in ('ramp', 'frames'):
        u8 = torch.from_numpy(fx[key]).cuda()
        assert np.array_equal(hip.frames_u8_to_nchw(u8, 2.0 / 255.0, -1.0).cpu().numpy(), fx[key + '_eye'])
        assert np.array_equal(hip.frames_u8_to_nchw(u8, 1.0 / 255.0, None).cpu().numpy(), fx[key + '_screen'])
    # packed stem input from uint8 == packed stem input from the reference's float tensor
    g = np.random.Generator(np.random.PCG64(3))
    u8 = torch.from_numpy(g.integers(0, 256, size=(5, 32, 128, 3), dtype=np.uint8)).cuda()
    want = hip.stem_pack_input(hip.frames_u8_to_nchw(u8, 2.0 / 255.0, -1.0))
    got = hip.frames_u8_to_stem(u8, 2.0 / 255.0, -1.0)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


WGRAD_HALO_CASES = [
    # N, H, W, Cin, Cout      (3x3 / stride 1 / pad 1, bf16; EVE_WGRAD_HALO_MIN_M=0 sends them to wgrad_halo_kernel)
    (3, 10, 32, 16, 16),       # partial last band (TH = 4), one 32-pixel chunk per row
    (2, 72, 128, 16, 16),      # RefineNet level 0 (initial.3, final.0)
    (2, 72, 128, 16, 32),      # its first encoder convolution
    (2, 36, 64, 32, 32),       # 36 accumulator tiles, TH = 2
    (2, 72, 128, 64, 16),      # first decoder convolution (K = 576)
    (5, 7, 64, 32, 16),        # odd height
    (5, 32, 32, 64, 64),       # ResNet layer 1, 64 -> 64 channels: wgrad_halo64_kernel (three bands resident, TH = 4)
    (3, 21, 64, 64, 64),       # the same with a partial last band and two chunks per row (layer 1 of 256 x 256 patches)
    (300, 6, 32, 64, 64),      # more bands than workgroups: the persistent walk and both LDS regions
    (2, 72, 128, 16, 32, 1),   # 1x1 skip layer of the first encoder block
    (2, 72, 128, 64, 16, 1),   # 1x1 skip layer of the first decoder block
    (3, 36, 64, 32, 64, 1),    # 1x1 skip layer one level down
]


@pytest.mark.parametrize('case', WGRAD_HALO_CASES, ids=lambda c: 'N%d_%dx%d_c%d-%d' % c[:5] + ('_k%d' % c[5] if len(c) > 5 else ''))
def test_band_resident_weight_gradient(hip, ref, case):
    """wgrad_halo_kernel (few channels, large planes) against the ATen restatement and against wgrad_tr_kernel on the
    same inputs, with and without the fused bias gradient, accumulating onto existing values."""
    N, H, W, Cin, Cout = case[:5]
    K = case[5] if len(case) > 5 else 3
    pad = K // 2
    x = rnd((N, H, W, Cin), torch.bfloat16, 31)
    dy = rnd((N, H, W, Cout), torch.bfloat16, 32)
    want_dw = ref.conv2d_wgrad(x, dy, K, K, 1, pad, torch.zeros((Cout, K, K, Cin)))
    want_db = ref.bias_grad(dy, torch.zeros(Cout))
    dw0, db0 = rnd((Cout, K, K, Cin), torch.float32, 33), rnd((Cout,), torch.float32, 34)
    res = {}
    for mode, min_m in (('halo', 0), ('tr', 1 << 40)):
        with hip.dispatch_override(wgrad_halo_min_m=min_m):
            plain = hip.conv2d_wgrad(dev(x), dev(dy), K, K, 1, pad, dev(dw0).clone())
            assert hip.lib.eve_last_kernel().decode().startswith('wgrad_halo' if mode == 'halo' else 'wgrad_tr_kernel')
            fused_dw, fused_db = dev(dw0).clone(), dev(db0).clone()
            hip.conv2d_wgrad(dev(x), dev(dy), K, K, 1, pad, fused_dw, db=fused_db)
        res[mode] = (plain.cpu(), fused_dw.cpu(), fused_db.cpu())
        close(plain, want_dw + dw0, torch.bfloat16, 'wgrad ' + mode)
        close(fused_dw, want_dw + dw0, torch.bfloat16, 'wgrad+bias ' + mode)
        close(fused_db, want_db + db0, torch.bfloat16, 'bias ' + mode)
    # same bf16 products, float32 accumulation in a different order
    scale = float(want_dw.abs().max())
    assert float((res['halo'][0] - res['tr'][0]).abs().max()) <= 2e-4 * scale + 1e-4


def _random_conv_cases(n, seed):
    import random
    rng = random.Random(seed)
    cases = []
    while len(cases) < n:
        k = rng.choice([1, 3, 3, 3])
        stride = rng.choice([1, 1, 2])
        H, W = rng.randint(3, 40), rng.choice([4, 5, 8, 9, 16, 24, 32, 40, 64])
        if (H + 2 * (k // 2) - k) // stride + 1 < 1:
            continue
        cases.append((rng.randint(1, 9), H, W, rng.choice([8, 16, 24, 32, 64, 96, 128]), rng.choice([8, 16, 24, 32, 40, 64, 128, 136, 256]),
                      k, stride, k // 2))
    return cases


@pytest.mark.parametrize('case', _random_conv_cases(10, 7), ids=conv_case_id)
def test_conv_random_shapes_f32(hip, ref, case):
    """The float32 instantiation (parity mode: f32 MFMA / register-staged kernels) on randomly drawn shapes."""
    N, IH, IW, Cin, Cout, K, stride, pad = case
    x = rnd((N, IH, IW, Cin), torch.float32, 51)
    w = rnd((Cout, K, K, Cin), torch.float32, 52, scale=(2.0 / (K * K * Cin)) ** 0.5)
    bias = rnd((Cout,), torch.float32, 53)
    want = ref.conv2d_fwd(x, w, bias, stride, pad, 1)
    close(hip.conv2d_fwd(dev(x), dev(w), dev(bias), stride, pad, 1), want, torch.float32, 'conv fwd')
    dy = rnd(tuple(want.shape), torch.float32, 54)
    w_ihwo = w.permute(3, 1, 2, 0).contiguous()
    close(hip.conv2d_dgrad(dev(dy), dev(w_ihwo), (IH, IW), stride, pad), ref.conv2d_dgrad(dy, w_ihwo, (IH, IW), stride, pad),
          torch.float32, 'conv dgrad')
    got_dw, got_db = torch.zeros((Cout, K, K, Cin), device='cuda'), torch.zeros(Cout, device='cuda')
    hip.conv2d_wgrad(dev(x), dev(dy), K, K, stride, pad, got_dw, db=got_db)
    close(got_dw, ref.conv2d_wgrad(x, dy, K, K, stride, pad, torch.zeros((Cout, K, K, Cin))), torch.float32, 'conv wgrad')
    close(got_db, ref.bias_grad(dy, torch.zeros(Cout)), torch.float32, 'bias grad')


@pytest.mark.parametrize('case', _random_conv_cases(24, 2026), ids=conv_case_id)
def test_conv_random_shapes_bf16(hip, ref, case):
    """Shapes nobody tuned for (odd heights, widths that are not powers of two, channel counts between the tile sizes):
    whichever kernel the dispatcher picks must agree with the ATen restatement, forward, data and weight gradient."""
    N, IH, IW, Cin, Cout, K, stride, pad = case
    dtype = torch.bfloat16
    x = rnd((N, IH, IW, Cin), dtype, 41)
    w = rnd((Cout, K, K, Cin), dtype, 42, scale=(2.0 / (K * K * Cin)) ** 0.5)
    bias = rnd((Cout,), torch.float32, 43)
    want = ref.conv2d_fwd(x, w, bias, stride, pad, 2)
    got = hip.conv2d_fwd(dev(x), dev(w), dev(bias), stride, pad, 2)
    close(got, want, dtype, 'conv fwd')
    dy = rnd(tuple(want.shape), dtype, 44)
    w_ihwo = w.permute(3, 1, 2, 0).contiguous()
    close(hip.conv2d_dgrad(dev(dy), dev(w_ihwo), (IH, IW), stride, pad), ref.conv2d_dgrad(dy, w_ihwo, (IH, IW), stride, pad), dtype, 'conv dgrad')
    want_dw = ref.conv2d_wgrad(x, dy, K, K, stride, pad, torch.zeros((Cout, K, K, Cin)))
    got_dw, got_db = torch.zeros((Cout, K, K, Cin), device='cuda'), torch.zeros(Cout, device='cuda')
    hip.conv2d_wgrad(dev(x), dev(dy), K, K, stride, pad, got_dw, db=got_db)
    close(got_dw, want_dw, dtype, 'conv wgrad')
    close(got_db, ref.bias_grad(dy, torch.zeros(Cout)), dtype, 'bias grad')


def test_heatmap_head_and_losses_match_aten():
    """eve_heatmap_head_fwd/bwd (float sigmoid of the last convolution's channel-0 logits) and eve_heatmap_loss_fwd/bwd
    (per-frame BCE / MSE means + the validity reduction of base_loss_with_validity.py:64-73) against ATen."""
    import torch.nn.functional as F
    from eve_amd import losses, ops
    g = torch.Generator().manual_seed(5)
    for dt, tol in ((torch.bfloat16, 0.0), (torch.float32, 0.0)):
        logits = (torch.randn((6, 72, 128, 8), generator=g) * 4).to(dt).cuda().requires_grad_(True)
        y = ops.HeatmapHeadFn.apply(logits)
        ref = torch.sigmoid(logits.detach()[..., 0].float()).unsqueeze(1)
        assert y.dtype == torch.float32 and tuple(y.shape) == (6, 1, 72, 128)
        assert float((y - ref).abs().max()) < 1e-6
        dy = torch.randn(y.shape, generator=g).cuda()
        y.backward(dy)
        want = torch.zeros_like(logits.detach().float())
        want[..., 0] = (dy * ref * (1 - ref))[:, 0]
        assert float((logits.grad.float() - want.to(dt).float()).abs().max()) <= 1e-6 + (4e-3 * float(want.abs().max()) if dt == torch.bfloat16 else 0)
    B, T = 3, 5
    pred = torch.rand((B, T, 1, 72, 128), generator=g).clamp(1e-6, 1 - 1e-6)
    pred[0, 0, 0, 0, :4] = torch.tensor([0.0, 1.0, 1e-30, 1 - 1e-7])          # the log clamp and the 1e-12 gradient floor
    gt = torch.rand((B, T, 1, 72, 128), generator=g)
    valid = torch.rand((B, T), generator=g) > 0.3
    valid[1] = False
    valid[2, 1:] = False                                                       # a clip with a single valid frame
    for kind, fn in ((0, losses.bce_loss), (1, losses.mse_loss)):
        p_ref = pred.clone().requires_grad_(True)
        per = (F.binary_cross_entropy(p_ref, gt, reduction='none') if kind == 0 else (p_ref - gt) ** 2).flatten(2).mean(dim=2)
        v = valid.float()
        n = v.sum(dim=1)
        want = ((per * v).sum(dim=1) / torch.where(n > 1, n, torch.ones_like(n))).mean()
        (3.0 * want).backward()
        p_dev = pred.clone().cuda().requires_grad_(True)
        got = fn(p_dev, gt.cuda(), valid.cuda())
        assert 'HeatmapLossFn' in type(got.grad_fn).__name__
        np.testing.assert_allclose(float(got.detach()), float(want.detach()), rtol=2e-5)
        (3.0 * got).backward()
        a, b = p_dev.grad.cpu(), p_ref.grad
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-9, kind


def test_eight_wave_convolution_matches_the_four_wave_kernel(tmp_path):
    """conv3x3_wg8_kernel (conv_wg8.h: eight-wave workgroups, 32x32x16 MFMA, phase-staggered wave pairs) is the default for
    16 x 16 x 128, 8 x 8 x 256 and 4 x 4 x 512 layers whose tiles fill the chip; EVE_CONV_WG8=0 sends everything to
    conv3x3_halo_kernel.  Forward (bias + ReLU epilogue) and data gradient, ragged image counts (the last tile holds fewer
    images than TI), several channel tiles, both 16-bit formats: the two kernels sum the same products in float32 in a
    different order, so they agree to the format's rounding.  (The switch is read once per process.)"""
    import os
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from eve_amd.kernels import HipKernels
k = HipKernels()
g = torch.Generator().manual_seed(3)
outs, names = [], []
for dt in (torch.bfloat16, torch.float16):
    for N, H, C, Co in ((13, 16, 128, 128), (9, 8, 256, 256), (27, 8, 128, 512), (70, 4, 512, 512), (35, 4, 256, 256)):
        x = torch.randn((N, H, H, C), generator=g).to(dt).cuda()
        w = (torch.randn((Co, 3, 3, C), generator=g) * (2.0 / (9 * C)) ** 0.5).to(dt).cuda()
        b = torch.randn((Co,), generator=g).cuda()
        y = k.conv2d_fwd(x, w, b, 1, 1, epi_act=1)
        names.append(k.lib.eve_last_kernel().decode())
        dy = torch.randn(y.shape, generator=g).to(dt).cuda()
        dx = k.conv2d_dgrad(dy, w.permute(3, 1, 2, 0).contiguous(), (H, H), 1, 1)
        outs += [y.float().cpu(), dx.float().cpu()]
s2_outs, s2_names = [], []
for dt in (torch.bfloat16, torch.float16):
    # data gradient of the stride-2 3x3 layers (dy grid 16 / 8 / 4 wide): two eight-wave launches over the dy halo tile
    # (conv3x3_wg8_kernel<.., NT = 2 | 4>, depth-to-space epilogue) against the four per-tap parity-class launches
    for N, OW, Cdx, Co in ((5, 16, 64, 128), (9, 8, 128, 256), (35, 4, 256, 512), (3, 16, 128, 128)):
        dy = torch.randn((N, OW, OW, Co), generator=g).to(dt).cuda()
        w = (torch.randn((Co, Cdx, 3, 3), generator=g) * (2.0 / (9 * Co)) ** 0.5).to(dt).cuda()
        dx = k.conv2d_dgrad(dy, w.permute(1, 2, 3, 0).contiguous(), (2 * OW, 2 * OW), 2, 1)
        s2_names.append(k.lib.eve_last_kernel().decode())
        s2_outs.append(dx.float().cpu())
        # ... and their forward (conv3x3s2_wg8_kernel: the input's four parity planes as rotating halo stages), bias + ReLU
        x = torch.randn((N, 2 * OW, 2 * OW, Cdx), generator=g).to(dt).cuda()
        wf = (torch.randn((Co, 3, 3, Cdx), generator=g) * (2.0 / (9 * Cdx)) ** 0.5).to(dt).cuda()
        y = k.conv2d_fwd(x, wf, torch.randn((Co,), generator=g).cuda(), 2, 1, epi_act=1)
        s2_names.append(k.lib.eve_last_kernel().decode())
        s2_outs.append(y.float().cpu())
torch.save((outs, names, s2_outs, s2_names), sys.argv[1])
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ('0', '2'):
        path = os.path.join(str(tmp_path), 'wg8_%s.pt' % mode)
        # (EVE_CONV_WG8_MIN_TILES=0: the launcher otherwise keeps wg8 for launches whose tiles fill the chip)
        env = dict(os.environ, EVE_CONV_WG8=mode, EVE_CONV_WG8_MIN_TILES='0', EVE_CONV_WG8_S2_MIN_TILES='0')
        p = subprocess.run([sys.executable, '-c', code, path], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:]
        res[mode] = torch.load(path)
    assert all('wg8' in n for n in res['2'][1]) and not any('wg8' in n for n in res['0'][1]), (res['2'][1], res['0'][1])
    for i, (a, b) in enumerate(zip(res['2'][0], res['0'][0])):
        tol = 2e-3 if i < 10 else 3e-4                       # bf16 cases first, then float16
        assert float((a - b).norm() / b.norm()) < tol and float((a - b).abs().max()) <= 8 * tol * float(b.abs().max()), i
    assert all(('s2dgrad' in n or 'conv3x3s2_wg8' in n) for n in res['2'][3]) and not any('wg8' in n for n in res['0'][3]), (res['2'][3], res['0'][3])
    for i, (a, b) in enumerate(zip(res['2'][2], res['0'][2])):
        tol = 2.5e-3 if i < 8 else 3e-4
        assert float((a - b).norm() / b.norm()) < tol and float((a - b).abs().max()) <= 8 * tol * float(b.abs().max()), ('s2', i)


@pytest.mark.parametrize('B,T', [(2, 4), (7, 30), (32, 120), (3, 1)])
def test_batched_vector_terms_match_the_tensor_expressions(B, T):
    """Round 5: eve_vector_terms (every masked [B, T, D <= 3] loss / metric of eve.py:286-439 in one launch, ops.VectorTermsFn)
    against the tensor expressions of eve_amd/losses.py it replaces (themselves pinned by the reference's golden scalars):
    MSE, Euclidean, L1 (D = 1, 2, 3) and the angular error, clips with 0 / 1 / several valid steps, values and gradients."""
    from eve_amd import losses, ops
    g = torch.Generator().manual_seed(5)
    dev = 'cuda'
    items, fns = [], []
    for kind, fn, D in (('mse', losses.mse_loss, 2), ('mse', losses.mse_loss, 3), ('euclidean', losses.euclidean_loss, 2),
                        ('l1', losses.l1_loss, 1), ('l1', losses.l1_loss, 2), ('angular', losses.angular_loss, 2),
                        ('angular', losses.angular_loss, 2)):
        shape = (B, T) if (D == 1 and kind == 'l1') else (B, T, D)
        scale = 0.4 if kind == 'angular' else 30.0
        pred = (torch.randn(shape, generator=g) * scale).to(dev).requires_grad_(kind != 'euclidean')
        tgt = (torch.randn(shape, generator=g) * scale).to(dev)
        val = (torch.rand((B, T), generator=g) < 0.7).to(dev)
        val[0] = False                                   # a clip without a valid step
        if B > 1:
            val[1] = False
            val[1, 0] = True                             # exactly one valid step: not divided by the count
        items.append((kind, pred, tgt, val))
        fns.append(fn)
    if T > 2:                                            # identical vectors: cos = 1, the clamp's zero-gradient branch
        with torch.no_grad():
            items[-1][1][:, 2] = items[-1][2][:, 2]
    got = ops.VectorTermsFn.apply(tuple((k_, t_, v_) for k_, _, t_, v_ in items), *[it[1] for it in items])
    weights = [float(i + 1) for i in range(len(items))]
    sum(w * v for w, v in zip(weights, got) if v.requires_grad).backward()
    mine = [None if it[1].grad is None else it[1].grad.clone() for it in items]
    for (kind, pred, tgt, val), fn, gv, w, gm in zip(items, fns, got, weights, mine):
        p2 = pred.detach().clone().requires_grad_(True)
        want = fn(p2, tgt, val)
        assert abs(float(gv) - float(want)) <= 2e-5 * max(1.0, abs(float(want))), (kind, float(gv), float(want))
        if kind != 'euclidean':
            (w * want).backward()
            ref = p2.grad
            if kind == 'angular':                        # torch's acos / cosine_similarity backward is NaN-prone at cos = +-1
                ok = torch.isfinite(ref).all(dim=-1)
                assert float((gm - ref)[ok].abs().max()) <= 2e-3 * float(ref[ok].abs().max()) + 1e-6
                assert bool(torch.isfinite(gm).all())
            else:
                assert float((gm - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-7, kind


@pytest.mark.parametrize('hdt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
@pytest.mark.parametrize('geom', [(8, 72, 128), (29, 37, 64)], ids=['72x128', '37x64'])
@pytest.mark.parametrize('cin,cout', [(16, 16), (16, 32), (32, 16), (32, 32), (16, 64), (64, 16), (32, 64), (32, 128), (64, 64)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize('offset', [3.0, 60.0], ids=['bias3', 'bias60'])
def test_convolution_epilogue_emits_instancenorm_statistics(hip, hdt, geom, cin, cout, offset):
    """Round 5: eve_conv2d_fwd_stats -- the row-streaming 3x3 kernel walks whole images, so it forms the InstanceNorm statistics
    (mean, rstd per plane, on the STORED values) of its output in its epilogue and the consumer's statistics pass is dropped
    (refine_net.py:45-53: every convolution of a pre-activation block is followed by InstanceNorm2d).  Against eve_instnorm_stats
    on the output it wrote, with a bias large against the spread -- 5 and (round 6) ~100 standard deviations: the sums
    are taken about each lane's first value and merged as (mean, M2) pairs, a plain E[x^2] - mean^2 in float does not survive the
    second case; a shape the kernel does not serve reports "not written" and gives the plain convolution."""
    N, H, W = geom
    x = rnd((N, H, W, cin), hdt, 91)
    w = rnd((cout, 3, 3, cin), hdt, 92, scale=(2.0 / (9 * cin)) ** 0.5)
    b = rnd((cout,), torch.float32, 93, scale=0.2) + offset
    y, mr = hip.conv2d_fwd_stats(dev(x), dev(w), dev(b), 1, 1)
    used = hip.lib.eve_last_kernel().decode()
    plain = hip.conv2d_fwd(dev(x), dev(w), dev(b), 1, 1)
    assert torch.equal(y, plain)
    if (cin, cout) == (64, 64):
        assert mr is None and 'conv3x3_stream' not in used
        return
    assert used.startswith('conv3x3_stream_kernel<') and mr is not None and tuple(mr.shape) == (N, cout, 2)
    want = hip.instnorm_stats(y, 1e-5)
    assert float((mr[..., 0] - want[..., 0]).abs().max()) <= 2e-5 * float(want[..., 0].abs().max())
    assert float(((mr[..., 1] - want[..., 1]) / want[..., 1]).abs().max()) <= 2e-3      # rstd: var = E[x^2] - mean^2 at mean ~ 5 std
    # the module path: RefineNet's mid-block InstanceNorm on these statistics == on its own statistics pass
    from eve_amd import ops
    g = rnd((cout,), torch.float32, 94).abs() + 0.5
    be = rnd((cout,), torch.float32, 95)
    a1 = ops.instnorm_act(y, dev(g), dev(be), act=1, stats=mr)
    a2 = ops.instnorm_act(y, dev(g), dev(be), act=1)
    close(a1, a2, hdt, 'InstanceNorm on epilogue statistics', scale=float(a2.float().abs().max()))
