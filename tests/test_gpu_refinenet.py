"""GPU parity: eve_amd.RefineNet (HIP kernels through the C ABI) against the golden fixtures made by the
reference RefineNet / conv-RNN cells (tests/golden/refinenet.npz, cells.npz) and the CPU oracle."""
import os

import numpy as np
import pytest
import torch
from eve_amd.kernels import default_kernels

from oracle import detweights, sequence
from oracle.config import OracleConfig

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
HEAT_TOL = 1e-4


def make_net(kind, dtype=torch.float32, screen=True):
    import eve_amd
    cfg = eve_amd.reset_standalone_config()
    cfg.import_dict({'load_screen_content': screen, 'refine_net_enabled': True, 'refine_net_rnn_type': kind})
    net = eve_amd.RefineNet()
    net.compute_dtype = dtype
    detweights.fill_module(net, seed=1)
    return net.cuda(), cfg


@pytest.mark.parametrize('kind', ['CGRU', 'CLSTM', 'CRNN'])
def test_refinenet_sequence_and_per_step_match_reference_golden(kind):
    fx = np.load(os.path.join(GOLDEN, 'refinenet.npz'))
    B, T = int(fx['B']), int(fx['T'])
    rb = detweights.refinenet_batch(B, T, seed=0, invalid_fraction=float(fx['invalid_fraction']))
    drb = {k: v.cuda() for k, v in rb.items()}
    net, cfg = make_net(kind)
    hf, states = net.forward_sequence(drb['heatmap_initial'], drb['screen_frame'])
    want = fx[kind + '_heatmap_final']
    got = hf.detach().cpu().numpy()
    got = got if kind == 'CGRU' else got[..., ::4, ::4]
    assert np.abs(got - want).max() < HEAT_TOL
    assert want.std() > 1e-3
    st = states[0]
    last = (st[0] if isinstance(st, tuple) else st)[:, -1]
    assert np.abs(last.detach().cpu().numpy() - fx[kind + '_state_last']).max() < HEAT_TOL
    if isinstance(st, tuple):
        assert np.abs(st[1][:, -1].cpu().numpy() - fx[kind + '_cell_last']).max() < HEAT_TOL
    # the reference's per-step dict contract (eve.py:145-147)
    outs, prev = [], None
    with torch.no_grad():
        for t in range(T):
            so = {'heatmap_initial': drb['heatmap_initial'][:, t]}
            net({'screen_frame': drb['screen_frame'][:, t]}, so, previous_output_dict=prev)
            outs.append(so['heatmap_final'])
            prev = so
    stepped = torch.stack(outs, dim=1).cpu().numpy()
    stepped = stepped if kind == 'CGRU' else stepped[..., ::4, ::4]
    assert np.abs(stepped - want).max() < HEAT_TOL
    # losses + gradients
    from eve_amd import losses
    terms = losses.refinenet_loss_terms(hf, drb['heatmap_final_gt'], drb['validity'], cfg)
    np.testing.assert_allclose(float(terms['loss_ce_heatmap_final'].detach()), float(fx[kind + '_loss_ce']), rtol=2e-5)
    np.testing.assert_allclose(float(terms['loss_mse_heatmap_final'].detach()), float(fx[kind + '_loss_mse']), rtol=2e-5)
    terms['full_loss'].backward()
    params = dict(net.named_parameters())
    dead = 0
    for n, ref_norm in zip(fx[kind + '_grad_names'], fx[kind + '_grad_norms']):
        p = params[str(n)]
        if ref_norm < 0:
            assert p.grad is None, n
            dead += 1
        else:
            got = float(p.grad.double().norm())
            assert abs(got - ref_norm) <= 2e-2 * ref_norm + 3e-5, '%s: %.6g vs %.6g' % (n, got, ref_norm)
    assert dead == (2 if kind == 'CLSTM' else 0)


def test_refinenet_without_screen_content():
    fx = np.load(os.path.join(GOLDEN, 'refinenet.npz'))
    rb = detweights.refinenet_batch(2, 3, seed=0, invalid_fraction=0.25)
    net, _ = make_net('CGRU', screen=False)
    out = {'heatmap_initial': rb['heatmap_initial'][:, 0].cuda()}
    with torch.no_grad():
        net({}, out)
    assert np.abs(out['heatmap_final'].cpu().numpy()[..., ::4, ::4] - fx['noscreen_heatmap_final']).max() < HEAT_TOL


def test_cgru_cell_matches_reference_golden_including_grads():
    """The CGRU cell alone (common.py:388-415) through the module's own step function."""
    fx = np.load(os.path.join(GOLDEN, 'cells.npz'))
    net, _ = make_net('CGRU')
    bott = net.network
    prefix = 'network'
    while hasattr(bott, 'between_module'):
        bott, prefix = bott.between_module, prefix + '.between_module'
    cell = bott.rnn_cells[0]
    detweights.fill_module(cell, seed=3)
    net.invalidate_packs()
    P = net._get_packs()
    from eve_amd import ops
    x = torch.from_numpy(fx['CGRU_x']).cuda().requires_grad_()
    h = torch.from_numpy(fx['CGRU_h']).cuda().requires_grad_()
    xn = ops.ToNHWCFn.apply(x, torch.float32, 64)
    hn = ops.ToNHWCFn.apply(h, torch.float32, 64)
    out, _ = net._cell_step(xn, hn, cell, prefix + '.rnn_cells.0', P)
    hnew = ops.FromNHWCFn.apply(out, 64)
    assert np.abs(hnew.detach().cpu().numpy() - fx['CGRU_h_new']).max() < 2e-5
    (hnew * hnew).sum().backward()
    assert np.abs(x.grad.cpu().numpy() - fx['CGRU_dx']).max() < 2e-4
    assert np.abs(h.grad.cpu().numpy() - fx['CGRU_dh']).max() < 2e-4
    out0, _ = net._cell_step(xn.detach(), None, cell, prefix + '.rnn_cells.0', P)
    assert np.abs(ops.FromNHWCFn.apply(out0, 64).detach().cpu().numpy() - fx['CGRU_h_new_from_none']).max() < 2e-5


def test_refinenet_bf16_deviation_reported():
    """Sanity bound only (bf16 storage vs the reference's float32 fixture); the bf16 parity statement is
    tests/test_gpu_bf16_parity.py (teacher-forced stages against the rounding-faithful oracle + noise envelope)."""
    fx = np.load(os.path.join(GOLDEN, 'refinenet.npz'))
    rb = detweights.refinenet_batch(2, 3, seed=0, invalid_fraction=0.25)
    net, _ = make_net('CGRU', dtype=torch.bfloat16)
    with torch.no_grad():
        hf, _ = net.forward_sequence(rb['heatmap_initial'].cuda(), rb['screen_frame'].cuda())
    dev = np.abs(hf.cpu().numpy() - fx['CGRU_heatmap_final']).max()
    print('RefineNet bf16 max heat-map deviation vs reference fp32: %.4e' % dev)
    assert dev < 0.3      # untrained deterministic weights through ~40 bf16 layers; a precision mode, not parity


@pytest.mark.parametrize('per_wg', [1, 3], ids=['one-sequence-per-workgroup', 'three-per-workgroup'])
@pytest.mark.parametrize('B,T,with_h0', [(2, 3, False), (7, 5, True), (3, 1, True)])
def test_fused_cgru_scan_kernel_matches_per_step_path(B, T, with_h0, per_wg):
    """eve_cgru_scan_fwd (the whole clip through the conv-GRU in one persistent launch) == the per-frame path
    (two conv launches + two gate kernels per frame), bf16; outputs and the tensors the backward consumes."""
    from eve_amd.kernels import HipKernels
    import fake_kernels
    hip, ref = HipKernels(), fake_kernels.FakeKernels()
    g = torch.Generator().manual_seed(11)
    xs = (torch.randn((B, T, 5, 8, 64), generator=g) * 0.8).bfloat16()
    h0 = (torch.randn((B, 5, 8, 64), generator=g) * 0.5).bfloat16() if with_h0 else None
    w1 = (torch.randn((128, 3, 3, 128), generator=g) * 0.04).bfloat16()
    w2 = (torch.randn((64, 3, 3, 128), generator=g) * 0.04).bfloat16()
    b1, b2 = torch.randn((128,), generator=g) * 0.2, torch.randn((64,), generator=g) * 0.2
    want = ref.cgru_scan_fwd(xs, h0, w1, b1, w2, b2)
    # round 5: cgru_scan1.hip (one sequence per workgroup, the default up to cgru_seq_max_b sequences) and cgru_scan.hip
    with hip.dispatch_override(cgru_seq_max_b=384 if per_wg == 1 else 0):
        got = hip.cgru_scan_fwd(xs.cuda(), h0.cuda() if with_h0 else None, w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda())
        assert hip.lib.eve_last_kernel().decode().startswith('cgru_scan1_fwd_kernel' if per_wg == 1 else 'cgru_scan_fwd_kernel')
    for name, a, b in zip(('hs', 'hs_tm', 'ru', 'rh', 'og'), got, want):
        d = (a.float().cpu() - b.float()).abs()
        # bf16 storage: one ulp at |v| <= 1 is 2^-8; the fused kernel rounds at fewer points than the per-frame path
        assert float(d.max()) < 3e-2 and float(d.mean()) < 2e-3, (name, float(d.max()), float(d.mean()))


def test_refinenet_fused_scan_trains_like_the_per_step_path():
    """RefineNet bf16 forward + backward with the fused conv-GRU scan vs eve_dispatch_config.cgru_scan = 0: heat-maps and gradients."""
    rb = detweights.refinenet_batch(3, 4, seed=3)
    outs = {}
    for mode in ('1', '0'):
        with default_kernels().dispatch_override(cgru_scan=int(mode)):
            net, _ = make_net('CGRU', dtype=torch.bfloat16)
            hf, states = net.forward_sequence(rb['heatmap_initial'].cuda(), rb['screen_frame'].cuda())
            (hf.float() * rb['heatmap_final_gt'].cuda()).sum().backward()
        outs[mode] = (hf.detach().float().cpu(), states[0].detach().float().cpu(),
                      {n: p.grad.detach().float().cpu() for n, p in net.named_parameters()})
    a, b = outs['1'], outs['0']
    assert float((a[0] - b[0]).abs().max()) < 0.05
    assert float((a[1] - b[1]).abs().max()) < 0.06
    for n in b[2]:
        ga, gb = a[2][n], b[2][n]
        if ga.dim() < 2:
            continue          # biases feeding an InstanceNorm have an exactly-zero gradient: what is computed is rounding noise
        # two bf16 evaluation orders of the same network: a few per cent of relative L2 is rounding noise, direction must agree
        assert float((ga - gb).norm()) <= 0.15 * float(gb.norm()) + 1e-4, n
        if float(gb.norm()) > 1e-3:
            assert float((ga * gb).sum() / (ga.norm() * gb.norm())) > 0.985, n


@pytest.mark.parametrize('cin,cout,H,W,ks', [(16, 16, 72, 128, 3), (16, 32, 36, 64, 3), (8, 16, 72, 128, 3),
                                              (32, 16, 9, 16, 3), (32, 32, 72, 128, 3), (32, 16, 72, 128, 3), (16, 32, 72, 128, 1), (64, 16, 72, 128, 1),
                                              (16, 8, 72, 128, 1), (32, 64, 36, 64, 1)])
def test_pixel_group_convolution_on_the_halo_kernel(monkeypatch, cin, cout, H, W, ks):
    """8/16-channel 3x3 convolutions (RefineNet's outer level) run as 32-channel convolutions over pixel groups on the
    halo-resident kernel (1x1: 64-channel groups on the LDS-DMA gather kernel): same forward, data, weight and bias gradients as the generic kernel on the plain layout."""
    from eve_amd import ops
    torch.manual_seed(cin * 100 + cout)
    x0 = torch.randn(3, H, W, cin).bfloat16().cuda()
    wt0 = (torch.randn(cout, cin, ks, ks) * (2.0 / (ks * ks * cin)) ** 0.5).cuda()
    b0 = torch.randn(cout).cuda()
    gy = torch.randn(3, H, W, cout).bfloat16().cuda()
    outs = []
    for grouped in (True, False):
        if not grouped:
            monkeypatch.setattr(ops, 'PAIR_FACTOR', {})
            monkeypatch.setattr(ops, 'PAIR_FACTOR_1X1', {})
        x = x0.clone().requires_grad_(True)
        wt, b = wt0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        pack = ops.PackedWeight(wt, torch.bfloat16)
        y = ops.conv2d(x, wt, b, pack, stride=1, pad=ks // 2, act=1)
        y.backward(gy)
        outs.append((y.detach().float().cpu(), x.grad.float().cpu(), wt.grad.cpu(), b.grad.cpu()))
    for name, a, c in zip(('y', 'dx', 'dw', 'db'), *outs):
        # same products, different fp32 summation order, one bf16 rounding at the end
        assert float((a - c).abs().max()) <= 1e-2 * float(c.abs().max()) + 1e-6, name


def test_full_size_refinenet_is_batch_invariant_deterministic_and_dp_linear():
    """BASELINE configs[2] at full size (B=32 clips x T=30, bf16, fused conv-GRU scan) through properties that need no
    oracle: clips are independent units (InstanceNorm is per frame, the conv-GRU per clip), so every clip's heat-maps are
    bit-identical whether it runs in the batch of 32 or in a batch of 8, a repeated forward is bit-identical, and the
    data-parallel identity grad(batch) = mean of grad(halves) holds to atomic-order noise."""
    from eve_amd import losses
    small = detweights.refinenet_batch(8, 30, seed=21, invalid_fraction=0.1)
    g = torch.Generator().manual_seed(0)
    full = {}
    for k, v in small.items():                       # 32 distinct clips: 4 perturbed copies of the 8 generated ones
        reps = [v] + [(v + 0.03 * torch.rand(v.shape, generator=g)).clamp(0, 1) if k == 'screen_frame' else v for _ in range(3)]
        full[k] = torch.cat(reps, dim=0).cuda()
    net, cfg = make_net('CGRU', dtype=torch.bfloat16)
    with torch.no_grad():
        a, sa = net.forward_sequence(full['heatmap_initial'], full['screen_frame'])
        b, sb = net.forward_sequence(full['heatmap_initial'], full['screen_frame'])
        assert torch.equal(a, b) and torch.equal(sa[0], sb[0]), 'not deterministic'
        for i in range(0, 32, 8):
            part, _ = net.forward_sequence(full['heatmap_initial'][i:i + 8].contiguous(), full['screen_frame'][i:i + 8].contiguous())
            assert torch.equal(part, a[i:i + 8]), 'clip outputs depend on the batch: clips %d..' % i
    assert float(a.std()) > 1e-3 and tuple(a.shape) == (32, 30, 1, 72, 128)

    def grads(sl):
        net.zero_grad(set_to_none=True)
        hf, _ = net.forward_sequence(full['heatmap_initial'][sl].contiguous(), full['screen_frame'][sl].contiguous())
        losses.refinenet_loss_terms(hf, full['heatmap_final_gt'][sl].contiguous(), full['validity'][sl].contiguous(), cfg)['full_loss'].backward()
        return torch.cat([p.grad.detach().float().reshape(-1) for p in net.parameters()])
    whole = grads(slice(0, 32))
    halves = 0.5 * (grads(slice(0, 16)) + grads(slice(16, 32)))
    rel = float((whole - halves).norm() / whole.norm())
    assert rel < 5e-3, rel


@pytest.mark.parametrize('per_wg', [1, 3], ids=['one-sequence-per-workgroup', 'three-per-workgroup'])
@pytest.mark.parametrize('B,T,with_h0', [(2, 3, False), (7, 5, True), (3, 1, True), (32, 30, False)])
def test_fused_cgru_scan_backward_kernel_matches_contract(B, T, with_h0, per_wg):
    """eve_cgru_scan_bwd (the whole frame-reversed conv-GRU backward in one persistent launch: gate gradients, both
    data-gradient GEMMs on MFMA, the float carry into the previous state) against the ATen restatement of its contract:
    gradients of the two pre-activations, d xs, d h0."""
    from eve_amd.kernels import HipKernels
    import fake_kernels
    hip, ref = HipKernels(), fake_kernels.FakeKernels()
    g = torch.Generator().manual_seed(17)
    bf = lambda *shape, scale=1.0: (torch.randn(shape, generator=g) * scale).bfloat16()
    dhs = bf(T, B, 5, 8, 64)
    ru = torch.sigmoid(torch.randn((T, B, 5, 8, 128), generator=g)).bfloat16()
    og = torch.tanh(torch.randn((T, B, 5, 8, 64), generator=g)).bfloat16()
    hs = bf(T, B, 5, 8, 64, scale=0.6)
    h0 = bf(B, 5, 8, 64, scale=0.5) if with_h0 else None
    w1t, w2t = bf(128, 3, 3, 128, scale=0.04), bf(128, 3, 3, 64, scale=0.05)
    want = ref.cgru_scan_bwd(dhs, ru, og, hs, h0, w1t, w2t, want_dh0=with_h0)
    with hip.dispatch_override(cgru_seq_max_b=384 if per_wg == 1 else 0):
        got = hip.cgru_scan_bwd(dhs.cuda(), ru.cuda(), og.cuda(), hs.cuda(), h0.cuda() if with_h0 else None, w1t.cuda(), w2t.cuda(),
                                want_dh0=with_h0)
        assert hip.lib.eve_last_kernel().decode().startswith('cgru_scan1_bwd_kernel' if per_wg == 1 else 'cgru_scan_bwd_kernel')
    for name, a, b in zip(('dg1', 'dg2', 'dxs', 'dh0'), got, want):
        if b is None:
            assert a is None
            continue
        a, b = a.float().cpu(), b.float()
        rel = float((a - b).norm() / b.norm())
        # bf16 storage of every output; the recursion amplifies a rounding flip of an early frame's dg by the carry
        assert rel < 6e-3 and float((a - b).abs().max()) <= 4e-2 * float(b.abs().max()), (name, rel, float((a - b).abs().max()))


@pytest.mark.parametrize('B,T,with_h0', [(2, 3, False), (7, 5, True), (3, 1, True), (32, 30, False)])
def test_float32_clip_scans_match_the_per_frame_contract(B, T, with_h0):
    """Round 5: the float32 clip-long scans (csrc/cell_scan_f32.hip: CGRU forward + backward, CRNN forward + backward, CLSTM
    forward; one persistent launch per clip and direction, v_mfma_f32_16x16x4_f32) against the ATen restatement of the per-frame
    contract (two / one convolutions + gate math per frame, common.py:331-415), float32 tolerances."""
    from eve_amd.kernels import HipKernels
    import fake_kernels
    hip, ref = HipKernels(), fake_kernels.FakeKernels()
    g = torch.Generator().manual_seed(23)
    rn = lambda *shape, scale=1.0: torch.randn(shape, generator=g) * scale
    cu = lambda t: None if t is None else t.cuda()
    xs, h0 = rn(B, T, 5, 8, 64, scale=0.8), (rn(B, 5, 8, 64, scale=0.5) if with_h0 else None)
    c0 = rn(B, 5, 8, 64, scale=0.5) if with_h0 else None

    def close(tag, got, want, tol=2e-5):
        for name, a, b in zip(tag, got, want):
            if b is None:
                assert a is None, name
                continue
            a, b = a.float().cpu(), b.float()
            assert tuple(a.shape) == tuple(b.shape), name
            err, ref_max = float((a - b).abs().max()), float(b.abs().max())
            assert err <= tol * max(1.0, ref_max), (name, err, ref_max)

    # CGRU forward
    w1, w2 = rn(128, 3, 3, 128, scale=0.04), rn(64, 3, 3, 128, scale=0.04)
    b1, b2 = rn(128, scale=0.2), rn(64, scale=0.2)
    want = ref.cgru_scan_fwd(xs, h0, w1, b1, w2, b2)
    got = hip.cgru_scan_fwd(cu(xs), cu(h0), cu(w1), cu(b1), cu(w2), cu(b2))
    assert hip.lib.eve_last_kernel().decode() == 'cgru_scan_f32_fwd_kernel'
    close(('hs', 'hs_tm', 'ru', 'rh', 'og'), got, want)
    # CGRU backward, on the forward's own tensors
    dhs = rn(T, B, 5, 8, 64)
    w1t, w2t = w1.permute(3, 1, 2, 0).contiguous(), w2.permute(3, 1, 2, 0).contiguous()
    hs_tm, ru, _, og = want[1:]
    want_b = ref.cgru_scan_bwd(dhs, ru, og, hs_tm, h0, w1t, w2t, want_dh0=with_h0)
    got_b = hip.cgru_scan_bwd(cu(dhs), cu(ru), cu(og), cu(hs_tm), cu(h0), cu(w1t), cu(w2t), want_dh0=with_h0)
    assert hip.lib.eve_last_kernel().decode() == 'cgru_scan_f32_bwd_kernel'
    close(('dg1', 'dg2', 'dxs', 'dh0'), got_b, want_b, tol=1e-4 if T > 8 else 3e-5)
    # CRNN forward + backward
    w, bias = rn(64, 3, 3, 128, scale=0.04), rn(64, scale=0.2)
    want = ref.crnn_scan_fwd(xs, h0, w, bias)
    got = hip.crnn_scan_fwd(cu(xs), cu(h0), cu(w), cu(bias))
    close(('hs', 'hs_tm'), got, want)
    wt = w.permute(3, 1, 2, 0).contiguous()
    want_b = ref.crnn_scan_bwd(dhs, want[1], wt, want_dh0=with_h0)
    got_b = hip.crnn_scan_bwd(cu(dhs), cu(want[1]), cu(wt), want_dh0=with_h0)
    close(('dpre', 'dxs', 'dh0'), got_b, want_b, tol=1e-4 if T > 8 else 3e-5)
    # CLSTM forward (state (h, c); gate order in / forget / out / cell)
    w, bias = rn(256, 3, 3, 128, scale=0.04), rn(256, scale=0.2)
    want = ref.clstm_scan_fwd(xs, h0, c0, w, bias)
    got = hip.clstm_scan_fwd(cu(xs), cu(h0), cu(c0), cu(w), cu(bias))
    close(('hs', 'cs'), got, want)


@pytest.mark.parametrize('kind', ['CGRU', 'CRNN', 'CLSTM'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_refinenet_clip_scans_train_like_the_per_frame_path(kind, dtype):
    """RefineNet forward + backward through the clip-long scans vs eve_dispatch_config.cgru_scan = 0 (the per-frame launches),
    every cell type, float32 (tight) and bf16 (CRNN / CLSTM: the float32 scan behind 16-bit convolutions)."""
    rb = detweights.refinenet_batch(3, 4, seed=3)
    outs = {}
    for mode in ('1', '0'):
        with default_kernels().dispatch_override(cgru_scan=int(mode)):
            net, _ = make_net(kind, dtype=dtype)
            hf, states = net.forward_sequence(rb['heatmap_initial'].cuda(), rb['screen_frame'].cuda())
            (hf.float() * rb['heatmap_final_gt'].cuda()).sum().backward()
        st = states[0]
        outs[mode] = (hf.detach().float().cpu(), [t.detach().float().cpu() for t in (st if isinstance(st, tuple) else (st,))],
                      {n: p.grad.detach().float().cpu() for n, p in net.named_parameters() if p.grad is not None})
    a, b = outs['1'], outs['0']
    f32 = dtype == torch.float32
    assert float((a[0] - b[0]).abs().max()) < (2e-5 if f32 else 0.05)
    for sa, sb in zip(a[1], b[1]):
        assert tuple(sa.shape) == tuple(sb.shape)
        assert float((sa - sb).abs().max()) < (2e-5 if f32 else 0.06)
    assert set(a[2]) == set(b[2])
    for n in b[2]:
        ga, gb = a[2][n], b[2][n]
        if ga.dim() < 2:
            continue          # biases feeding an InstanceNorm have an exactly-zero gradient: what is computed is rounding noise
        assert float((ga - gb).norm()) <= (2e-3 if f32 else 0.15) * float(gb.norm()) + 1e-4, n


def test_float32_refinenet_gradients_match_the_reference_float64_full_tensors():
    """tests/golden/grads_f64.npz: the reference's RefineNet (CGRU, per-step contract over T = 3) + CrossEntropyLoss in
    float64.  The float32 HIP path: heat-map BCE to 2e-6; the FULL gradient tensors of the conv-GRU's gates_1 / gate_2
    banks and of the first / last convolution, and every parameter's gradient norm, within max(1e-4, 2 x the deviation of the
    reference's OWN float32 run from its float64 run on that parameter) -- 1e-4 on the last convolution, 7.6e-3 on the gate
    banks, whose gradient passes the encoder's max-pool / ReLU ties."""
    from eve_amd import losses
    fx = np.load(os.path.join(GOLDEN, 'grads_f64.npz'))
    rb = detweights.refinenet_batch(2, 3, seed=0, invalid_fraction=0.25)
    drb = {k: v.cuda() for k, v in rb.items()}
    net, cfg = make_net('CGRU')
    hf, _ = net.forward_sequence(drb['heatmap_initial'], drb['screen_frame'])
    terms = losses.refinenet_loss_terms(hf, drb['heatmap_final_gt'], drb['validity'], cfg)
    np.testing.assert_allclose(float(terms['loss_ce_heatmap_final'].detach()), float(fx['refine_loss_ce']), rtol=2e-6)
    terms['loss_ce_heatmap_final'].backward()
    params = dict(net.named_parameters())
    scale = float(fx['refine_norms'].max())
    # adaptive max-pool / (leaky-)ReLU decisions on float ties re-route gradient in ANY float32 evaluation: the reference's
    # own float32 run is up to 6e-3 away from its float64 run on the encoder side (refine_ref_f32_dev, per parameter) and
    # ~1e-5 from the bottleneck on.  The HIP float32 path must stay within 1e-4 where float32 itself does, and within the
    # reference-float32 deviation (x2: two independent float32 evaluations) elsewhere.
    ref_dev = dict(zip((str(n) for n in fx['refine_names']), fx['refine_ref_f32_dev']))
    for n, want in zip(fx['refine_names'], fx['refine_norms']):
        got = float(params[str(n)].grad.double().norm())
        tol = max(1e-4, 2.0 * float(ref_dev[str(n)]))
        # (conv biases that only feed an InstanceNorm have an exactly-zero gradient: rounding residue on both sides)
        assert abs(got - float(want)) <= tol * float(want) + 1e-6 * scale, '%s: |g| %.8g vs %.8g' % (n, got, float(want))
    tight = 0
    for k in fx.files:
        if k.startswith('refine_grad_'):
            n = k[len('refine_grad_'):]
            g = params[n].grad.detach().double().cpu()
            want = torch.from_numpy(fx[k]).double()
            e = float((g - want).norm() / want.norm())
            tol = max(1e-4, 2.0 * float(ref_dev[n]))       # (another float32 evaluation of the same ties: same size, not same sign)
            tight += tol <= 2.5e-4
            assert e <= tol, '%s: relative L2 %.3e (tolerance %.1e)' % (k, e, tol)
    assert tight >= 1                        # the last convolution carries the 1e-4 bound (the reference's own float32 run is 1.4e-7 from float64 there, 3.8e-3 on the gate banks)
