"""GPU parity proper: eve_amd.EyeNet (HIP kernels through the C ABI) against
  (a) the golden fixtures produced by the reference classes (tests/golden/eyenet.npz), and
  (b) the CPU oracle on the same seeded inputs.
Tolerance from BASELINE.json north_star: gaze angles within 1e-4 rad in the float32 instantiation;
the bf16 deviation is measured and bounded separately (it is a precision mode, not a parity claim)."""
import os

import numpy as np
import pytest
import torch

from oracle import detweights, sequence
from oracle.config import OracleConfig

pytestmark = pytest.mark.gpu

GAZE_TOL = 1e-4      # rad, north_star
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def eye_cfg():
    return OracleConfig(batch_size=16, weight_decay=0.005, base_learning_rate=0.001)


def make_net(dtype):
    import eve_amd
    cfg = eve_amd.reset_standalone_config()
    cfg.import_dict({'batch_size': 16, 'weight_decay': 0.005, 'base_learning_rate': 0.001})
    net = eve_amd.EyeNet()
    net.compute_dtype = dtype
    detweights.fill_module(net, seed=0)
    return net.cuda()


def to_dev(batch):
    return {k: v.cuda() for k, v in batch.items()}


def test_state_dict_contract_matches_oracle():
    from oracle.eye_net import EyeNet as OracleEyeNet
    net = make_net(torch.float32)
    ref = OracleEyeNet(eye_cfg())
    a, b = net.state_dict(), ref.state_dict()
    assert list(a.keys()) == list(b.keys())
    assert all(tuple(a[k].shape) == tuple(b[k].shape) for k in a)


def test_single_frame_matches_reference_golden():
    fx = np.load(os.path.join(GOLDEN, 'eyenet.npz'))
    B, T = int(fx['B']), int(fx['T'])
    batch = to_dev(detweights.eyenet_batch(B, T, seed=0, invalid_fraction=float(fx['invalid_fraction'])))
    net = make_net(torch.float32)
    sub_in = {k: v[:, 0] for k, v in batch.items()}
    out = {}
    with torch.no_grad():
        net(sub_in, out, side='left')
        net(sub_in, out, side='right')
    for k, v in out.items():
        err = np.abs(v.cpu().numpy() - fx['frame0_' + k]).max()
        assert err < GAZE_TOL, '%s: %.3e' % (k, err)
    assert np.abs(fx['frame0_left_g_initial']).max() > 0.05      # not the zero-init vacuous case


def test_per_step_contract_and_sequence_match_reference_golden():
    fx = np.load(os.path.join(GOLDEN, 'eyenet.npz'))
    B, T = int(fx['B']), int(fx['T'])
    batch = to_dev(detweights.eyenet_batch(B, T, seed=0, invalid_fraction=float(fx['invalid_fraction'])))
    net = make_net(torch.float32)
    with torch.no_grad():
        steps = []                       # exactly how src/models/eve.py:91-111 drives the module
        for t in range(T):
            sub_in = {k: v[:, t] for k, v in batch.items()}
            sub_out = {}
            prev = steps[-1] if steps else None
            net(sub_in, sub_out, side='left', previous_output_dict=prev)
            net(sub_in, sub_out, side='right', previous_output_dict=prev)
            steps.append(sub_out)
        stepped = {k: torch.stack([s[k] for s in steps], dim=1) for k in steps[0]}
        folded = net.forward_sequence(batch)
    for k in stepped:
        want = fx['seq_' + k]
        assert np.abs(stepped[k].cpu().numpy() - want).max() < GAZE_TOL, k + ' (per-step)'
        assert np.abs(folded[k].cpu().numpy() - want).max() < GAZE_TOL, k + ' (folded sequence)'


def test_train_step_matches_reference_eve_golden():
    """losses, masking, full_loss, gradients, clip norm and the Adam update of the reference's own EVE.forward +
    training loop (fixture; /root/reference/src/train.py:49-55, src/core/training.py:492-502), reproduced by the
    product train step: HIP forward / backward with weight gradients written in place into the flat gradient buffer,
    eve_sumsq (clip norm) and eve_adam_step on the flat parameter buffer (eve_amd.train.Trainer)."""
    from eve_amd import train
    fx = np.load(os.path.join(GOLDEN, 'eyenet.npz'))
    cfg = eye_cfg()
    B, T = int(fx['B']), int(fx['T'])
    batch = to_dev(detweights.eyenet_batch(B, T, seed=0, invalid_fraction=float(fx['invalid_fraction'])))
    net = make_net(torch.float32)
    trainer = train.Trainer([net], net.config, lambda b: sequence.eyenet_losses(net.forward_sequence(b), b, cfg))
    assert all(getattr(p, '_eve_flat_grad', False) for p in net.parameters())
    terms = trainer._forward_backward(batch)
    for k in ('loss_ang_left_g_initial', 'loss_ang_right_g_initial', 'loss_l1_left_pupil_size',
              'loss_l1_right_pupil_size', 'full_loss'):
        np.testing.assert_allclose(float(terms[k].detach()), float(fx['eve_' + k]), rtol=2e-5)
    params = dict(net.named_parameters())
    for n, ref_norm, head in zip(fx['grad_names'], fx['grad_norms'], fx['grad_heads']):
        g = params[str(n)].grad.reshape(-1)
        got = float(g.double().norm())
        # float32 rounding noise of the problem itself (see test_float32_gradient_deviation_is_float_rounding_noise)
        assert abs(got - ref_norm) <= 1e-2 * ref_norm + 1e-5, '%s: |g| %.6g vs %.6g' % (n, got, ref_norm)
    trainer._update(1.0)
    np.testing.assert_allclose(float(trainer.sumsq.sqrt()), float(fx['clip_total_norm']), rtol=5e-3)
    sd = net.state_dict()
    for k in fx.files:
        if k.startswith('updated_'):
            np.testing.assert_allclose(sd[k[len('updated_'):]].reshape(-1)[:16].cpu().numpy(), fx[k],
                                       rtol=1e-3, atol=2e-5)


def test_sequence_matches_cpu_oracle_other_shape_and_grads():
    """A second shape (B=3, T=5, ragged validity) against the oracle run here on the host."""
    from oracle.eye_net import EyeNet as OracleEyeNet
    cfg = eye_cfg()
    batch = detweights.eyenet_batch(3, 5, seed=4, invalid_fraction=0.3)
    ref = detweights.fill_module(OracleEyeNet(cfg), seed=0)
    rout = sequence.eyenet_sequence(ref, batch)
    sequence.eyenet_losses(rout, batch, cfg)['full_loss'].backward()
    net = make_net(torch.float32)
    dbatch = to_dev(batch)
    out = net.forward_sequence(dbatch)
    for k in rout:
        assert float((out[k].cpu() - rout[k]).abs().max()) < GAZE_TOL, k
    sequence.eyenet_losses(out, dbatch, cfg)['full_loss'].backward()
    rp = dict(ref.named_parameters())
    for n, p in net.named_parameters():
        a, b = p.grad.cpu().double(), rp[n].grad.double()
        # relative L2: single elements carry fp32 ReLU / max-pool mask-flip noise (also CPU-vs-CPU)
        err = float((a - b).norm() / (b.norm() + 1e-12))
        assert err <= 2e-2 or float((a - b).abs().max()) < 1e-5, '%s: rel L2 %.3e' % (n, err)


@pytest.mark.parametrize('B,T', [(3, 5), (4, 30), (2, 70)])
def test_fused_losses_match_oracle_values_and_gradients(B, T):
    """eve_eye_losses (one launch: four masked terms, weighted sum, gradients) == the oracle's torch losses."""
    from eve_amd import losses as hip_losses
    cfg = eye_cfg()
    batch = detweights.eyenet_batch(B, T, seed=7, invalid_fraction=0.4)
    batch['left_g_tobii_validity'][0, 1:] = False          # a clip with a single valid step (denominator rule)
    batch['right_p_validity'][1] = False                   # and one with none
    g = torch.Generator().manual_seed(3)
    out = {'left_g_initial': torch.randn((B, T, 2), generator=g) * 0.4, 'right_g_initial': torch.randn((B, T, 2), generator=g) * 0.4,
           'left_pupil_size': torch.rand((B, T), generator=g) * 4, 'right_pupil_size': torch.rand((B, T), generator=g) * 4}
    out['left_g_initial'][0, 0] = batch['left_g_tobii'][0, 0]          # identical vectors: clamp boundary, zero gradient
    ref_in = {k: v.clone().requires_grad_(True) for k, v in out.items()}
    ref_terms = sequence.eyenet_losses(ref_in, batch, cfg)
    ref_terms['full_loss'].backward()
    dev_in = {k: v.clone().cuda().requires_grad_(True) for k, v in out.items()}
    terms = hip_losses.eyenet_loss_terms(dev_in, to_dev(batch), cfg)
    assert 'EyeLossesFn' in type(terms['full_loss'].grad_fn).__name__          # the fused node, not torch ops
    for k in ref_terms:
        np.testing.assert_allclose(float(terms[k]), float(ref_terms[k]), rtol=2e-5, atol=1e-6, err_msg=k)
    terms['full_loss'].backward()
    for k in out:
        a, b = dev_in[k].grad.cpu(), ref_in[k].grad
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-7, k


from test_host_logic import VARIANTS, _variant_id, check_eyenet_variant  # noqa: E402


@pytest.mark.parametrize('over', VARIANTS, ids=_variant_id)
def test_recurrent_variants_match_oracle(over):
    """RNN / LSTM / stacked cells / static_fc (eye_net.py:58-78) through the HIP kernels, fp32: outputs, states handed over
    through the dicts ((h, c) tuples for LSTM) and parameter gradients against the oracle."""
    check_eyenet_variant(over, to_device=lambda t: t.cuda(), tol=GAZE_TOL, grad_tol=1e-2)


@pytest.mark.parametrize('name', ['RNN1', 'LSTM1', 'GRU2', 'LSTM2', 'STATIC'])
def test_recurrent_variants_match_reference_golden(name):
    """The same variants against vectors produced by the reference's own EyeNet (tests/golden/eyenet_variants.npz)."""
    import eve_amd
    from test_oracle_golden import VARIANT_OVERRIDES, compare_with_variant_fixture, run_per_step
    fx = np.load(os.path.join(GOLDEN, 'eyenet_variants.npz'))
    B, T = int(fx['B']), int(fx['T'])
    batch = detweights.eyenet_batch(B, T, seed=int(fx['seed']), invalid_fraction=float(fx['invalid_fraction']))
    cfg = eve_amd.reset_standalone_config()
    cfg.import_dict(VARIANT_OVERRIDES[name])
    net = eve_amd.EyeNet()
    net.compute_dtype = torch.float32
    detweights.fill_module(net, seed=0)
    compare_with_variant_fixture(fx, name, run_per_step(net.cuda(), batch, T, device='cuda'), GAZE_TOL)


def test_bf16_deviation_is_bounded_and_reported():
    """Sanity bound only: bf16 against the REFERENCE's float32 fixture differs by the precision of bf16 storage (a few
    1e-2 rad on untrained weights), which says nothing about kernel correctness.  The parity statement for the bf16
    kernels is tests/test_gpu_bf16_parity.py: every stage against the rounding-faithful oracle on the same inputs
    (forward <= 3e-3, backward <= 1e-2 relative L2) and the end-to-end deviation inside the rounding-noise envelope."""
    fx = np.load(os.path.join(GOLDEN, 'eyenet.npz'))
    B, T = int(fx['B']), int(fx['T'])
    batch = to_dev(detweights.eyenet_batch(B, T, seed=0, invalid_fraction=float(fx['invalid_fraction'])))
    net = make_net(torch.bfloat16)
    with torch.no_grad():
        out = net.forward_sequence(batch)
    dev = max(np.abs(out[s + '_g_initial'].cpu().numpy() - fx['seq_' + s + '_g_initial']).max()
              for s in ('left', 'right'))
    print('bf16 max gaze deviation vs reference fp32: %.4e rad' % dev)
    assert dev < 0.08, 'bf16 gaze deviation %.3e rad' % dev


def test_256x256_patches_match_oracle_and_train_in_bf16():
    """BASELINE configs[4]'s geometry: 256 x 256 eye patches (conv work x4, 128 x 128 stem output, 8 x 8 final planes).
    float32 against the CPU oracle within the gaze tolerance, with gradients; then the bf16 instantiation (the stem goes
    through the stand-alone 7x7 kernel + fused IN/ReLU/pool: the one-wave-per-image fused stem is 128 pixels wide)."""
    from oracle.eye_net import EyeNet as OracleEyeNet
    cfg = eye_cfg()
    batch = detweights.eyenet_batch(1, 3, size=256, seed=9, invalid_fraction=0.2)
    ref = detweights.fill_module(OracleEyeNet(cfg), seed=0)
    rout = sequence.eyenet_sequence(ref, batch)
    sequence.eyenet_losses(rout, batch, cfg)['full_loss'].backward()
    net = make_net(torch.float32)
    dbatch = to_dev(batch)
    out = net.forward_sequence(dbatch)
    for k in rout:
        assert float((out[k].cpu() - rout[k]).abs().max()) < GAZE_TOL, k
    sequence.eyenet_losses(out, dbatch, cfg)['full_loss'].backward()
    rp = dict(ref.named_parameters())
    for n, p in net.named_parameters():
        a, b = p.grad.cpu().double(), rp[n].grad.double()
        err = float((a - b).norm() / (b.norm() + 1e-12))
        assert err <= 2e-2 or float((a - b).abs().max()) < 1e-5, '%s: rel L2 %.3e' % (n, err)
    net16 = make_net(torch.bfloat16)
    out16 = net16.forward_sequence(dbatch)
    sequence.eyenet_losses(out16, dbatch, cfg)['full_loss'].backward()
    dev = max(float((out16[s + '_g_initial'].float().cpu() - rout[s + '_g_initial']).abs().max()) for s in ('left', 'right'))
    print('256x256 bf16 max gaze deviation vs oracle fp32: %.4e rad' % dev)
    assert dev < 0.08
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net16.parameters())


def test_uint8_clips_through_the_prefetcher_equal_the_float_path():
    """Input pipeline: uint8 [B,T,H,W,C] clips staged through pinned memory on the copy stream
    (data.DevicePrefetcher) and normalised on the device give bit-identical EyeNet outputs to the reference-style float
    batches, in float32 and in bf16 (where the frames go straight into the stem's packed layout)."""
    from eve_amd import data
    from oracle import frames as oframes
    g = np.random.Generator(np.random.PCG64(21))
    host = []
    for _ in range(3):
        b = {s + '_eye_patch': torch.from_numpy(g.integers(0, 256, size=(2, 3, 128, 128, 3), dtype=np.uint8)) for s in ('left', 'right')}
        b.update({s + '_h': torch.from_numpy(g.normal(0, 0.1, size=(2, 3, 2)).astype(np.float32)) for s in ('left', 'right')})
        b['tag'] = 'clip'
        host.append(b)
    for dt in (torch.float32, torch.bfloat16):
        net = make_net(dt)
        seen = 0
        with torch.no_grad():
            for dev_batch, hb in zip(data.DevicePrefetcher(host), host):
                assert dev_batch['tag'] == 'clip' and dev_batch['left_eye_patch'].is_cuda and dev_batch['left_eye_patch'].dtype == torch.uint8
                ref = dict(dev_batch)
                for s in ('left', 'right'):
                    f = oframes.preprocess_frames(hb[s + '_eye_patch'].numpy().reshape(6, 128, 128, 3))
                    ref[s + '_eye_patch'] = torch.from_numpy(f).view(2, 3, 3, 128, 128).cuda()
                a, b = net.forward_sequence(dev_batch), net.forward_sequence(ref)
                for k in ('left_g_initial', 'right_g_initial', 'left_pupil_size'):
                    assert torch.equal(a[k], b[k]), (dt, k)
                seen += 1
        assert seen == 3


def test_full_size_forward_is_batch_invariant_and_deterministic():
    """BASELINE configs[1] at full size (B=32 clips x T=30, bf16), through properties that do not need the oracle:
    clips are independent units (InstanceNorm is per image, the GRU per sequence), so every clip's outputs are
    bit-identical whether it runs in the batch of 32 or in a batch of 8, and a repeated forward is bit-identical;
    the data-parallel identity grad(batch) = mean of grad(halves) holds to bf16 / atomic-order noise."""
    cfg = eye_cfg()
    small = detweights.eyenet_batch(8, 30, seed=12, invalid_fraction=0.1)
    g = torch.Generator().manual_seed(0)
    full = {}
    for k, v in small.items():                       # 32 distinct clips: 4 perturbed copies of the 8 generated ones
        reps = [v] + [(v + 0.05 * torch.randn(v.shape, generator=g)).clamp(-1, 1) if v.dtype == torch.float32 and 'patch' in k
                      else v for _ in range(3)]
        full[k] = torch.cat(reps, dim=0)
    full = to_dev(full)
    net = make_net(torch.bfloat16)
    with torch.no_grad():
        a = net.forward_sequence(full)
        b = net.forward_sequence(full)
        for k in ('left_g_initial', 'right_g_initial', 'left_pupil_size', 'right_pupil_size'):
            assert torch.equal(a[k], b[k]), 'not deterministic: ' + k
        for i in range(0, 32, 8):
            part = net.forward_sequence({k: v[i:i + 8].contiguous() for k, v in full.items()})
            for k in ('left_g_initial', 'right_g_initial', 'left_pupil_size', 'right_pupil_size'):
                assert torch.equal(part[k], a[k][i:i + 8]), 'clip outputs depend on the batch: %s, clips %d..' % (k, i)
    assert float(a['left_g_initial'].abs().max()) > 0.05

    def grads(batch):
        net.zero_grad(set_to_none=True)
        sequence.eyenet_losses(net.forward_sequence(batch), batch, cfg)['full_loss'].backward()
        return torch.cat([p.grad.detach().float().reshape(-1) for p in net.parameters()])
    whole = grads(full)
    halves = 0.5 * (grads({k: v[:16].contiguous() for k, v in full.items()}) + grads({k: v[16:].contiguous() for k, v in full.items()}))
    rel = float((whole - halves).norm() / whole.norm())
    assert rel < 2e-2, rel


def test_long_sequence_t120():
    """BASELINE configs[4]'s sequence length: T = 120 frames through the GRU scan (one launch for the whole clip, hidden
    state carried on-chip) -- float32 forward against the oracle's per-frame loop, and a bf16 train step with finite
    gradients at that length."""
    from oracle.eye_net import EyeNet as OracleEyeNet
    cfg = eye_cfg()
    batch = detweights.eyenet_batch(1, 120, seed=31, invalid_fraction=0.1)
    ref = detweights.fill_module(OracleEyeNet(cfg), seed=0)
    with torch.no_grad():
        rout = sequence.eyenet_sequence(ref, batch)
        out = make_net(torch.float32).forward_sequence(to_dev(batch))
    for k in ('left_g_initial', 'right_g_initial', 'left_pupil_size', 'right_eye_rnn_states_0'):
        assert float((out[k].cpu() - rout[k]).abs().max()) < GAZE_TOL, k
    # the recurrence matters: the last frame's state differs from a fresh start
    assert float((rout['left_eye_rnn_states_0'][:, -1] - rout['left_eye_rnn_states_0'][:, 0]).abs().max()) > 1e-2
    net16 = make_net(torch.bfloat16)
    dbatch = to_dev(batch)
    o16 = net16.forward_sequence(dbatch)
    sequence.eyenet_losses(o16, dbatch, cfg)['full_loss'].backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net16.parameters())
    dev = max(float((o16[s + '_g_initial'].float().cpu() - rout[s + '_g_initial']).abs().max()) for s in ('left', 'right'))
    assert dev < 0.08, dev


@pytest.mark.parametrize('tag', ['p128', 'p256'])
def test_trunk_matches_independent_resnet_fixture(tag):
    """The HIP ResNet-18(InstanceNorm) trunk, float32, against vectors from an INDEPENDENT implementation of the
    published architecture (transformers' ResNet with InstanceNorm2d, tests/golden/make_golden_trunk.py) -- the external
    pin for the un-vendored torchvision trunk of /root/reference/src/models/eye_net.py:26,48-50,106: layer4 output,
    pooled features, fc output and every trunk parameter gradient."""
    from eve_amd import ops
    from eve_amd.kernels import pad_channels
    from test_oracle_golden import check_trunk_case
    fx = np.load(os.path.join(GOLDEN, 'trunk_independent.npz'))
    size, B, T, seed = (int(fx['%s_%s' % (tag, k)]) for k in ('size', 'B', 'T', 'seed'))
    batch = detweights.eyenet_batch(B, T, size=size, seed=seed)
    x = torch.cat([batch['left_eye_patch'].reshape(B * T, 3, size, size),
                   batch['right_eye_patch'].reshape(B * T, 3, size, size)], dim=0).cuda()
    net = make_net(torch.float32)
    P = net._get_packs()
    y4 = net._trunk_layers(ops.ToNHWCFn.apply(x, torch.float32, pad_channels(3, torch.float32)), P)
    pooled = ops.cast(ops.AvgPoolFn.apply(y4), torch.float32)
    fc = ops.linear(pooled, net.cnn_layers.fc.weight, net.cnn_layers.fc.bias, P['fc'])
    (fc * torch.from_numpy(fx[tag + '_proj']).cuda()).sum().backward()
    grads = {n: p.grad for n, p in net.cnn_layers.named_parameters()}
    check_trunk_case(fx, tag, {'layer4': y4.permute(0, 3, 1, 2)}, pooled, fc, grads, atol=1e-4, grad_rtol=1e-2)


def test_float32_gradients_match_the_reference_float64_full_tensors():
    """tests/golden/grads_f64.npz (the reference's EVE + EyeNet evaluated in float64, make_golden_grads.py): the float32 HIP
    path's gradients -- the FULL tensors of the stem convolution, a layer-1 convolution, the GRU's hidden weights, the gaze
    head, a 64 x 64 channel block of the last layer-4 convolution -- within 1e-4 relative L2, and every parameter's norm
    within 1e-4.  (The reference's own float32 CPU evaluation is 5.5e-4 away from these on the first layers: the fixture
    records that figure; a float32 fixture could not carry this tolerance.)"""
    fx = np.load(os.path.join(GOLDEN, 'grads_f64.npz'))
    cfg = eye_cfg()
    batch = to_dev(detweights.eyenet_batch(2, 3, seed=0, invalid_fraction=0.25))
    net = make_net(torch.float32)
    terms = sequence.eyenet_losses(net.forward_sequence(batch), batch, cfg)
    np.testing.assert_allclose(float(terms['full_loss'].detach()), float(fx['eye_full_loss_f64']), rtol=2e-6)
    terms['full_loss'].backward()
    params = dict(net.named_parameters())
    for n, want in zip(fx['eye_names'], fx['eye_norms']):
        got = float(params[str(n)].grad.double().norm())
        assert abs(got - float(want)) <= 1e-4 * float(want) + 1e-9, '%s: |g| %.8g vs %.8g' % (n, got, float(want))
    assert float(fx['eye_ref_f32_dev'].max()) > 4e-4            # (the reference's own float32 run does not meet that bound)
    worst = 0.0
    for k in fx.files:
        if k.startswith('eye_grad_') or k.startswith('eye_block_'):
            g = params[k.split('_', 2)[2]].grad.detach().double().cpu()
            g = g[:64, :64] if k.startswith('eye_block_') else g
            want = torch.from_numpy(fx[k]).double()
            e = float((g - want).norm() / want.norm())
            worst = max(worst, e)
            assert e <= 1e-4, '%s: relative L2 %.3e' % (k, e)
    print('worst full-tensor gradient deviation from the float64 reference: %.2e (reference float32: %.2e)' % (
        worst, float(fx['eye_ref_f32_vs_f64_worst'])))


@pytest.mark.parametrize('dtype,B,T', [(torch.float32, 3, 5), (torch.bfloat16, 2, 7), (torch.float32, 1, 1)])
def test_tail_and_losses_as_one_node_equal_the_per_layer_path(dtype, B, T):
    """Round 4: EyeNet.loss_terms_sequence runs the tail and the losses as ONE autograd node (ops.EyeTailLossFn: every layer one
    direct launch of the float32 tail kernels, gradients written into the trainer's flat buffer by one batched launch) -- against
    the per-layer composition it replaces (tail_loss_node = False: LinearFn x 8, GRUScanFn, EyeLossesFn and autograd's glue) on
    the same batch and weights: the five loss terms, the predictions, and the WHOLE flat gradient (trunk included: d(features)
    enters it).  The tail is float32 in both; same products, sums in the same order per output element except the weight
    gradients' float atomics.  Then the product trainer's step on the golden fixture (test_train_step_matches_reference_eve_golden
    through train.eyenet_trainer, which takes the node)."""
    from eve_amd import train
    batch = to_dev(detweights.eyenet_batch(B, T, seed=5, invalid_fraction=0.25))
    res = {}
    for node in (False, True):
        net = make_net(dtype)
        with torch.no_grad():                      # the reference zero-initialises the last gaze layer: give it a signal
            g = torch.Generator().manual_seed(12)
            net.fc_to_gaze[2].weight.copy_(0.05 * torch.randn(net.fc_to_gaze[2].weight.shape, generator=g).cuda())
        net.tail_loss_node = node
        tr = train.eyenet_trainer(net, net.config)
        terms = tr._forward_backward(batch)
        torch.cuda.synchronize()
        assert net.last_tail_path == ('node' if node else 'layers')
        res[node] = ({k: v.detach().clone() for k, v in terms.items()}, tr.fp.grad.clone(),
                     {n: p.grad.detach().clone() for n, p in net.named_parameters()})
    (ta, ga, pa), (tb, gb, pb) = res[False], res[True]
    assert set(ta) == set(tb)
    for k in ta:
        d = float((ta[k].float() - tb[k].float()).abs().max())
        assert d <= 2e-6 * max(1.0, float(ta[k].abs().max())), (k, d)
    assert torch.isfinite(gb).all()
    tail_names = [n for n in pa if not n.startswith('cnn_layers.') or n.startswith('cnn_layers.fc')]
    assert len(tail_names) == 17
    for n in tail_names:
        assert float((pa[n] - pb[n]).norm()) <= 2e-5 * float(pa[n].norm()) + 1e-7, (n, float((pa[n] - pb[n]).norm()), float(pa[n].norm()))
    # the trunk's gradients follow from d(features): float32 reproduces to rounding, bf16 to its own run-to-run atomics noise
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert float((ga - gb).norm()) <= tol * float(ga.norm()), float((ga - gb).norm() / ga.norm())


def test_eyenet_trainer_takes_the_one_node_tail_and_matches_the_reference_step():
    from eve_amd import train
    fx = np.load(os.path.join(GOLDEN, 'eyenet.npz'))
    B, T = int(fx['B']), int(fx['T'])
    batch = to_dev(detweights.eyenet_batch(B, T, seed=0, invalid_fraction=float(fx['invalid_fraction'])))
    net = make_net(torch.float32)
    trainer = train.eyenet_trainer(net, net.config)
    terms = trainer._forward_backward(batch)
    assert net.last_tail_path == 'node'
    for k in ('loss_ang_left_g_initial', 'loss_ang_right_g_initial', 'loss_l1_left_pupil_size', 'loss_l1_right_pupil_size', 'full_loss'):
        np.testing.assert_allclose(float(terms[k].detach()), float(fx['eve_' + k]), rtol=2e-5)
    params = dict(net.named_parameters())
    for n, ref_norm in zip(fx['grad_names'], fx['grad_norms']):
        got = float(params[str(n)].grad.reshape(-1).double().norm())
        assert abs(got - ref_norm) <= 1e-2 * ref_norm + 1e-5, '%s: |g| %.6g vs %.6g' % (n, got, ref_norm)
    trainer._update(1.0)
    np.testing.assert_allclose(float(trainer.sumsq.sqrt()), float(fx['clip_total_norm']), rtol=5e-3)
    sd = net.state_dict()
    for k in fx.files:
        if k.startswith('updated_'):
            np.testing.assert_allclose(sd[k[len('updated_'):]].reshape(-1)[:16].cpu().numpy(), fx[k], rtol=1e-3, atol=2e-5)
