"""CPU: the oracle (plain-torch restatement) against fixtures produced by the REFERENCE classes
(tests/golden/make_golden.py, run in the build container).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import detweights, sequence
from oracle.config import OracleConfig
from oracle.eye_net import EyeNet
from oracle.refine_net import CGRUCell, CLSTMCell, CRNNCell, RefineNet

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EYE_JSON = os.path.join(REPO, 'tests', 'golden', 'eye_net.json')


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def eye_cfg():
    # the values of /root/reference/src/configs/eye_net.json that reach the hot path
    return OracleConfig(batch_size=16, weight_decay=0.005, base_learning_rate=0.001)


def test_eyenet_param_contract():
    net = EyeNet(eye_cfg())
    sd = net.state_dict()
    assert len(sd) == 37
    assert sum(p.numel() for p in net.parameters()) == 11398337
    assert 'cnn_layers.layer2.0.downsample.0.weight' in sd and 'fc_to_gaze.2.weight' in sd
    assert float(sd['fc_to_gaze.2.weight'].abs().max()) == 0.0     # eye_net.py:96


def test_eyenet_frame_and_sequence_match_reference(golden_dir):
    fx = load(golden_dir, 'eyenet.npz')
    B, T = int(fx['B']), int(fx['T'])
    batch = detweights.eyenet_batch(B, T, seed=int(fx['seed']), invalid_fraction=float(fx['invalid_fraction']))
    net = detweights.fill_module(EyeNet(eye_cfg()), seed=0)
    with torch.no_grad():
        sub_in = {k: v[:, 0] for k, v in batch.items()}
        out = {}
        net(sub_in, out, side='left')
        net(sub_in, out, side='right')
        for k, v in out.items():
            np.testing.assert_allclose(v.numpy(), fx['frame0_' + k], atol=2e-6, rtol=0)
        seq = sequence.eyenet_sequence(net, batch)
    for k, v in seq.items():
        np.testing.assert_allclose(v.numpy(), fx['seq_' + k], atol=5e-6, rtol=0)
    # the vacuous-parity trap: outputs must not be identically zero
    assert np.abs(fx['seq_left_g_initial']).max() > 0.05


VARIANT_OVERRIDES = {
    'RNN1': dict(eye_net_use_rnn=True, eye_net_rnn_type='RNN', eye_net_rnn_num_cells=1),
    'LSTM1': dict(eye_net_use_rnn=True, eye_net_rnn_type='LSTM', eye_net_rnn_num_cells=1),
    'GRU2': dict(eye_net_use_rnn=True, eye_net_rnn_type='GRU', eye_net_rnn_num_cells=2),
    'LSTM2': dict(eye_net_use_rnn=True, eye_net_rnn_type='LSTM', eye_net_rnn_num_cells=2),
    'STATIC': dict(eye_net_use_rnn=False),
}


def run_per_step(net, batch, T, device=None):
    """The reference's per-step contract: states handed over through previous_output_dict."""
    steps, prev = [], None
    with torch.no_grad():
        for t in range(T):
            si = {k: (v[:, t].to(device) if device else v[:, t]) for k, v in batch.items()}
            so = {}
            net(si, so, side='left', previous_output_dict=prev)
            net(si, so, side='right', previous_output_dict=prev)
            steps.append(so)
            prev = so
    return steps


def compare_with_variant_fixture(fx, name, steps, tol):
    keys = [k for k in fx.files if k.startswith(name + '/')]
    assert keys
    for fk in keys:
        parts = fk.split('/')
        want = fx[fk]
        if len(parts) == 3:                       # LSTM: the state is an (h, c) tuple
            got = torch.stack([s[parts[1]][int(parts[2])] for s in steps], 1)
        else:
            got = torch.stack([s[parts[1]] for s in steps], 1)
        err = float(np.abs(got.cpu().numpy() - want).max())
        assert err < tol, (fk, err)


@pytest.mark.parametrize('name', sorted(VARIANT_OVERRIDES))
def test_eyenet_recurrent_variants_match_reference(golden_dir, name):
    """RNNCell / LSTMCell / stacked cells / static_fc (eye_net.py:58-78): oracle == the reference's EyeNet, per step."""
    fx = np.load(os.path.join(golden_dir, 'eyenet_variants.npz'))
    B, T = int(fx['B']), int(fx['T'])
    batch = detweights.eyenet_batch(B, T, seed=int(fx['seed']), invalid_fraction=float(fx['invalid_fraction']))
    net = detweights.fill_module(EyeNet(OracleConfig(**VARIANT_OVERRIDES[name])), seed=0)
    compare_with_variant_fixture(fx, name, run_per_step(net, batch, T), 2e-6)


def test_eyenet_train_step_matches_reference_eve(golden_dir):
    fx = load(golden_dir, 'eyenet.npz')
    cfg = eye_cfg()
    B, T = int(fx['B']), int(fx['T'])
    batch = detweights.eyenet_batch(B, T, seed=0, invalid_fraction=float(fx['invalid_fraction']))
    net = detweights.fill_module(EyeNet(cfg), seed=0)
    opt = sequence.make_optimizer(net.parameters(), cfg)
    opt.zero_grad()
    out = sequence.eyenet_sequence(net, batch)
    terms = sequence.eyenet_losses(out, batch, cfg)
    for k in ('loss_ang_left_g_initial', 'loss_ang_right_g_initial', 'loss_l1_left_pupil_size',
              'loss_l1_right_pupil_size', 'full_loss'):
        np.testing.assert_allclose(float(terms[k]), float(fx['eve_' + k]), rtol=2e-6)
    terms['full_loss'].backward()
    names = [str(n) for n in fx['grad_names']]
    params = dict(net.named_parameters())
    for n, ref_norm, head in zip(names, fx['grad_norms'], fx['grad_heads']):
        g = params[n].grad.reshape(-1)
        np.testing.assert_allclose(float(g.double().norm()), ref_norm, rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(g[:8].numpy(), head[:min(8, g.numel())], rtol=2e-3, atol=2e-5)
    total = torch.nn.utils.clip_grad_norm_(net.parameters(), cfg.gradient_clip_amount)
    np.testing.assert_allclose(float(total), float(fx['clip_total_norm']), rtol=2e-4)
    opt.step()
    sd = net.state_dict()
    for k in fx.files:
        if k.startswith('updated_'):
            np.testing.assert_allclose(sd[k[len('updated_'):]].reshape(-1)[:16].numpy(), fx[k],
                                       rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize('kind,cls', [('CGRU', CGRUCell), ('CLSTM', CLSTMCell), ('CRNN', CRNNCell)])
def test_cells_match_reference(golden_dir, kind, cls):
    fx = load(golden_dir, 'cells.npz')
    cell = detweights.fill_module(cls(64, 64), seed=3)
    x = torch.from_numpy(fx[kind + '_x']).requires_grad_()
    h = torch.from_numpy(fx[kind + '_h']).requires_grad_()
    if kind == 'CLSTM':
        c = torch.from_numpy(fx[kind + '_c']).requires_grad_()
        hn, cn = cell(x, (h, c))
        (hn.sum() + 0.5 * (cn * cn).sum()).backward()
        np.testing.assert_allclose(cn.detach().numpy(), fx[kind + '_c_new'], atol=2e-6)
        np.testing.assert_allclose(c.grad.numpy(), fx[kind + '_dc'], atol=1e-5)
    else:
        hn = cell(x, h)
        (hn * hn).sum().backward()
    np.testing.assert_allclose(hn.detach().numpy(), fx[kind + '_h_new'], atol=2e-6)
    np.testing.assert_allclose(x.grad.numpy(), fx[kind + '_dx'], atol=2e-5)
    np.testing.assert_allclose(h.grad.numpy(), fx[kind + '_dh'], atol=2e-5)
    h0 = cell(x.detach())
    h0 = h0[0] if isinstance(h0, tuple) else h0
    np.testing.assert_allclose(h0.detach().numpy(), fx[kind + '_h_new_from_none'], atol=2e-6)


@pytest.mark.parametrize('kind', ['CGRU', 'CLSTM', 'CRNN'])
def test_refinenet_matches_reference(golden_dir, kind):
    fx = load(golden_dir, 'refinenet.npz')
    cfg = OracleConfig(load_screen_content=True, refine_net_enabled=True, refine_net_rnn_type=kind)
    rb = detweights.refinenet_batch(int(fx['B']), int(fx['T']), seed=0,
                                    invalid_fraction=float(fx['invalid_fraction']))
    net = detweights.fill_module(RefineNet(cfg), seed=1)
    hf, states = sequence.refinenet_sequence(net, rb['heatmap_initial'], rb['screen_frame'])
    want = fx[kind + '_heatmap_final']
    got = hf.detach().numpy() if kind == 'CGRU' else hf.detach().numpy()[..., ::4, ::4]
    np.testing.assert_allclose(got, want, atol=3e-6)
    assert want.std() > 1e-3          # not the constant-0.5 vacuous case
    last = states[-1]
    np.testing.assert_allclose((last[0] if isinstance(last, tuple) else last).detach().numpy(),
                               fx[kind + '_state_last'], atol=3e-6)
    terms = sequence.refinenet_losses(hf, rb['heatmap_final_gt'], rb['validity'], cfg)
    np.testing.assert_allclose(float(terms['loss_ce_heatmap_final']), float(fx[kind + '_loss_ce']), rtol=3e-6)
    np.testing.assert_allclose(float(terms['loss_mse_heatmap_final']), float(fx[kind + '_loss_mse']), rtol=3e-6)
    terms['full_loss'].backward()
    params = dict(net.named_parameters())
    dead = 0
    for n, ref_norm in zip(fx[kind + '_grad_names'], fx[kind + '_grad_norms']):
        p = params[str(n)]
        if ref_norm < 0:                     # CLSTM dead-output quirk (refine_net.py:168-174)
            assert p.grad is None
            dead += 1
        else:
            np.testing.assert_allclose(float(p.grad.double().norm()), ref_norm, rtol=5e-4, atol=2e-5)  # conv biases feeding an IN have ~0 grad
    assert dead == (2 if kind == 'CLSTM' else 0)


def test_refinenet_without_screen_content(golden_dir):
    fx = load(golden_dir, 'refinenet.npz')
    cfg = OracleConfig(load_screen_content=False, refine_net_enabled=True)
    rb = detweights.refinenet_batch(2, 3, seed=0, invalid_fraction=0.25)
    net = detweights.fill_module(RefineNet(cfg), seed=1)
    out = {'heatmap_initial': rb['heatmap_initial'][:, 0]}
    with torch.no_grad():
        net({}, out)
    np.testing.assert_allclose(out['heatmap_final'].numpy()[..., ::4, ::4],
                               fx['noscreen_heatmap_final'], atol=3e-6)


# ---------------------------------------------------------------------------------------------- EVE sequence harness
from oracle import eve as oracle_eve  # noqa: E402

EVE_CASES = {
    # tag: (training, config overrides on top of configs/refine_net.json with CGRU)
    'c3': (True, {}),
    'eval': (False, {}),
    'joint': (True, dict(eye_net_frozen=False, loss_coeff_PoG_cm_initial=0.002, loss_coeff_g_ang_initial=1.0,
                         loss_coeff_pupil_size=1.0, loss_coeff_heatmap_ce_initial=0.0, loss_coeff_heatmap_mse_final=0.5,
                         loss_coeff_PoG_cm_final=0.01)),
}


def eve_cfg(**over):
    return OracleConfig(os.path.join(REPO, 'configs', 'refine_net.json'), refine_net_rnn_type='CGRU',
                        eye_net_load_pretrained=False, **over)


def grad_norms(module):
    return {n: (-1.0 if p.grad is None else float(p.grad.double().norm())) for n, p in module.named_parameters()}


@pytest.mark.parametrize('tag', sorted(EVE_CASES))
def test_eve_harness_matches_reference_eve(golden_dir, tag):
    """oracle.eve.eve_forward against the reference's own models.eve.EVE run (tests/golden/make_golden_eve.py): label
    synthesis, kappa draw order, gaze geometry, heat-maps, soft-argmax, all 29-31 loss/metric scalars, full_loss and the
    gradients that reach both networks."""
    fx = load(golden_dir, 'eve_harness.npz')
    training, over = EVE_CASES[tag]
    cfg = eve_cfg(**over)
    B, T = int(fx['B']), int(fx['T'])
    batch = detweights.eve_batch(B, T, seed=int(fx['seed']), invalid_fraction=float(fx['invalid_fraction']))
    eye = detweights.fill_module(EyeNet(cfg), seed=0)
    ref = detweights.fill_module(RefineNet(cfg), seed=1)
    if cfg.eye_net_frozen:
        for p in eye.parameters():
            p.requires_grad = False
    np.random.seed(0)                                       # the reference draws kappa from the global numpy RNG
    out, inter, labels = oracle_eve.eve_forward(eye, ref, batch, cfg, training, create_images=(tag == 'eval'))
    if training:
        np.testing.assert_allclose(labels['left_kappa_fake'].numpy(), fx[tag + '_kappa_left'], atol=0, rtol=0)
        np.testing.assert_allclose(labels['right_kappa_fake'].numpy(), fx[tag + '_kappa_right'], atol=0, rtol=0)
    for k in ('g', 'PoG_px_tobii', 'PoG_cm_tobii', 'o'):
        np.testing.assert_allclose(labels[k].numpy(), fx['%s_label_%s' % (tag, k)], atol=1e-4, rtol=1e-6)
    np.testing.assert_allclose(labels['heatmap_final'].numpy()[..., ::4, ::4], fx[tag + '_label_heatmap_final'], atol=1e-6)
    scalars = [k[len(tag) + 1:] for k in fx.files if k.startswith(tag + '_') and fx[k].ndim == 0
               and ('loss' in k or 'metric' in k)]
    assert len(scalars) >= 26 and set(scalars) == set(out.keys()), set(scalars) ^ set(out.keys())
    for k in scalars:
        np.testing.assert_allclose(float(out[k].detach()), float(fx['%s_%s' % (tag, k)]), rtol=2e-4, atol=1e-5, err_msg=k)
    with torch.no_grad():
        for k in ('g_initial', 'PoG_px_initial', 'PoG_cm_initial', 'g_final', 'PoG_px_final', 'PoG_cm_final',
                  'left_g_initial', 'right_g_initial'):
            tol = 2e-2 if 'px' in k else (2e-3 if 'cm' in k else 1e-5)        # px ~ 1e3, cm ~ 1e1, angles ~ 1
            np.testing.assert_allclose(inter[k].numpy(), fx['%s_%s' % (tag, k)], atol=tol, rtol=0, err_msg=k)
        np.testing.assert_allclose(inter['heatmap_initial'].numpy()[..., ::4, ::4], fx[tag + '_heatmap_initial'], atol=2e-5)
        if tag == 'eval':
            np.testing.assert_allclose(inter['history_initial_last'].numpy()[..., ::4, ::4], fx['eval_initial_gaze_history'], atol=2e-5)
            np.testing.assert_allclose(inter['refined_gaze_history'].numpy()[..., ::4, ::4], fx['eval_refined_gaze_history'], atol=2e-5)
    if training:
        out['full_loss'].backward()
        for net, mod in (('eye_net', eye), ('refine_net', ref)):
            got = grad_norms(mod)
            for n, want in zip(fx['%s_%s_grad_names' % (tag, net)], fx['%s_%s_grad_norms' % (tag, net)]):
                if want < 0:
                    assert got[str(n)] < 0, n
                else:
                    assert abs(got[str(n)] - want) <= 2e-3 * want + 1e-6, (n, got[str(n)], want)


# ---------------------------------------------------------------------------------------------------------------------
# The ResNet-18(InstanceNorm) trunk against an INDEPENDENT implementation (transformers' ResNet, BatchNorm swapped for
# InstanceNorm2d; tests/golden/make_golden_trunk.py): the only external arithmetic available for the un-vendored
# torchvision 0.6.1 trunk of /root/reference/src/models/eye_net.py:26,48-50,106.
def trunk_taps_and_grads(cnn, x, proj):
    """Stage outputs, pooled features, fc output of an oracle-shaped ResNet (`conv1, bn1, relu, maxpool, layer1..4,
    avgpool, fc`) and its parameter gradients for the scalar sum(fc * proj)."""
    taps = {}
    hooks = [getattr(cnn, nm).register_forward_hook(lambda m, i, o, nm=nm: taps.__setitem__(nm, o))
             for nm in ('maxpool', 'layer1', 'layer2', 'layer3', 'layer4', 'avgpool')]
    y = cnn(x)
    for h in hooks:
        h.remove()
    (y * proj).sum().backward()
    return taps, y


def check_trunk_case(fx, tag, taps, pooled, fc_out, grads, atol, grad_rtol):
    """taps: name -> N x C x H x W tensors (any subset), grads: torchvision-style name -> tensor."""
    for nm, v in taps.items():
        v = v.detach().double().cpu()
        scale = float(np.sqrt(fx['%s_tap_%s_sqsum' % (tag, nm)] / v.numel()))            # RMS of the stage output
        np.testing.assert_allclose(v.reshape(v.shape[0], -1)[:, :256].float().numpy(), fx['%s_tap_%s_head' % (tag, nm)],
                                   atol=atol * max(scale, 1.0), rtol=0, err_msg=nm)
        np.testing.assert_allclose(float(v.sum()), float(fx['%s_tap_%s_sum' % (tag, nm)]),
                                   atol=atol * scale * v.numel() ** 0.5 * 4, rtol=0, err_msg=nm + ' checksum')
        np.testing.assert_allclose(float((v * v).sum()), float(fx['%s_tap_%s_sqsum' % (tag, nm)]), rtol=20 * atol,
                                   err_msg=nm + ' sum of squares')
    if 'layer4' in taps:
        np.testing.assert_allclose(taps['layer4'].detach().float().cpu().numpy(), fx[tag + '_layer4'], atol=atol * 4, rtol=0)
    np.testing.assert_allclose(pooled.detach().float().cpu().numpy(), fx[tag + '_pooled'], atol=atol, rtol=0)
    np.testing.assert_allclose(fc_out.detach().float().cpu().numpy(), fx[tag + '_fc'], atol=atol * 4, rtol=0)
    for n, ref_norm, head in zip(fx[tag + '_grad_names'], fx[tag + '_grad_norms'], fx[tag + '_grad_heads']):
        g = grads[str(n)].detach().double().cpu().reshape(-1)
        assert abs(float(g.norm()) - ref_norm) <= grad_rtol * ref_norm, '%s: |g| %.6g vs %.6g' % (n, float(g.norm()), ref_norm)
        k = min(64, g.numel())
        np.testing.assert_allclose(g[:k].float().numpy(), head[:k], atol=grad_rtol * ref_norm / g.numel() ** 0.5 * 8 + 1e-7,
                                   rtol=0, err_msg=str(n))


@pytest.mark.parametrize('tag', ['p128', 'p256'])
def test_trunk_restatement_matches_independent_resnet(golden_dir, tag):
    from oracle.resnet_in import BasicBlock, ResNet
    fx = load(golden_dir, 'trunk_independent.npz')
    size, B, T, seed = (int(fx['%s_%s' % (tag, k)]) for k in ('size', 'B', 'T', 'seed'))
    cnn = ResNet(block=BasicBlock, layers=[2, 2, 2, 2], num_classes=128, norm_layer=torch.nn.InstanceNorm2d)
    with torch.no_grad():
        for n, t in cnn.state_dict().items():
            t.copy_(detweights.tensor_for('cnn_layers.' + n, t.shape, 0))
    batch = detweights.eyenet_batch(B, T, size=size, seed=seed)
    x = torch.cat([batch['left_eye_patch'].reshape(B * T, 3, size, size),
                   batch['right_eye_patch'].reshape(B * T, 3, size, size)], dim=0)
    taps, y = trunk_taps_and_grads(cnn, x, torch.from_numpy(fx[tag + '_proj']))
    pooled = taps.pop('avgpool').flatten(1)
    grads = {n: p.grad for n, p in cnn.named_parameters()}
    check_trunk_case(fx, tag, taps, pooled, y, grads, atol=1e-5, grad_rtol=1e-4)
    assert float(np.abs(fx[tag + '_fc']).max()) > 0.1


# ---------------------------------------------------------------------------------------------------------------------
# The rounding-faithful mode of the oracle (oracle/bf16_faithful.py) is tied to the pinned float32 oracle: with its
# rounding points switched off it must reproduce the oracle's outputs and gradients.
def _grad_rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_bf16_faithful_mode_without_rounding_is_the_float32_oracle_eyenet():
    from oracle import bf16_faithful as bf
    cfg = eye_cfg()
    net = detweights.fill_module(EyeNet(cfg), seed=0)
    batch = detweights.eyenet_batch(2, 3, seed=0, invalid_fraction=0.25)
    ref = sequence.eyenet_sequence(net, batch)
    sequence.eyenet_losses(ref, batch, cfg)['full_loss'].backward()
    g_ref = {n: p.grad.clone() for n, p in net.named_parameters()}
    net.zero_grad()
    with bf.rounding(False):
        out = bf.eyenet_sequence(net, batch)
        sequence.eyenet_losses(out, batch, cfg)['full_loss'].backward()
    for k in ref:
        assert float((ref[k] - out[k]).abs().max()) < 1e-5, k        # float32 noise: the stem pools before it normalises
    for n, p in net.named_parameters():
        assert _grad_rel(p.grad, g_ref[n]) < 1e-3, n
    # and with rounding on it is a different (bf16-sized) computation: the mode does something
    with torch.no_grad():
        out16 = bf.eyenet_sequence(net, batch)
    dev = float((out16['left_g_initial'] - ref['left_g_initial']).abs().max())
    assert 1e-4 < dev < 0.1, dev


def test_bf16_faithful_mode_without_rounding_is_the_float32_oracle_refinenet():
    from oracle import bf16_faithful as bf
    cfg = OracleConfig(load_screen_content=True, refine_net_enabled=True, refine_net_rnn_type='CGRU')
    net = detweights.fill_module(RefineNet(cfg), seed=1)
    rb = detweights.refinenet_batch(2, 3, seed=0, invalid_fraction=0.25)
    hf, st = sequence.refinenet_sequence(net, rb['heatmap_initial'], rb['screen_frame'])
    sequence.refinenet_losses(hf, rb['heatmap_final_gt'], rb['validity'], cfg)['full_loss'].backward()
    g_ref = {n: p.grad.clone() for n, p in net.named_parameters()}
    net.zero_grad()
    with bf.rounding(False):
        hf2, st2 = bf.refinenet_sequence(net, rb['heatmap_initial'], rb['screen_frame'])
        sequence.refinenet_losses(hf2, rb['heatmap_final_gt'], rb['validity'], cfg)['full_loss'].backward()
    assert float((hf - hf2).abs().max()) < 1e-6
    assert float((st[-1] - st2[-1]).abs().max()) < 1e-6
    for n, p in net.named_parameters():
        if float(g_ref[n].norm()) < 1e-6:        # biases in front of an InstanceNorm: exactly zero gradient, float noise
            assert float(p.grad.norm()) < 1e-6, n
        else:
            assert _grad_rel(p.grad, g_ref[n]) < 1e-4, n


# ---------------------------------------------------------------------------------------------------------------------
# LR schedule: the oracle's restatement and the product's eve_amd.schedule against values produced by the
# reference's own learning_rate_schedule driving torch's LambdaLR (tests/golden/make_golden_schedule.py).
SCHEDULE_CASES = ['none', 'eye_net_json', 'warmup_exponential', 'warmup_cyclic']


def _schedule_cfg(fx, name, cfg):
    for k in ('batch_size', 'base_learning_rate', 'num_warmup_epochs', 'lr_decay_strategy', 'lr_decay_factor', 'lr_decay_epoch_interval'):
        key = '%s_cfg_%s' % (name, k)
        if key in fx.files:
            v = fx[key].item()
            setattr(cfg, k, v) if not hasattr(cfg, 'override') else cfg.override(k, v)
    return cfg, int(fx[name + '_epoch_len'])


@pytest.mark.parametrize('name', SCHEDULE_CASES)
def test_lr_schedule_matches_reference(golden_dir, name):
    fx = load(golden_dir, 'lr_schedule.npz')
    cfg, epoch_len = _schedule_cfg(fx, name, OracleConfig())
    want_s, want_e = fx[name + '_schedule'], fx[name + '_effective']
    got_s = np.array([sequence.lr_schedule(cfg, epoch_len, s) for s in range(len(want_s))])
    got_e = np.array([sequence.lr_used_by_step(cfg, epoch_len, s) for s in range(len(want_e))])
    np.testing.assert_allclose(got_s, want_s, rtol=1e-12)
    np.testing.assert_allclose(got_e, want_e, rtol=1e-12)
    assert want_e[0] == pytest.approx(cfg.learning_rate * want_s[0])          # the LambdaLR quirk is in the fixture


@pytest.mark.parametrize('name', SCHEDULE_CASES)
def test_product_lr_schedule_matches_reference(golden_dir, name):
    import eve_amd
    from eve_amd import schedule
    fx = load(golden_dir, 'lr_schedule.npz')
    cfg, epoch_len = _schedule_cfg(fx, name, eve_amd.reset_standalone_config())
    want_s, want_e = fx[name + '_schedule'], fx[name + '_effective']
    np.testing.assert_allclose([schedule.learning_rate_schedule(cfg, epoch_len, s) for s in range(len(want_s))], want_s, rtol=1e-12)
    np.testing.assert_allclose([schedule.effective_learning_rate(cfg, epoch_len, s) for s in range(len(want_e))], want_e, rtol=1e-12)
    np.testing.assert_allclose([schedule.effective_learning_rate(cfg, epoch_len, s, reference_quirk=False) for s in range(len(want_s))],
                               want_s, rtol=1e-12)
    eve_amd.reset_standalone_config()


def test_oracle_float64_gradients_match_the_reference_float64_full_tensors(golden_dir):
    """tests/golden/grads_f64.npz: FULL gradient tensors from the reference's own EVE / RefineNet evaluated in float64
    (make_golden_grads.py).  The oracle in float64 must reproduce them to float64 accuracy -- every stored tensor
    element-wise and every parameter's norm -- for the EyeNet train step and the RefineNet / CGRU sequence."""
    fx = load(golden_dir, 'grads_f64.npz')
    cfg = eye_cfg()
    batch = detweights.eyenet_batch(2, 3, seed=0, invalid_fraction=0.25)
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    net = detweights.fill_module(EyeNet(cfg), seed=0).double()
    terms = sequence.eyenet_losses(sequence.eyenet_sequence(net, b64), b64, cfg)
    np.testing.assert_allclose(float(terms['full_loss'].detach()), float(fx['eye_full_loss_f64']), rtol=1e-10)
    terms['full_loss'].backward()
    params = dict(net.named_parameters())
    for n, want in zip(fx['eye_names'], fx['eye_norms']):
        np.testing.assert_allclose(float(params[str(n)].grad.norm()), float(want), rtol=1e-8, err_msg=str(n))
    checked = 0
    for k in fx.files:
        if k.startswith('eye_grad_') or k.startswith('eye_block_'):
            g = params[k.split('_', 2)[2]].grad
            g = g[:64, :64] if k.startswith('eye_block_') else g
            want = torch.from_numpy(fx[k]).double()
            assert float((g - want).norm() / want.norm()) < 1e-6, k          # (the fixture stores float32 of the float64 values)
            checked += 1
    assert checked == 5 and float(fx['eye_ref_f32_vs_f64_worst']) > 1e-4       # why the fixture is float64: see its generator
    ocfg = OracleConfig(load_screen_content=True, refine_net_enabled=True, refine_net_rnn_type='CGRU')
    rb = detweights.refinenet_batch(2, 3, seed=0, invalid_fraction=0.25)
    rnet = detweights.fill_module(RefineNet(ocfg), seed=1).double()
    hf, _ = sequence.refinenet_sequence(rnet, rb['heatmap_initial'].double(), rb['screen_frame'].double())
    rterms = sequence.refinenet_losses(hf, rb['heatmap_final_gt'].double(), rb['validity'], ocfg)
    np.testing.assert_allclose(float(rterms['loss_ce_heatmap_final'].detach()), float(fx['refine_loss_ce']), rtol=1e-10)
    rterms['loss_ce_heatmap_final'].backward()
    rparams = dict(rnet.named_parameters())
    for n, want in zip(fx['refine_names'], fx['refine_norms']):
        np.testing.assert_allclose(float(rparams[str(n)].grad.norm()), float(want), rtol=1e-7, atol=1e-12, err_msg=str(n))
    for k in fx.files:
        if k.startswith('refine_grad_'):
            want = torch.from_numpy(fx[k]).double()
            assert float((rparams[k[len('refine_grad_'):]].grad - want).norm() / want.norm()) < 1e-6, k
