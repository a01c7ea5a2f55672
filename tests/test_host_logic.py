"""CPU: host logic of eve_amd (autograd wiring, packing/padding, drop-in dict contract, state_dict keys,
flat-parameter trainer) with the torch-CPU stand-in of tests/fake_kernels.py in place of the HIP library,
checked against the oracle / the reference goldens.  Also: the C-ABI library loads and exports every
symbol include/eve_hip.h declares, and the product path raises when the library is missing."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import eve_amd
from eve_amd import _lib, kernels, train
from fake_kernels import FakeKernels
from oracle import detweights, sequence
from oracle.config import OracleConfig
from oracle.eye_net import EyeNet as OracleEyeNet
from oracle.refine_net import RefineNet as OracleRefineNet

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def fake():
    kernels.set_default_kernels(FakeKernels())
    yield
    kernels.set_default_kernels(None)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, 'include', 'eve_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(eve_[a-z0-9_]+)\s*\(', header)))
    assert declared == _lib.EXPORTS, set(declared) ^ set(_lib.EXPORTS)
    from eve_amd import build
    if not os.path.isfile(build.LIB_PATH):
        build.build()
    lib = ctypes.CDLL(build.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.eve_abi_version.restype = ctypes.c_int
    assert lib.eve_abi_version() == _lib.ABI_VERSION == 10
    # kernel selection is resolved once at load (include/eve_hip.h eve_dispatch_config): with a clean environment the loaded
    # table IS the default table, the Python mirror of the struct has the library's size, and a wrong-sized struct is refused
    for n in ('eve_get_dispatch_config', 'eve_get_default_dispatch_config', 'eve_set_dispatch_config'):
        getattr(lib, n).restype = ctypes.c_int
    cur, dflt = _lib.DispatchConfig(), _lib.DispatchConfig()
    assert lib.eve_get_dispatch_config(ctypes.byref(cur)) == 0 and lib.eve_get_default_dispatch_config(ctypes.byref(dflt)) == 0
    assert dflt.struct_bytes == ctypes.sizeof(_lib.DispatchConfig)
    if not any(k.startswith('EVE_') and not k.startswith('EVE_AMD_') and k not in ('EVE_HIP_LIB', 'EVE_CASES') for k in os.environ):
        assert cur.as_dict() == dflt.as_dict()
    assert dflt.as_dict()['conv_wg8_min_tiles'] == 224 and dflt.as_dict()['wgrad_halo_min_m'] == 1 << 20
    bad = _lib.DispatchConfig()
    bad.struct_bytes = 8
    assert lib.eve_set_dispatch_config(ctypes.byref(bad)) != 0


def test_no_kernel_selection_reads_the_environment_on_a_call_path():
    """getenv appears in exactly one place of the library: the load-time initialiser of the dispatch table (api.hip)."""
    import glob
    hits = []
    for path in glob.glob(os.path.join(REPO, 'eve_amd', 'csrc', '*')):
        for i, line in enumerate(open(path), 1):
            if 'getenv' in line and not line.lstrip().startswith('//'):
                hits.append((os.path.basename(path), i))
    assert hits and all(f == 'api.hip' for f, _ in hits), hits


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setenv('EVE_HIP_LIB', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.EveLibraryError):
        _lib.load()


def test_product_path_has_no_cpu_fallback():
    """Without a GPU the real kernels refuse CPU tensors instead of computing anything."""
    kernels.set_default_kernels(None)
    cfg = eve_amd.reset_standalone_config()
    net = eve_amd.EyeNet()
    batch = detweights.eyenet_batch(1, 1)
    with pytest.raises(RuntimeError, match='not on the GPU|no CPU fallback|libeve_hip'):
        net.forward_sequence(batch)
    for mod in ('eye_net', 'refine_net', 'ops', 'kernels', 'train', 'losses', 'parallel'):
        src = open(os.path.join(REPO, 'eve_amd', mod + '.py')).read()
        assert 'import oracle' not in src and 'from oracle' not in src, mod


def test_config_mirror_defaults_and_json():
    cfg = eve_amd.reset_standalone_config()
    assert cfg.refine_net_rnn_type == 'CGRU' and cfg.eye_net_rnn_type == 'GRU'      # config_default.py:101,120
    cfg.import_json(os.path.join(REPO, 'configs', 'refine_net.json'))
    assert cfg.refine_net_rnn_type == 'CLSTM' and cfg.load_screen_content is True    # refine_net.json:46,55
    assert abs(cfg.learning_rate - 8 * 0.001) < 1e-12                                  # config_default.py:81-83
    with pytest.raises(ValueError):
        cfg.override('no_such_key', 1)
    with pytest.raises(TypeError):
        cfg.import_dict({'batch_size': 'x'})


def test_unknown_rnn_type_raises_like_reference():
    cfg = eve_amd.reset_standalone_config()
    cfg.override('eye_net_rnn_type', 'XYZ')
    with pytest.raises(ValueError, match='Unknown RNN type for EyeNet'):            # eye_net.py:72
        eve_amd.EyeNet()
    eve_amd.reset_standalone_config()


def test_eyenet_host_logic_matches_oracle(fake):
    cfg = eve_amd.reset_standalone_config()
    net = eve_amd.EyeNet()
    ref = OracleEyeNet(OracleConfig())
    assert list(net.state_dict().keys()) == list(ref.state_dict().keys())
    assert float(net.fc_to_gaze[2].weight.abs().max()) == 0.0
    detweights.fill_module(net); detweights.fill_module(ref)
    batch = detweights.eyenet_batch(2, 3, seed=2, invalid_fraction=0.2)
    out = net.forward_sequence(batch)
    want = sequence.eyenet_sequence(ref, batch)
    for k in want:
        assert float((out[k] - want[k]).abs().max()) < 1e-4, k
    # the reference's per-step contract gives the same numbers as the folded pass
    steps, prev = [], None
    with torch.no_grad():
        for t in range(3):
            si = {k: v[:, t] for k, v in batch.items()}
            so = {}
            net(si, so, side='left', previous_output_dict=prev)
            net(si, so, side='right', previous_output_dict=prev)
            steps.append(so)
            prev = so
    for k in want:
        got = torch.stack([s[k] for s in steps], dim=1)
        assert float((got - want[k]).abs().max()) < 1e-4, k
    with pytest.raises(KeyError):
        net({}, {}, side='left')


VARIANTS = [dict(eye_net_rnn_type='RNN'), dict(eye_net_rnn_type='LSTM'), dict(eye_net_rnn_type='GRU', eye_net_rnn_num_cells=2),
            dict(eye_net_rnn_type='LSTM', eye_net_rnn_num_cells=2), dict(eye_net_use_rnn=False)]


def _variant_id(v):
    return '-'.join('%s' % x for x in v.values())


def check_eyenet_variant(over, to_device=lambda t: t, tol=1e-4, grad_tol=2e-3):
    """Recurrent-stage variants of eye_net.py:58-78 (RNN / LSTM / stacked cells / static_fc): folded pass, per-step
    contract (states handed over through the dicts, (h, c) tuples for LSTM) and parameter gradients vs the oracle."""
    cfg = eve_amd.reset_standalone_config()
    cfg.import_dict(over)
    net = eve_amd.EyeNet()
    ref = OracleEyeNet(OracleConfig(**over))
    assert list(net.state_dict().keys()) == list(ref.state_dict().keys())
    detweights.fill_module(net); detweights.fill_module(ref)
    net = to_device(net)
    batch = detweights.eyenet_batch(2, 3, seed=5, invalid_fraction=0.2)
    dbatch = {k: to_device(v) for k, v in batch.items()}
    # oracle: the reference's per-step loop, keeping every state (tensor or tuple)
    rsteps, prev = [], None
    for t in range(3):
        si = {k: v[:, t] for k, v in batch.items()}
        so = {}
        ref(si, so, side='left', previous_output_dict=prev)
        ref(si, so, side='right', previous_output_dict=prev)
        rsteps.append(so)
        prev = so
    out = net.forward_sequence(dbatch)
    for k in rsteps[0]:
        if isinstance(rsteps[0][k], tuple):
            for j in range(2):
                want = torch.stack([s_[k][j] for s_ in rsteps], dim=1)
                assert float((out[k][j].cpu() - want).abs().max()) < tol, (k, j)
        else:
            want = torch.stack([s_[k] for s_ in rsteps], dim=1)
            assert float((out[k].cpu() - want).abs().max()) < tol, k
    # gradients of a loss on both outputs
    def loss(o_g, o_p):
        return (o_g ** 2).sum() + o_p.sum()
    loss(torch.stack([s_['left_g_initial'] for s_ in rsteps], 1), torch.stack([s_['right_pupil_size'] for s_ in rsteps], 1)).backward()
    loss(out['left_g_initial'], out['right_pupil_size']).backward()
    rp = dict(ref.named_parameters())
    for n, p in net.named_parameters():
        if not (n.startswith('rnn_cells') or n.startswith('static_fc') or n.startswith('fc_common')):
            continue
        a, b = p.grad.cpu().double(), rp[n].grad.double()
        assert float((a - b).norm()) <= grad_tol * float(b.norm()) + 1e-7, n
    # the per-step contract of the drop-in itself
    steps, prev = [], None
    with torch.no_grad():
        for t in range(3):
            si = {k: v[:, t] for k, v in dbatch.items()}
            so = {}
            net(si, so, side='left', previous_output_dict=prev)
            net(si, so, side='right', previous_output_dict=prev)
            steps.append(so)
            prev = so
    for k in rsteps[0]:
        for t in range(3):
            a, b = steps[t][k], rsteps[t][k]
            if isinstance(b, tuple):
                assert isinstance(a, tuple) and all(float((x.cpu() - y).abs().max()) < tol for x, y in zip(a, b)), (k, t)
            else:
                assert float((a.cpu() - b.detach()).abs().max()) < tol, (k, t)


@pytest.mark.parametrize('over', VARIANTS, ids=_variant_id)
def test_eyenet_recurrent_variants_host_logic(fake, over):
    check_eyenet_variant(over)


def test_cgru_scan_function_matches_autograd(fake):
    """ops.CGRUScanFn (one launch for the clip + manual BPTT with batched weight gradients) == autograd through the
    CGRUCell formula of common.py:388-415 applied frame by frame."""
    from eve_amd import ops
    torch.manual_seed(0)
    B, T, H, W, C = 2, 3, 5, 8, 64
    xs = torch.randn(B, T, H, W, C, requires_grad=True)
    h0 = (torch.randn(B, H, W, C) * 0.5).requires_grad_(True)
    w1 = (torch.randn(2 * C, 2 * C, 3, 3) * 0.03).requires_grad_(True)
    b1 = (torch.randn(2 * C) * 0.1).requires_grad_(True)
    w2 = (torch.randn(C, 2 * C, 3, 3) * 0.03).requires_grad_(True)
    b2 = (torch.randn(C) * 0.1).requires_grad_(True)
    p1, p2 = ops.PackedWeight(w1, torch.float32), ops.PackedWeight(w2, torch.float32)
    hs = ops.CGRUScanFn.apply(xs, w1, b1, w2, b2, h0, p1, p2)
    probe = torch.randn_like(hs)
    (hs * probe).sum().backward()
    got = [t.grad.clone() for t in (xs, h0, w1, b1, w2, b2)]
    for t in (xs, h0, w1, b1, w2, b2):
        t.grad = None
    h, outs = h0.permute(0, 3, 1, 2), []
    for t in range(T):
        x = xs[:, t].permute(0, 3, 1, 2)
        g1 = torch.sigmoid(torch.nn.functional.conv2d(torch.cat([x, h], 1), w1, b1, padding=1))
        r, u = g1.chunk(2, 1)
        o = torch.tanh(torch.nn.functional.conv2d(torch.cat([r * h, x], 1), w2, b2, padding=1))
        h = (1. - u) * o + u * h
        outs.append(h.permute(0, 2, 3, 1))
    ref = torch.stack(outs, 1)
    assert float((hs - ref).abs().max()) < 1e-5
    (ref * probe).sum().backward()
    for a, t, n in zip(got, (xs, h0, w1, b1, w2, b2), ('xs', 'h0', 'w1', 'b1', 'w2', 'b2')):
        assert float((a - t.grad).abs().max()) <= 1e-4 * float(t.grad.abs().max()) + 1e-6, n


def test_eyenet_frozen_detaches(fake):
    cfg = eve_amd.reset_standalone_config()
    cfg.override('eye_net_frozen', True)
    net = detweights.fill_module(eve_amd.EyeNet())
    batch = detweights.eyenet_batch(1, 1)
    out = {}
    net({k: v[:, 0] for k, v in batch.items()}, out, side='left')
    assert not out['left_g_initial'].requires_grad and out['left_pupil_size'].requires_grad
    eve_amd.reset_standalone_config()


def test_flat_trainer_reproduces_reference_train_step(fake):
    fx = np.load(os.path.join(REPO, 'tests', 'golden', 'eyenet.npz'))
    cfg = eve_amd.reset_standalone_config()
    cfg.import_json(os.path.join(REPO, 'configs', 'eye_net.json'))
    net = detweights.fill_module(eve_amd.EyeNet())
    keys = list(net.state_dict().keys())
    tr = train.eyenet_trainer(net, cfg)
    assert list(net.state_dict().keys()) == keys
    batch = detweights.eyenet_batch(int(fx['B']), int(fx['T']), seed=0, invalid_fraction=float(fx['invalid_fraction']))
    terms = tr.step(batch)
    np.testing.assert_allclose(float(terms['full_loss'].detach()), float(fx['eve_full_loss']), rtol=1e-5)
    np.testing.assert_allclose(float(tr.sumsq.sqrt()), float(fx['clip_total_norm']), rtol=1e-3)
    sd = net.state_dict()
    for k in fx.files:
        if k.startswith('updated_'):
            np.testing.assert_allclose(sd[k[8:]].reshape(-1)[:16].numpy(), fx[k], rtol=1e-4, atol=2e-6)
    # a second step keeps .grad inside the flat buffer and moves the weights again
    before = tr.fp.flat.clone()
    tr.step(batch)
    assert float((tr.fp.flat - before).abs().max()) > 0
    for p, off, n in tr.fp.entries:
        assert p.grad.data_ptr() == tr.fp.grad.data_ptr() + 4 * off


@pytest.mark.parametrize('kind', ['CGRU', 'CLSTM', 'CRNN'])
def test_refinenet_host_logic_matches_oracle(fake, kind):
    cfg = eve_amd.reset_standalone_config()
    cfg.import_dict({'load_screen_content': True, 'refine_net_enabled': True, 'refine_net_rnn_type': kind})
    ocfg = OracleConfig(load_screen_content=True, refine_net_enabled=True, refine_net_rnn_type=kind)
    net, ref = eve_amd.RefineNet(), OracleRefineNet(ocfg)
    assert list(net.state_dict().keys()) == list(ref.state_dict().keys())
    assert float(net.final[2].weight.abs().max()) == 0.0                                # refine_net.py:235
    detweights.fill_module(net, 1); detweights.fill_module(ref, 1)
    rb = detweights.refinenet_batch(2, 2, seed=5)
    hf, _ = net.forward_sequence(rb['heatmap_initial'], rb['screen_frame'])
    want, _ = sequence.refinenet_sequence(ref, rb['heatmap_initial'], rb['screen_frame'])
    assert float((hf - want).abs().max()) < 1e-4
    sequence.refinenet_losses(hf, rb['heatmap_final_gt'], rb['validity'], ocfg)['full_loss'].backward()
    sequence.refinenet_losses(want, rb['heatmap_final_gt'], rb['validity'], ocfg)['full_loss'].backward()
    rp = dict(ref.named_parameters())
    for n, p in net.named_parameters():
        if rp[n].grad is None:
            assert p.grad is None, n
        else:
            a, b = p.grad.double(), rp[n].grad.double()
            assert float((a - b).norm()) <= 3e-2 * float(b.norm()) + 1e-5, n
    eve_amd.reset_standalone_config()


@pytest.mark.parametrize('cin,cout,W,ks', [(16, 16, 16, 3), (16, 32, 8, 3), (8, 16, 32, 3), (32, 16, 128, 3),
                                            (16, 32, 8, 1), (64, 16, 12, 1), (16, 8, 16, 1), (32, 64, 6, 1)])
def test_pixel_group_convolution_equals_plain_convolution(fake, monkeypatch, cin, cout, W, ks):
    """ops.Conv2dFn on 8/16-channel tensors runs the convolution over groups of 4/2 pixels (32-channel rows); the
    grouped filter must give the same forward, data gradient, weight and bias gradient as the plain one."""
    from eve_amd import ops
    torch.manual_seed(cin + cout)
    x0 = torch.randn(2, 6, W, cin).bfloat16()
    wt0 = (torch.randn(cout, cin, ks, ks) * 0.1)
    b0 = torch.randn(cout)
    fac = (8, 16, 32)
    outs = []
    for grouped in (True, False):
        if not grouped:
            monkeypatch.setattr(ops, 'PAIR_FACTOR', {})
            monkeypatch.setattr(ops, 'PAIR_FACTOR_1X1', {})
        x = x0.clone().requires_grad_(True)
        wt, b = wt0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        pack = ops.PackedWeight(wt, torch.bfloat16)
        assert (pack.pair_fwd is not None) == (grouped and cin in fac)
        assert (pack.pair_dgrad is not None) == (grouped and cout in fac)
        y = ops.conv2d(x, wt, b, pack, stride=1, pad=ks // 2, act=1)
        (y.float() * torch.linspace(-1, 1, y.numel()).view_as(y)).sum().backward()
        outs.append((y.detach().float(), x.grad.float(), wt.grad, b.grad))
    for a, c in zip(*outs):
        assert float((a - c).abs().max()) <= 2e-2 * float(c.abs().max()) + 1e-6


EVE_CASES = {
    'c3': (True, {}),
    'eval': (False, {}),
    'joint': (True, dict(eye_net_frozen=False, loss_coeff_PoG_cm_initial=0.002, loss_coeff_g_ang_initial=1.0,
                         loss_coeff_pupil_size=1.0, loss_coeff_heatmap_mse_final=0.5, loss_coeff_PoG_cm_final=0.01)),
}


@pytest.mark.parametrize('tag', sorted(EVE_CASES))
def test_eve_harness_host_logic_matches_oracle(fake, tag):
    """eve_amd.EVE (batched schedule, kernels faked on CPU) against oracle.eve.eve_forward (the reference's per-frame
    data flow): same label synthesis, kappa draws, every loss / metric scalar, full_loss, and the same gradients in
    both networks -- including the path RefineNet -> heat-map -> PoG -> augmentation -> EyeNet of the joint case."""
    from oracle import eve as oracle_eve
    training, over = EVE_CASES[tag]
    json_path = os.path.join(REPO, 'configs', 'refine_net.json')
    cfg = eve_amd.reset_standalone_config()
    cfg.import_json(json_path)
    cfg.import_dict(dict(refine_net_rnn_type='CGRU', eye_net_load_pretrained=False, **over))
    ocfg = OracleConfig(json_path, refine_net_rnn_type='CGRU', eye_net_load_pretrained=False, **over)
    model = eve_amd.EVE(output_predictions=True)
    detweights.fill_module(model.eye_net, 0); detweights.fill_module(model.refine_net, 1)
    oeye, oref = detweights.fill_module(OracleEyeNet(ocfg), 0), detweights.fill_module(OracleRefineNet(ocfg), 1)
    if ocfg.eye_net_frozen:
        for p in oeye.parameters():
            p.requires_grad = False
    model.train(training)
    batch = detweights.eve_batch(2, 3, seed=4, invalid_fraction=0.25)
    np.random.seed(5)
    want, winter, wlabels = oracle_eve.eve_forward(oeye, oref, dict(batch), ocfg, training, create_images=not training)
    np.random.seed(5)
    mine = dict(batch)
    got = model({'src': mine} if training else mine, create_images=not training, current_epoch=0.0)
    assert {k for k in got if k.startswith(('loss_', 'metric_'))} == set(want.keys()) - {'full_loss'}
    for k, v in want.items():
        assert abs(float(got[k].detach()) - float(v.detach())) <= 2e-4 * abs(float(v)) + 1e-5, (k, float(got[k]), float(v))
    for k in ('g_initial', 'PoG_px_initial', 'PoG_cm_initial', 'g_final', 'PoG_px_final', 'PoG_cm_final'):
        assert float((got[k] - winter[k]).abs().max()) < (0.1 if 'px' in k else 4e-3), k     # soft-argmax (beta 100) amplifies fp32 noise
    for k in ('g', 'PoG_px_tobii', 'heatmap_final', 'heatmap_initial', 'o'):
        assert float((mine[k] - wlabels[k]).abs().max()) < 1e-4, k
    if training:
        assert torch.equal(mine['left_kappa_fake'], wlabels['left_kappa_fake'])
        got['full_loss'].backward(); want['full_loss'].backward()
        for net, onet in ((model.eye_net, oeye), (model.refine_net, oref)):
            ref = dict(onet.named_parameters())
            for n, p in net.named_parameters():
                if ref[n].grad is None:
                    assert p.grad is None, n
                else:
                    a, b = p.grad.double(), ref[n].grad.double()
                    # (biases in front of an InstanceNorm have a zero gradient: what both sides hold is rounding noise
                    #  that scales with the loss)
                    assert float((a - b).norm()) <= 3e-2 * float(b.norm()) + 1e-5 * max(1.0, float(want['full_loss'])), n
    else:
        assert float((got['initial_gaze_history'] - winter['history_initial_last']).abs().max()) < 1e-3
        assert float((got['refined_gaze_history'] - winter['refined_gaze_history']).abs().max()) < 1e-3
    eve_amd.reset_standalone_config()


def test_uint8_frames_follow_the_reference_normalisation(fake, golden_dir):
    """oracle/frames.py against the reference's own preprocess_frames / preprocess_screen_frames (fixture), and the
    product's uint8 entry (eve_amd.data + EyeNet.forward_sequence on [B,T,H,W,C] uint8) against the float path."""
    from eve_amd import data
    from oracle import frames as oframes
    fx = np.load(os.path.join(golden_dir, 'frames.npz'))
    for key in ('ramp', 'frames'):
        assert np.array_equal(oframes.preprocess_frames(fx[key]), fx[key + '_eye'])
        assert np.array_equal(oframes.preprocess_screen_frames(fx[key]), fx[key + '_screen'])
        assert np.array_equal(data.preprocess_frames(torch.from_numpy(fx[key])).numpy(), fx[key + '_eye'])
        assert np.array_equal(data.preprocess_screen_frames(torch.from_numpy(fx[key])).numpy(), fx[key + '_screen'])
    cfg = eve_amd.reset_standalone_config()
    net = detweights.fill_module(eve_amd.EyeNet(), 0)
    g = np.random.Generator(np.random.PCG64(5))
    u8 = {s: torch.from_numpy(g.integers(0, 256, size=(1, 2, 64, 64, 3), dtype=np.uint8)) for s in ('left', 'right')}
    head = {s + '_h': torch.zeros(1, 2, 2) for s in ('left', 'right')}
    as_float = {s + '_eye_patch': torch.from_numpy(oframes.preprocess_frames(u8[s].numpy().reshape(2, 64, 64, 3))).view(1, 2, 3, 64, 64)
                for s in ('left', 'right')}
    a = net.forward_sequence(dict(head, **{s + '_eye_patch': u8[s] for s in u8}))
    b = net.forward_sequence(dict(head, **as_float))
    for k in ('left_g_initial', 'right_pupil_size'):
        assert torch.equal(a[k], b[k]), k


def test_checkpoints_interchange_with_the_reference_layout(fake, tmp_path):
    """eve_amd.checkpoint writes what the reference's CheckpointManager writes (checkpoint_manager.py:47-74): a directory
    checkpoints/%07d.pt/ with eye_net.pt / refine_net.pt (prefixed keys) and optimizer_0.pt in torch.optim.Adam's
    format -- loadable, strictly, by a model with the reference's parameter names, and the other way round."""
    from eve_amd import checkpoint, train
    json_path = os.path.join(REPO, 'configs', 'refine_net.json')
    cfg = eve_amd.reset_standalone_config()
    cfg.import_json(json_path)
    cfg.import_dict(dict(refine_net_rnn_type='CGRU', eye_net_load_pretrained=False, refine_net_do_offset_augmentation=False))
    ocfg = OracleConfig(json_path, refine_net_rnn_type='CGRU', eye_net_load_pretrained=False)
    model = eve_amd.EVE().train()
    detweights.fill_module(model.eye_net, 0); detweights.fill_module(model.refine_net, 1)
    tr = train.eve_trainer(model, cfg)
    tr.step(detweights.eve_batch(1, 2, seed=8))
    out = str(tmp_path)
    path = checkpoint.save(model, out, 12, trainer=tr)
    assert path.endswith(os.path.join('checkpoints', '0000012.pt')) and os.path.isdir(path)
    assert sorted(os.listdir(path)) == ['eye_net.pt', 'optimizer_0.pt', 'refine_net.pt']
    part = torch.load(os.path.join(path, 'refine_net.pt'))
    assert all(k.startswith('refine_net.') for k in part) and all(v.is_contiguous() for v in part.values())

    class ReferenceShaped(torch.nn.Module):               # the module tree of models/eve.py:49-66 with the oracle's networks
        def __init__(self):
            super().__init__()
            self.eye_net, self.refine_net = OracleEyeNet(ocfg), OracleRefineNet(ocfg)
    ref = ReferenceShaped()
    merged = {}
    for f in ('eye_net.pt', 'refine_net.pt'):
        merged.update(torch.load(os.path.join(path, f)))
    ref.load_state_dict(merged)                           # strict
    for (n, a), (_, b) in zip(model.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(a.cpu(), b), n
    # the reference builds Adam over ALL parameters, the frozen EyeNet included (src/train.py:49-55): the saved state must
    # load into exactly that optimizer, with the RefineNet entries numbered after the 37 EyeNet parameters
    for p_ in ref.eye_net.parameters():
        p_.requires_grad = False
    opt = torch.optim.Adam(ref.parameters(), lr=cfg.learning_rate, weight_decay=cfg.weight_decay)
    opt.load_state_dict(torch.load(os.path.join(path, 'optimizer_0.pt')))
    st = opt.state_dict()['state']
    n_eye = len(list(ref.eye_net.parameters()))
    assert n_eye == 37 and min(st) == n_eye and len(st) == len(list(ref.refine_net.parameters()))
    assert float(st[n_eye]['step']) == 1.0 and float(st[n_eye + 3]['exp_avg'].abs().max()) > 0
    rparams = list(ref.parameters())
    assert all(tuple(v['exp_avg'].shape) == tuple(rparams[i].shape) for i, v in st.items())
    # a state numbered by another parameter list is refused, not copied into the wrong tensors
    wrong = torch.optim.Adam(ref.refine_net.parameters()).state_dict()
    with pytest.raises(ValueError):
        checkpoint.load_adam_state_dict(tr, wrong, model)

    # the other direction: a checkpoint written the reference's way, read by the drop-in (+ keep-N, + load_last)
    detweights.fill_module(ref.eye_net, 5); detweights.fill_module(ref.refine_net, 6)
    their = os.path.join(out, 'checkpoints', '0000040.pt')
    os.makedirs(their)
    sd = ref.state_dict()
    for prefix in ('eye_net', 'refine_net'):
        torch.save({k: v for k, v in sd.items() if k.startswith(prefix + '.')}, os.path.join(their, prefix + '.pt'))
    torch.save(opt.state_dict(), os.path.join(their, 'optimizer_0.pt'))
    assert checkpoint.load_last(model, out, trainer=tr) == 40
    for (n, a), (_, b) in zip(model.state_dict().items(), sd.items()):
        assert torch.equal(a.cpu(), b), n
    assert tr.step_count == 1 and float(tr.fp.m.abs().max()) > 0
    for s in (41, 42, 43):
        checkpoint.save(model, out, s, trainer=tr, keep_n=3)
    assert [s for s, _ in checkpoint.available(out)] == [41, 42, 43]
    eve_amd.reset_standalone_config()


def test_trainer_follows_the_reference_lr_schedule(fake):
    """Trainer(lr_schedule=...) with eve_amd.schedule.effective_learning_rate == torch.optim.Adam stepped with the LR the
    reference's LambdaLR arrangement produces (src/core/training.py:382-418,436-442,576-577; pinned by
    tests/golden/lr_schedule.npz): six steps across the warm-up -> decay boundary, on the same gradients."""
    from eve_amd import schedule
    from oracle import sequence
    cfg = eve_amd.reset_standalone_config()
    cfg.import_dict(dict(batch_size=8, base_learning_rate=0.0005, num_warmup_epochs=0.5, lr_decay_strategy='exponential',
                         lr_decay_factor=0.5, lr_decay_epoch_interval=0.5, weight_decay=0.005))
    epoch_len = 4                                       # warm-up = 2 steps, then a decay every 2 steps
    torch.manual_seed(0)
    lin = torch.nn.Linear(6, 5)
    shadow = [torch.nn.Parameter(p.detach().clone()) for p in lin.parameters()]
    opt = torch.optim.Adam(shadow, lr=cfg.learning_rate, weight_decay=cfg.weight_decay)
    data = torch.randn(6, 7, 6)
    tr = train.Trainer([lin], cfg, lambda b: {'full_loss': (lin(b['x']) ** 2).mean()},
                       lr_schedule=lambda s: schedule.effective_learning_rate(cfg, epoch_len, s))
    ocfg = OracleConfig(batch_size=8, base_learning_rate=0.0005, num_warmup_epochs=0.5, lr_decay_strategy='exponential',
                        lr_decay_factor=0.5, lr_decay_epoch_interval=0.5)
    seen = []
    for s in range(6):
        tr.step({'x': data[s]})
        seen.append(tr.lr)
        for g in opt.param_groups:
            g['lr'] = sequence.lr_used_by_step(ocfg, epoch_len, s)
        for p, q in zip(shadow, lin.parameters()):
            p.grad = q.grad.detach().clone()
        torch.nn.utils.clip_grad_norm_(shadow, cfg.gradient_clip_amount)
        opt.step()
        for p, q in zip(shadow, lin.parameters()):
            assert float((p - q).abs().max()) < 1e-6, s
    assert seen[0] < seen[1] < seen[2] == seen[3] and seen[4] == 0.5 * seen[3]     # warm-up rises, the decay halves
    assert float(tr.lr_dev) == pytest.approx(seen[-1])
    eve_amd.reset_standalone_config()


def test_gradient_bucket_plan_matches_the_survey(fake):
    """the gradient all-reduce runs in a few large buckets filled back to front, so that the layers whose
    gradients are ready first (EyeNet: layer 4 = 74 % of the bytes; RefineNet: the decoder / final convolutions) are on the
    wire while the rest of backward still runs.  Checked on the plan itself (no process group needed): 3-5 buckets for the
    EyeNet (configs[1]) and RefineNet (configs[2]) trainers, a partition of the flat gradient buffer, bucket 0 = the LAST
    parameters, and every bucket except the remainder at least the configured size."""
    from eve_amd.parallel import GradSync
    cfg = eve_amd.reset_standalone_config()
    cfg.import_json(os.path.join(REPO, 'configs', 'eye_net.json'))
    eye = eve_amd.EyeNet()
    tr = train.eyenet_trainer(eye, cfg)
    sync = GradSync(tr.fp.grad, tr.fp.entries)
    assert 3 <= len(sync.buckets) <= 5, [b['hi'] - b['lo'] for b in sync.buckets]
    # a partition of [0, padded numel), back to front
    assert sync.buckets[0]['hi'] == tr.fp.grad.numel() and sync.buckets[-1]['lo'] == 0
    for a, b in zip(sync.buckets, sync.buckets[1:]):
        assert a['lo'] == b['hi']
    assert all(b['hi'] - b['lo'] >= 4 * 1024 * 1024 for b in sync.buckets[:-1])
    names = {id(p): n for n, p in eye.named_parameters()}
    first = [names[id(p)] for p in sync.buckets[0]['params']]
    # the tail (heads, GRU, fc) and layer 4's last convolutions -- the first gradients backward produces -- lead
    assert 'fc_to_gaze.2.weight' in first and 'cnn_layers.layer4.1.conv2.weight' in first
    assert not any(n.startswith(('cnn_layers.conv1', 'cnn_layers.layer1', 'cnn_layers.layer2')) for n in first)
    l4 = sum(p.numel() for n, p in eye.named_parameters() if n.startswith('cnn_layers.layer4'))
    assert 0.70 < l4 / sum(p.numel() for p in eye.parameters()) < 0.78            # "74 % of the bytes"
    last = [names[id(p)] for p in sync.buckets[-1]['params']]
    assert 'cnn_layers.conv1.weight' in last                                      # the stem's gradient is ready last
    # RefineNet (configs[2]): 5.27 M parameters -> the 4 M-element default gives two buckets; the DP step of configs[3]
    # lowers the bucket size so that the transfer still overlaps backward (eve_dispatch_config.bucket_elems), 3-5 buckets at 1.5 M
    cfg = eve_amd.reset_standalone_config()
    cfg.import_dict({'load_screen_content': True, 'refine_net_enabled': True, 'refine_net_rnn_type': 'CGRU'})
    ref = eve_amd.RefineNet()
    rtr = train.refinenet_trainer(ref, cfg)
    rsync = GradSync(rtr.fp.grad, rtr.fp.entries, bucket_elems=1536 * 1024)
    assert 3 <= len(rsync.buckets) <= 5, [b['hi'] - b['lo'] for b in rsync.buckets]
    rnames = {id(p): n for n, p in ref.named_parameters()}
    assert 'final.2.weight' in [rnames[id(p)] for p in rsync.buckets[0]['params']]
    assert 'initial.0.weight' in [rnames[id(p)] for p in rsync.buckets[-1]['params']]
    assert sum(b['hi'] - b['lo'] for b in rsync.buckets) == rtr.fp.grad.numel()
    eve_amd.reset_standalone_config()


def test_trainer_factories_take_the_reference_schedule_and_the_fp16_loss_scale(fake):
    """eyenet_trainer(..., steps_per_epoch=n) steps with the reference's effective learning rate (LambdaLR quirk included,
    eve_amd.schedule); without it the LR is the constant config.learning_rate.  A float16 module gets the static loss scale
    (1024) and the Adam kernel's gradient scale divides it out again: the update equals the unscaled one."""
    from eve_amd import schedule
    cfg = eve_amd.reset_standalone_config()
    cfg.import_json(os.path.join(REPO, 'configs', 'eye_net.json'))
    net = detweights.fill_module(eve_amd.EyeNet())
    tr = train.eyenet_trainer(net, cfg, steps_per_epoch=100)
    batch = detweights.eyenet_batch(1, 2, seed=5)
    for s in range(3):
        tr.step(batch)
        assert tr.lr == pytest.approx(schedule.effective_learning_rate(cfg, 100, s))
    assert tr.lr < cfg.learning_rate                       # warm-up x the LambdaLR multiplication
    assert train.eyenet_trainer(detweights.fill_module(eve_amd.EyeNet()), cfg).lr_schedule is None
    # loss scale: same data, same weights, scale 1 vs 1024 -> identical updates (float32 fake kernels, exact powers of two)
    res = []
    for scale in (1.0, 1024.0):
        n2 = detweights.fill_module(eve_amd.EyeNet())
        t2 = train.Trainer([n2], cfg, lambda b, n2=n2: sequence.eyenet_losses(n2.forward_sequence(b), b, OracleConfig(
            batch_size=16, weight_decay=0.005, base_learning_rate=0.001)), loss_scale=scale)
        t2.step(batch)
        res.append((t2.fp.flat.clone(), float(t2.sumsq.sqrt()) / scale))
    assert float((res[0][0] - res[1][0]).abs().max()) < 1e-7
    assert res[0][1] == pytest.approx(res[1][1], rel=1e-5)
    n16 = eve_amd.EyeNet()
    n16.compute_dtype = torch.float16
    t16 = train.Trainer([n16], cfg, lambda b: None)
    assert t16.loss_scale == 1024.0 and t16.check_overflow
    # ---- an overflowed step is SKIPPED on the device, visibly: weights, moments and the bias-correction counter stay, the
    # skip is counted, two in a row halve the loss scale, and the next finite step is Adam step 2 (not 4) ----
    n3 = detweights.fill_module(eve_amd.EyeNet())
    poison = {'on': False}

    def loss_fn(b):
        terms = sequence.eyenet_losses(n3.forward_sequence(b), b, OracleConfig(batch_size=16, weight_decay=0.005, base_learning_rate=0.001))
        if poison['on']:
            terms['full_loss'] = terms['full_loss'] * float('inf')
        return terms
    t3 = train.Trainer([n3], cfg, loss_fn, loss_scale=1024.0)
    t3.step(batch)
    assert t3.optimizer_state() == {'steps_taken': 1, 'steps_skipped': 0, 'loss_scale': 1024.0, 'steps_skipped_gate_timeout': 0}
    w1, m1 = t3.fp.flat.clone(), t3.fp.m.clone()
    poison['on'] = True
    t3.step(batch)
    t3.step(batch)
    assert torch.equal(t3.fp.flat, w1) and torch.equal(t3.fp.m, m1)
    assert t3.optimizer_state() == {'steps_taken': 1, 'steps_skipped': 2, 'loss_scale': 512.0, 'steps_skipped_gate_timeout': 0}
    poison['on'] = False
    t3.step(batch)
    assert t3.optimizer_state()['steps_taken'] == 2 and not torch.equal(t3.fp.flat, w1) and torch.isfinite(t3.fp.flat).all()
    from eve_amd import checkpoint
    assert float(checkpoint.adam_state_dict(t3)['state'][0]['step']) == 2.0          # steps taken, not step() calls (4)
    # ---- a POISONED gradient exchange (a stream gate of the data-parallel replay timed out: parallel.GradSync.launch_gated) is
    # skipped the same way, without touching the loss scale: the poison word is the first of FlatParameters' leading pad floats,
    # inside the last bucket, so every rank reads it after the all-reduce (stand-in semantics = csrc/optim.hip adam_prepare_kernel)
    assert train.FlatParameters.LEAD == 64 and t3.fp.entries[0][1] == 64 and t3.fp.poison.data_ptr() == t3.fp.grad.data_ptr()
    k = kernels.default_kernels()
    w2, st = t3.fp.flat.clone(), t3.optimizer_state()
    t3.fp.poison.fill_(float('inf'))
    k.adam_step(t3.fp.flat, t3.fp.grad, t3.fp.m, t3.fp.v, None, 0.0, 1.0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, guard=t3.guard, poison=t3.fp.poison)
    assert torch.equal(t3.fp.flat, w2)
    assert t3.optimizer_state() == dict(st, steps_skipped=st['steps_skipped'] + 1, steps_skipped_gate_timeout=1)
    t3.step(batch)                                                                   # zero_grad clears the word: training goes on
    assert float(t3.fp.poison[0]) == 0.0 and t3.optimizer_state()['steps_taken'] == 3
    eve_amd.reset_standalone_config()


def test_gated_launch_order_puts_the_poison_bucket_last():
    """parallel.GradSync.launch_gated: whatever order the ready points were captured in, the bucket that carries the poison word
    (lo == 0) is all-reduced behind every gate that could still write it, and a fall-back issues self.buckets order on every rank."""
    from eve_amd.parallel import GradSync
    flat = torch.zeros(4 + 3000)
    ps = [torch.nn.Parameter(torch.zeros(1000)) for _ in range(3)]
    sync = GradSync(flat, [(p, 4 + 1000 * i, 1000) for i, p in enumerate(ps)], bucket_elems=1000, poison=flat[0:1])
    assert [(b['lo'], b['hi']) for b in sync.buckets] == [(2004, 3004), (1004, 2004), (0, 1004)]
    calls = []
    sync.active = True
    sync._flags = None
    sync._comm = None
    sync._launch = lambda b, inline=False: (calls.append((sync.buckets.index(b), inline)), b.__setitem__('launched', True))

    class _Ctx(object):
        def __enter__(self): return self
        def __exit__(self, *a): return False
    import unittest.mock as um
    with um.patch('torch.cuda.stream', lambda s: _Ctx()), um.patch('torch.cuda.current_stream', lambda: um.MagicMock()):
        sync._comm = um.MagicMock()
        sync._gated = [sync.buckets[2], sync.buckets[0]]          # captured order: the poison bucket reported FIRST
        sync.launch_gated()
    # gated buckets first, the poison bucket behind the other GATED one; bucket 1 never reported, follows the whole replay ungated
    assert calls == [(0, True), (2, True), (1, True)], calls        # (inline: the collectives go onto the communication stream itself)
    calls.clear()
    with um.patch('torch.cuda.stream', lambda s: _Ctx()), um.patch('torch.cuda.current_stream', lambda: um.MagicMock()):
        sync.disable_gating()                                     # the fall-back: self.buckets order, poison bucket last
        sync.launch_gated()
    assert [c[0] for c in calls] == [0, 1, 2], calls
