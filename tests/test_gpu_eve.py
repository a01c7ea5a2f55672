"""GPU: the EVE sequence harness -- the gaze-geometry / heat-map / soft-argmax kernels against the
oracle's restatement on seeded inputs, and eve_amd.EVE end to end against the golden fixture produced by the reference's
own models.eve.EVE (tests/golden/make_golden_eve.py)."""
import os

import numpy as np
import pytest
import torch

import eve_amd
from eve_amd import kernels, ops
from oracle import detweights
from oracle import eve as oracle_eve
from oracle.config import OracleConfig

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, 'tests', 'golden')


@pytest.fixture(scope='module')
def hip():
    return kernels.default_kernels()


def frames(N, seed):
    """Flat geometry inputs of N frames from the synthetic clip generator."""
    b = detweights.eve_batch(N, 1, seed=seed)
    g = torch.Generator().manual_seed(seed)
    return {'g': 0.4 * torch.randn(N, 2, generator=g), 'kappa': 0.05 * torch.randn(N, 2, generator=g),
            'o': b['left_o'].reshape(N, 3), 'R': b['left_R'].reshape(N, 3, 3), 'head_R': b['head_R'].reshape(N, 3, 3),
            'inv': b['inv_camera_transformation'].reshape(N, 4, 4), 'cam': b['camera_transformation'].reshape(N, 4, 4),
            'ppm': b['pixels_per_millimeter'].reshape(N, 2)}


@pytest.mark.parametrize('augment', [False, True], ids=['plain', 'kappa'])
def test_gaze_to_pog_values_and_jacobians(hip, augment):
    N, screen = 300, (1920, 1080)
    f = frames(N, 11)
    # a third of the rays are pushed off-screen so that the clamp (and its zero gradient) is exercised
    f['g'][::3] *= 3.0
    gi = f['g'].clone().requires_grad_(True)
    go = oracle_eve.offset_augmentation(gi, f['head_R'], f['kappa']) if augment else gi
    mm, px = oracle_eve.to_screen_coordinates(f['o'], go, f['R'], f['inv'], f['ppm'], screen)
    c = lambda t: t.cuda()
    got_g, got_mm, got_px, jac = hip.gaze_to_pog(c(f['g']), c(f['o']), c(f['R']), c(f['inv']), c(f['ppm']), screen,
                                                 c(f['head_R']) if augment else None, c(f['kappa']) if augment else None)
    assert float((got_g.cpu() - go.detach()).abs().max()) < 3e-5           # asin near +-1 (rays pushed to 1+ rad) is ill-conditioned
    assert float((got_mm.cpu() - mm.detach()).abs().max()) < 2e-5 * float(mm.detach().abs().max())
    assert float((got_px.cpu() - px.detach()).abs().max()) < 2e-2
    assert 0.05 < float(((px.detach() == 0) | (px.detach()[:, :1] == 1920)).float().mean()) < 0.9
    want = torch.zeros(N, 6, 2)
    for oi, o in enumerate((go, mm, px)):
        for comp in range(2):
            if o.requires_grad and o.grad_fn is not None:
                want[:, 2 * oi + comp] = torch.autograd.grad(o[:, comp].sum(), gi, retain_graph=True)[0]
            else:
                want[:, 2 * oi + comp, comp] = 1.0            # g_out is g itself
    err = (jac.cpu() - want).abs()
    assert float((err / (want.abs() + 1e-3 * want.abs().max())).max()) < 2e-3
    # the backward kernel is the transposed product with those Jacobians
    d = [torch.randn(N, 2) for _ in range(3)]
    dg = hip.gaze_to_pog_bwd(jac, c(d[0]), c(d[1]), c(d[2])).cpu()
    ref = sum(torch.einsum('ni,nij->nj', d[i], jac.cpu()[:, 2 * i:2 * i + 2]) for i in range(3))
    assert float((dg - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    only_mm = hip.gaze_to_pog_bwd(jac, None, c(d[1]), None).cpu()
    assert float((only_mm - torch.einsum('ni,nij->nj', d[1], jac.cpu()[:, 2:4])).abs().max()) <= 1e-5 * float(ref.abs().max())


def test_combined_gaze_direction(hip):
    N = 257
    f = frames(N, 12)
    pog = torch.stack([torch.rand(N) * 553, torch.rand(N) * 311], dim=1)
    want = oracle_eve.combined_gaze_direction(f['o'], pog, f['R'], f['cam'])
    got = hip.combined_gaze(f['o'].cuda(), pog.cuda(), f['R'].cuda(), f['cam'].cuda()).cpu()
    assert float((got - want).abs().max()) < 2e-6


@pytest.mark.parametrize('sigma', [10.0, 3.0])
def test_heatmaps_forward_validity_and_backward(hip, sigma):
    cfg = OracleConfig()
    N = 37
    g = torch.Generator().manual_seed(3)
    centres = torch.stack([torch.rand(N, generator=g) * 2200 - 100, torch.rand(N, generator=g) * 1300 - 100], dim=1)
    valid = torch.rand(N, generator=g) > 0.3
    ci = centres.clone().requires_grad_(True)
    want = oracle_eve.make_heatmaps(ci, sigma, cfg)
    got = hip.make_heatmaps(centres.cuda(), sigma, (72, 128), (1920, 1080)).cpu()
    assert got.shape == (N, 1, 72, 128) and float((got - want.detach()).abs().max()) < 2e-6
    masked = hip.make_heatmaps(centres.cuda(), sigma, (72, 128), (1920, 1080), validity=valid.cuda()).cpu()
    assert float((masked - want.detach() * valid.float().view(-1, 1, 1, 1)).abs().max()) < 2e-6
    dout = torch.randn(N, 1, 72, 128, generator=g)
    wantd = torch.autograd.grad((want * dout).sum(), ci)[0]
    gotd = hip.make_heatmaps_bwd(centres.cuda(), sigma, (1920, 1080), dout.cuda()).cpu()
    assert float((gotd - wantd).abs().max()) <= 1e-4 * float(wantd.abs().max()) + 1e-7


def test_soft_argmax_forward_and_backward(hip):
    cfg = OracleConfig()
    N = 29
    g = torch.Generator().manual_seed(5)
    centres = torch.stack([torch.rand(N, generator=g) * 1920, torch.rand(N, generator=g) * 1080], dim=1)
    heat = oracle_eve.make_heatmaps(centres, 5.0, cfg) * 0.9 + 0.05 * torch.rand(N, 1, 72, 128, generator=g)
    heat[0].zero_()                                               # flat map -> centre of the screen
    heat[1, 0, 0, 0] = 5.0                                        # a spike in the corner: the clamp boundary
    hi = heat.clone().requires_grad_(True)
    want = oracle_eve.soft_argmax(hi, cfg)
    got, stats = hip.soft_argmax_fwd(heat.cuda(), (1920, 1080))
    assert float((got.cpu() - want.detach()).abs().max()) < 5e-2           # pixels on a 1920-wide screen, fp32 softmax
    dp = torch.randn(N, 2, generator=g)
    wantd = torch.autograd.grad((want * dp).sum(), hi)[0]
    gotd = hip.soft_argmax_bwd(heat.cuda(), stats, dp.cuda(), (1920, 1080)).cpu()
    assert float((gotd - wantd).abs().max()) <= 2e-4 * float(wantd.abs().max())


# ---------------------------------------------------------------------------------------------- the module, end to end
EVE_CASES = {
    'c3': (True, {}),
    'eval': (False, {}),
    'joint': (True, dict(eye_net_frozen=False, loss_coeff_PoG_cm_initial=0.002, loss_coeff_g_ang_initial=1.0,
                         loss_coeff_pupil_size=1.0, loss_coeff_heatmap_mse_final=0.5, loss_coeff_PoG_cm_final=0.01)),
}


def make_eve(over, dtype=torch.float32):
    cfg = eve_amd.reset_standalone_config()
    cfg.import_json(os.path.join(REPO, 'configs', 'refine_net.json'))
    cfg.import_dict(dict(dict(refine_net_rnn_type='CGRU', eye_net_load_pretrained=False), **over))
    model = eve_amd.EVE(output_predictions=True)
    model.eye_net.compute_dtype = dtype
    model.refine_net.compute_dtype = dtype
    detweights.fill_module(model.eye_net, 0)
    detweights.fill_module(model.refine_net, 1)
    return model.cuda()


F64_TAGS = ('c3', 'joint', 'clstm', 'crnn', 'noskip', 'noaug')


def check_grads_against_float64_reference(tag, net, module):
    """Gradients of `module` against tests/golden/eve_grads_f64.npz: the REFERENCE's models.eve.EVE evaluated in float64
    (make_golden_eve_grads.py) -- the complete tensors of the representative parameters (relative L2), the norms of all others.
    Bound per parameter: max(1e-4, 4 x the largest deviation the reference's OWN float32 evaluation shows for that parameter
    over the fixture's six cases).  Why that and not "2 x its deviation in this case": the harness differentiates through
    softmax(100 h) and ~40 layers of ReLU / max-pool / adaptive-pool decisions, so a float32 evaluation is one DRAW -- a few
    flipped decisions re-route gradient -- and the reference's own draws for one parameter spread 5x over the cases
    (initial.0.weight: 4.2e-3 .. 2.3e-2).  Measured on the HIP float32 path (tools/dbg_eve_grads.py): 0.4 - 1.4 x the
    reference's deviation typically (3.6 x at worst, on a decoder convolution whose reference deviation is 1.6e-4), 7.6e-3 ..
    4.6e-2 on initial.0.weight.  Parameters the reference leaves without a gradient
    (frozen EyeNet, CLSTM's gates: refine_net.py:168-174) must have none."""
    fx = np.load(os.path.join(GOLDEN, 'eve_grads_f64.npz'))
    params = dict(module.named_parameters())
    for n in fx['%s_%s_dead' % (tag, net)]:
        if str(n):
            assert params[str(n)].grad is None, n
    key = '%s_%s_names' % (tag, net)
    if key not in fx.files:
        assert all(p.grad is None for p in params.values())
        return 0.0
    noise = {}                                           # parameter -> its largest reference float32 deviation over the cases
    for t in F64_TAGS:
        if '%s_%s_names' % (t, net) in fx.files:
            nm, dv, nr = fx['%s_%s_names' % (t, net)], fx['%s_%s_ref_f32_dev' % (t, net)], fx['%s_%s_norms' % (t, net)]
            for n, d, w in zip(nm, dv, nr):
                if w > 1e-9 * nr.max():
                    noise[str(n)] = max(noise.get(str(n), 0.0), float(d))
    names, norms = fx[key], fx['%s_%s_norms' % (tag, net)]
    scale = float(norms.max())
    worst = 0.0
    for n, want in zip(names, norms):
        g = params[str(n)].grad
        assert g is not None, n
        if want <= 1e-9 * scale:
            continue
        tol = max(1e-4, 4.0 * noise[str(n)])
        if net == 'eye_net':
            # (joint training: EyeNet's gradient arrives partly through RefineNet's INPUT gradient, which carries the 2 - 3 % the
            #  reference's float32 shows on RefineNet's first layers; two HIP runs measured 6e-3 and 2e-2 on conv1.weight)
            tol = max(tol, 3e-2)
        full = '%s_%s_grad_%s' % (tag, net, n)
        if full in fx.files:
            err = float((g.detach().cpu().double() - torch.from_numpy(fx[full]).double()).norm())
        else:
            err = abs(float(g.detach().double().norm()) - float(want))
        # relative to the parameter's own norm, plus 2e-5 of the network's LARGEST gradient norm (`scale`): a bias gradient is a sum
        # over pixels that cancels by three orders of magnitude (decoder_blocks.0.layers.3.bias: 0.19 where the network's largest
        # norm is 25.7), so the float32 rounding of its terms shows ~1e4 x amplified in its RELATIVE error, on either
        # implementation and with a sign that follows the summation order (1.9e-3 against the reference's own 2.3e-4 here once the
        # float32 convolutions cut their accumulation chains: profiles/r06_notes.md 13)
        assert err <= tol * float(want) + 2e-5 * scale, (tag, net, str(n), err / float(want), tol, scale)
        worst = max(worst, err / float(want) / tol)
    return worst


@pytest.mark.parametrize('tag', sorted(EVE_CASES))
def test_eve_matches_reference_golden(tag):
    """float32 instantiation of eve_amd.EVE on the GPU against the REFERENCE's EVE (fixture): every loss / metric scalar,
    full_loss, gaze within 1e-4 rad, PoG, and per-parameter gradient norms of both networks."""
    fx = np.load(os.path.join(GOLDEN, 'eve_harness.npz'))
    training, over = EVE_CASES[tag]
    model = make_eve(over)
    model.train(training)
    B, T = int(fx['B']), int(fx['T'])
    batch = {k: v.cuda() for k, v in detweights.eve_batch(B, T, seed=int(fx['seed']),
                                                          invalid_fraction=float(fx['invalid_fraction'])).items()}
    np.random.seed(0)
    out = model({'synthetic': batch} if training else batch, create_images=not training, current_epoch=0.0)
    scalars = [k[len(tag) + 1:] for k in fx.files if k.startswith(tag + '_') and fx[k].ndim == 0 and ('loss' in k or 'metric' in k)]
    assert set(scalars) == {k for k in out if k.startswith(('loss_', 'metric_', 'full_loss'))}
    for k in scalars:
        want = float(fx['%s_%s' % (tag, k)])
        assert abs(float(out[k].detach()) - want) <= 5e-4 * abs(want) + 1e-4, (k, float(out[k].detach()), want)
    for k in ('g_initial', 'g_final'):
        assert np.abs(out[k].detach().cpu().numpy() - fx['%s_%s' % (tag, k)]).max() < 1e-4, k       # radians (north star)
    for k, tol in (('PoG_px_initial', 0.05), ('PoG_cm_initial', 2e-3), ('PoG_px_final', 0.5), ('PoG_cm_final', 2e-2)):
        assert np.abs(out[k].detach().cpu().numpy() - fx['%s_%s' % (tag, k)]).max() < tol, k
    assert np.abs(batch['g'].cpu().numpy() - fx[tag + '_label_g']).max() < 1e-5
    assert np.abs(batch['heatmap_final'].cpu().numpy()[..., ::4, ::4] - fx[tag + '_label_heatmap_final']).max() < 1e-5
    if training:
        assert np.array_equal(batch['left_kappa_fake'].cpu().numpy(), fx[tag + '_kappa_left'])
        out['full_loss'].backward()
        # gradients: against the reference's own float64 evaluation, bounded by its own float32 deviation (round 4; the
        # float32 norm fixture of eve_harness.npz with a flat 3e-2 before)
        for net, mod in (('eye_net', model.eye_net), ('refine_net', model.refine_net)):
            check_grads_against_float64_reference(tag, net, mod)
    else:
        for k in ('initial_gaze_history', 'refined_gaze_history', 'initial_heatmap', 'final_heatmap', 'gt_heatmap'):
            assert np.abs(out[k].detach().cpu().numpy()[..., ::4, ::4] - fx['eval_' + k]).max() < 2e-3, k
    eve_amd.reset_standalone_config()


def test_eve_bf16_train_step_tracks_float32():
    """The bf16 instantiation (the one that is benchmarked): same harness, losses within bf16 noise of float32."""
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        model = make_eve(EVE_CASES['joint'][1], dtype=dt).train()
        batch = {k: v.cuda() for k, v in detweights.eve_batch(2, 4, seed=2).items()}
        np.random.seed(1)
        out = model({'s': batch}, current_epoch=0.0)
        out['full_loss'].backward()
        res[dt] = {k: float(v.detach()) for k, v in out.items() if torch.is_tensor(v) and v.dim() == 0}
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.refine_net.parameters())
    for k, v in res[torch.float32].items():
        assert abs(res[torch.bfloat16][k] - v) <= 0.08 * abs(v) + 0.05, (k, res[torch.bfloat16][k], v)
    eve_amd.reset_standalone_config()


def test_eve_trainer_steps_and_learns():
    """A few optimiser steps of the C3 pipeline (EyeNet frozen) through train.eve_trainer: finite losses, RefineNet
    weights move, EyeNet's do not, and the heat-map loss goes down on a repeated batch."""
    from eve_amd import train
    model = make_eve({}, dtype=torch.bfloat16).train()
    cfg = eve_amd.get_config()
    tr = train.eve_trainer(model, cfg)
    batch = {k: v.cuda() for k, v in detweights.eve_batch(2, 4, seed=6).items()}
    eye0 = model.eye_net.fc_common[0].weight.detach().clone()
    ref0 = model.refine_net.final[0].weight.detach().clone()
    np.random.seed(3)
    hist = [float(tr.step(batch)['loss_ce_heatmap_final'].detach()) for _ in range(6)]
    assert all(np.isfinite(hist)) and hist[-1] < hist[0], hist
    assert torch.equal(eye0, model.eye_net.fc_common[0].weight.detach())
    assert not torch.equal(ref0, model.refine_net.final[0].weight.detach())
    eve_amd.reset_standalone_config()


def test_full_size_round_trips(hip):
    """Size-independent properties at BASELINE configs[2]'s frame count (N = 32 x 30 = 960): the two encode -> decode pairs
    of the harness.  (a) gaze geometry: the combined gaze towards a screen point, cast back through to_screen_coordinates,
    lands on that point; (b) heat-maps: soft-argmax of the Gaussian map drawn around a centre returns the centre (up to the
    reference's own grid conventions: maps are sampled at x = 15 i px, soft-argmax spreads i over 1920 / 127 px)."""
    N, screen = 960, (1920, 1080)
    b = detweights.eve_batch(32, 30, seed=17, with_screen=False)
    c = lambda k, *s: b[k].reshape(N, *s).cuda()
    g = torch.Generator().manual_seed(1)
    px = torch.stack([torch.rand(N, generator=g) * 1500 + 200, torch.rand(N, generator=g) * 800 + 150], dim=1)
    mm = (px * 0.288).cuda()
    o = 0.5 * (c('left_o', 3) + c('right_o', 3))
    gaze = hip.combined_gaze(o, mm, c('left_R', 3, 3), c('camera_transformation', 4, 4))
    _, back_mm, back_px, _ = hip.gaze_to_pog(gaze, o, c('left_R', 3, 3), c('inv_camera_transformation', 4, 4),
                                             c('pixels_per_millimeter', 2), screen)
    assert float((back_mm - mm).abs().max()) < 5e-3                      # mm, on a 553 x 311 mm screen
    assert float((back_px.cpu() - px).abs().max()) < 2e-2
    maps = hip.make_heatmaps(px.cuda(), 5.0, (72, 128), screen)
    got, _ = hip.soft_argmax_fwd(maps, screen)
    # expected position under the reference's conventions: centre at heat-map column cx = px / 15 -> pixel 1920 cx / 127
    want = torch.stack([px[:, 0] / 15.0 * (1920.0 / 127.0), px[:, 1] / 15.0 * (1080.0 / 71.0)], dim=1)
    assert float((got.cpu() - want).abs().max()) < 1.0
    assert float((got.cpu() - px).abs().max()) < 16.0                     # within one heat-map cell of the centre itself


def _oracle_gaze_float64(ocfg, batch, seed):
    """g_initial / g_final of the CPU oracle evaluated in float64 on the same float32-representable inputs and weights."""
    from oracle.eye_net import EyeNet as OracleEyeNet
    from oracle.refine_net import RefineNet as OracleRefineNet
    torch.set_default_dtype(torch.float64)
    try:
        oeye = detweights.fill_module(OracleEyeNet(ocfg), 0).double()
        oref = detweights.fill_module(OracleRefineNet(ocfg), 1).double()
        b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
        np.random.seed(seed)
        with torch.no_grad():
            _, inter, _ = oracle_eve.eve_forward(oeye, oref, dict(b64), ocfg, True)
        return {k: inter[k].detach().double() for k in ('g_initial', 'g_final')}
    finally:
        torch.set_default_dtype(torch.float32)


@pytest.mark.parametrize('over', [dict(refine_net_rnn_type='CLSTM'), dict(refine_net_rnn_type='CRNN'),
                                  dict(refine_net_rnn_type='CGRU', refine_net_use_skip_connections=False),
                                  dict(refine_net_rnn_type='CGRU', refine_net_do_offset_augmentation=False)],
                         ids=lambda o: '-'.join('%s=%s' % (k.replace('refine_net_', ''), v) for k, v in o.items()))
def test_eve_config_variants_match_oracle(over):
    """The refine_net.json pipeline as shipped (CLSTM: the cell whose output never reaches the decoder, refine_net.py:168-174)
    and its other switches, float32 on the GPU against the CPU oracle: every scalar, gaze within 1e-4 rad, gradients."""
    from oracle.eye_net import EyeNet as OracleEyeNet
    from oracle.refine_net import RefineNet as OracleRefineNet
    json_path = os.path.join(REPO, 'configs', 'refine_net.json')
    cfg = eve_amd.reset_standalone_config()
    cfg.import_json(json_path)
    cfg.import_dict(dict(eye_net_load_pretrained=False, **over))
    ocfg = OracleConfig(json_path, eye_net_load_pretrained=False, **over)
    model = eve_amd.EVE(output_predictions=True)
    model.eye_net.compute_dtype = model.refine_net.compute_dtype = torch.float32
    detweights.fill_module(model.eye_net, 0); detweights.fill_module(model.refine_net, 1)
    model = model.cuda().train()
    oeye, oref = detweights.fill_module(OracleEyeNet(ocfg), 0), detweights.fill_module(OracleRefineNet(ocfg), 1)
    for p in oeye.parameters():
        p.requires_grad = False
    batch = detweights.eve_batch(2, 3, seed=23, invalid_fraction=0.2)
    np.random.seed(2)
    want, winter, _ = oracle_eve.eve_forward(oeye, oref, dict(batch), ocfg, True)
    # The oracle's own error bar: the same algorithm on the same (float32-representable) inputs and weights in float64.  The
    # refinement amplifies a perturbation of g_initial ~20 x into g_final (soft-argmax at temperature 100 through a recurrent cell):
    # the float32 CPU evaluation itself sits up to 3.3e-5 rad from its float64 evaluation (no-skip variant), a third of the
    # tolerance, so the GPU result is held to 1e-4 against the float64 evaluation, and against the float32 one with that
    # evaluation's own deviation as slack (profiles/r06_notes.md 13).
    winter64 = _oracle_gaze_float64(ocfg, batch, seed=2)
    np.random.seed(2)
    got = model({'s': {k: v.cuda() for k, v in batch.items()}}, current_epoch=0.0)
    assert {k for k in got if k.startswith(('loss_', 'metric_'))} == set(want.keys()) - {'full_loss'}
    for k, v in want.items():
        assert abs(float(got[k].detach()) - float(v.detach())) <= 5e-4 * abs(float(v.detach())) + 1e-4, (k, float(got[k].detach()), float(v.detach()))
    for k in ('g_initial', 'g_final'):
        g = got[k].detach().cpu().double()
        bar = float((winter[k].detach().double() - winter64[k]).abs().max())
        assert bar < 5e-5, (k, bar)
        assert float((g - winter64[k]).abs().max()) < 1e-4, k
        assert float((g - winter[k].detach().double()).abs().max()) < 1e-4 + bar, k
    got['full_loss'].backward()
    # gradients: against the REFERENCE's float64 evaluation of this variant (eve_grads_f64.npz), bounded per parameter by the
    # reference's own float32 deviation (round 4; 1e-1 against the float32 CPU oracle before).  CLSTM: the cell's gates get no
    # gradient (its output never reaches the decoder, refine_net.py:168-174) -- the fixture lists them as dead.
    vtag = {'CLSTM': 'clstm', 'CRNN': 'crnn'}.get(over['refine_net_rnn_type'])
    if vtag is None:
        vtag = 'noskip' if over.get('refine_net_use_skip_connections') is False else 'noaug'
    check_grads_against_float64_reference(vtag, 'refine_net', model.refine_net)
    assert all(p.grad is None for p in model.eye_net.parameters())
    eve_amd.reset_standalone_config()


_CONFIGS4 = {}


@pytest.mark.parametrize('half', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
def test_configs4_long_sequence_large_patches_whole_pipeline(half):
    """BASELINE configs[4] as one case: T = 120 frames, 256 x 256 eye patches, EyeNet + RefineNet (CGRU) together, the
    conv-GRU / GRU hidden state carried on-chip across all 120 frames, in BOTH 16-bit instantiations of the kernels:
    float16 (what configs[4] names; static loss scale in the Trainer) and bfloat16 (the headline configs[1] format).  The
    reference is float32 only, so: (1) float32 forward of the whole pipeline against the CPU oracle -- gaze within 1e-4
    rad, refined heat-map PoG within half a pixel; (2) one 16-bit optimiser step of both networks at that size through
    train.eve_trainer -- finite losses and gradients, weights move, 16-bit gaze within the format's envelope of float32
    (float16's 11-bit significand must land closer than bfloat16's 8-bit one)."""
    from eve_amd import train
    from oracle.eye_net import EyeNet as OracleEyeNet
    from oracle.refine_net import RefineNet as OracleRefineNet
    json_path = os.path.join(REPO, 'configs', 'refine_net.json')
    over = dict(refine_net_rnn_type='CGRU', eye_net_load_pretrained=False, eye_net_frozen=False, loss_coeff_g_ang_initial=1.0,
                loss_coeff_pupil_size=1.0, refine_net_do_offset_augmentation=False)
    B, T, size = 1, 120, 256
    batch = detweights.eve_batch(B, T, seed=41, invalid_fraction=0.1)
    patches = detweights.eyenet_batch(B, T, size=size, seed=42)
    for k in ('left_eye_patch', 'right_eye_patch'):
        batch[k] = patches[k]
    if 'winter' not in _CONFIGS4:                       # the CPU oracle's forward is shared by the two cases
        ocfg = OracleConfig(json_path, **over)
        oeye, oref = detweights.fill_module(OracleEyeNet(ocfg), 0), detweights.fill_module(OracleRefineNet(ocfg), 1)
        with torch.no_grad():
            _, winter, _ = oracle_eve.eve_forward(oeye, oref, dict(batch), ocfg, False)
        _CONFIGS4['winter'] = {k: winter[k].detach() for k in ('g_initial', 'g_final', 'PoG_px_final')}
    winter = _CONFIGS4['winter']
    res = {}
    for dt in (torch.float32, half):
        if dt == torch.float32 and 'f32' in _CONFIGS4:
            res[dt] = _CONFIGS4['f32']
            continue
        model = make_eve(over, dtype=dt)
        dbatch = {k: v.cuda() for k, v in batch.items()}
        with torch.no_grad():
            model.eval()
            got = model(dict(dbatch), current_epoch=0.0)
        res[dt] = {k: got[k].detach().float().cpu() for k in ('g_initial', 'g_final', 'PoG_px_final')}
        if dt == torch.float32:
            _CONFIGS4['f32'] = res[dt]
        else:
            cfg = eve_amd.get_config()
            tr = train.eve_trainer(model.train(), cfg)
            assert len(tr.modules) == 2
            assert tr.loss_scale == (1024.0 if dt == torch.float16 else 1.0)
            before = tr.fp.flat.clone()
            terms = tr.step(dbatch)
            torch.cuda.synchronize()
            assert all(bool(torch.isfinite(v).all()) for v in terms.values() if torch.is_tensor(v))
            assert bool(torch.isfinite(tr.fp.grad).all()) and float(tr.fp.grad.abs().max()) > 0
            assert float((tr.fp.flat - before).abs().max()) > 0
    f32, h16 = res[torch.float32], res[half]
    for k in ('g_initial', 'g_final'):
        assert float((f32[k] - winter[k]).abs().max()) < 1e-4, k
    assert float((f32['PoG_px_final'] - winter['PoG_px_final']).abs().max()) < 0.5
    dev = float((h16['g_initial'] - f32['g_initial']).abs().max())
    print('%s gaze vs float32: %.3e rad' % (half, dev))
    assert dev < (0.08 if half == torch.bfloat16 else 0.02)
    # the recurrences matter at this length: the refined estimate at the last frame differs from the first frame's
    assert float((winter['PoG_px_final'][:, -1] - winter['PoG_px_final'][:, 0]).abs().max()) > 1.0
    eve_amd.reset_standalone_config()


def _configs4_clip(seed):
    batch = detweights.eve_batch(1, 120, seed=seed, invalid_fraction=0.1)
    patches = detweights.eyenet_batch(1, 120, size=256, seed=seed + 1)
    for k in ('left_eye_patch', 'right_eye_patch'):
        batch[k] = patches[k]
    return batch


def test_configs4_at_the_bench_batch_runs_the_kernels_the_bench_dispatches_and_equals_the_oracle_checked_clip():
    """configs[4]'s parity case above is B = 1, and kernel selection depends on the image count (register-
    resident InstanceNorm on 16-32 Ki-vector planes, the stem weight gradient in chunks below 2 GiB, conv3x3_wg8_kernel<2,4,16>
    only fill the chip from a few hundred images on).  Clips are independent units (InstanceNorm per frame, recurrences per
    clip), so: B = 4 x T = 120 x 256 x 256 (960 patches per eye: bench.py's c5 is B = 8) whose clip 0 IS the oracle-checked clip
    of the B = 1 case -- float32: clip 0's gaze / heat-map PoG equal the B = 1 run to float rounding (1e-5 rad / 0.05 px, hence
    the oracle to 1e-4); fp16 (what configs[4] names): within the format's envelope of float32; and the data-parallel identity
    grad(4 clips) = mean of the 4 single-clip gradients for the fp16 train step (both networks trained)."""
    from eve_amd import train
    over = dict(refine_net_rnn_type='CGRU', eye_net_load_pretrained=False, eye_net_frozen=False, loss_coeff_g_ang_initial=1.0,
                loss_coeff_pupil_size=1.0, refine_net_do_offset_augmentation=False)
    clips = [_configs4_clip(41 + 2 * i) for i in range(4)]
    full = {k: (torch.cat([c[k] for c in clips], dim=0) if torch.is_tensor(clips[0][k]) else clips[0][k]) for k in clips[0]}
    keys = ('g_initial', 'g_final', 'PoG_px_final')

    def forward(dt, batch):
        model = make_eve(over, dtype=dt)
        with torch.no_grad():
            model.eval()
            got = model({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}, current_epoch=0.0)
        return {k: got[k].detach().float().cpu() for k in keys}
    one = _CONFIGS4.get('f32') or forward(torch.float32, clips[0])
    four = forward(torch.float32, full)
    for k in ('g_initial', 'g_final'):
        assert float((four[k][:1] - one[k]).abs().max()) < 1e-5, k
    assert float((four['PoG_px_final'][:1] - one['PoG_px_final']).abs().max()) < 0.05
    h16 = forward(torch.float16, full)
    dev = float((h16['g_initial'] - four['g_initial']).abs().max())
    print('fp16 gaze vs float32 at B = 4: %.3e rad' % dev)
    assert dev < 0.02

    def grads(batch):
        model = make_eve(over, dtype=torch.float16)
        tr = train.eve_trainer(model.train(), eve_amd.get_config())
        tr.fp.grad.zero_()
        terms = tr._forward_backward({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()})
        torch.cuda.synchronize()
        assert bool(torch.isfinite(terms['full_loss']).all())
        return tr.fp.grad.detach().float().clone() / tr.loss_scale
    whole = grads(full)
    parts = sum(grads(c) for c in clips) / 4.0
    assert bool(torch.isfinite(whole).all()) and float(whole.abs().max()) > 0
    rel = float((whole - parts).norm() / parts.norm())
    print('fp16 grad(4 clips) vs mean of single-clip gradients: %.3e' % rel)
    assert rel < 2e-2, rel
    eve_amd.reset_standalone_config()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_eve_trainer_hipgraph_replay_equals_eager_steps(dtype):
    """configs[2] through train.eve_trainer(use_graph=True): the whole step (label synthesis, offset augmentation, frozen
    EyeNet forward, geometry, heat-maps, RefineNet forward + backward, soft-argmax, the 31 losses, clip, Adam) replays as one
    hipGraph.  The reference draws kappa_fake on the host inside forward (eve.py:463-479); the graph path makes the SAME draw
    (same numpy stream, one per step) before each replay into fixed device buffers.  Three steps eager vs three steps
    captured + replayed from identical weights and RNG state: every step's losses agree, the augmented gaze of each step is
    the same draw, and the weights end up equal to the noise of the weight-gradient atomics."""
    from eve_amd import train
    batch = {k: v.cuda() for k, v in detweights.eve_batch(2, 3, seed=23, invalid_fraction=0.1).items()}
    runs = {}
    for mode in ('eager', 'graph'):
        model = make_eve({}, dtype=dtype).train()
        cfg = eve_amd.get_config()
        cfg.import_dict({'base_learning_rate': 1e-6})       # (Adam's ~lr * sign(g) first steps: keep the comparison about plumbing)
        tr = train.eve_trainer(model, cfg, use_graph=(mode == 'graph'))
        np.random.seed(7)
        log = []
        first_grad = None
        for _ in range(3):
            terms = tr.step(batch)
            log.append({k: float(terms[k].detach()) for k in ('full_loss', 'loss_ce_heatmap_final', 'metric_euc_PoG_px_initial',
                                                              'metric_euc_PoG_px_final')})
            if first_grad is None:
                first_grad = tr.fp.grad.clone()         # (graph mode: capture + the first replay have run; this is step 1's)
        torch.cuda.synchronize()
        assert tr.optimizer_state()['steps_taken'] == 3
        runs[mode] = (log, first_grad)
        model.drop_static_kappa()
    # step 1 starts from identical weights: 2e-5.  Steps 2 and 3 start from weights that differ in the last bits (the order of the
    # weight-gradient atomics of step 1), and the refined point of gaze is a soft-argmax over softmax(100 h): one run in ~6 of the
    # float32 case moved `metric_euc_PoG_px_final` by 4.5e-5 relative (610.454 vs 610.482 px) -- 2e-4 for those steps
    for i, (a, b) in enumerate(zip(runs['eager'][0], runs['graph'][0])):
        tol = (2e-5 if i == 0 else 2e-4) if dtype == torch.float32 else 2e-2
        for k in a:
            assert abs(a[k] - b[k]) <= tol * max(1.0, abs(a[k])), (i, k, a[k], b[k])
    # the three draws differ from one another (the augmentation is live) ...
    assert len({round(s['metric_euc_PoG_px_initial'], 3) for s in runs['graph'][0]}) == 3
    # ... and the FIRST step's gradient is the eager one's up to the weight-gradient atomics' order (later steps' gradients are
    # not comparable: ulp-level weight differences move these ill-conditioned gradients by per cents -- 3 - 5 % measured in
    # float32, 38 % in bf16 -- see check_grads_against_float64_reference; the per-step LOSSES above are the check for steps 2, 3)
    ge, gg = runs['eager'][1], runs['graph'][1]
    assert float((ge - gg).norm() / ge.norm()) < (1e-4 if dtype == torch.float32 else 2e-2)
    eve_amd.reset_standalone_config()
