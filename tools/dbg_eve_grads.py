import os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import eve_amd
from oracle import detweights
fx = np.load(os.path.join(REPO, 'tests', 'golden', 'eve_grads_f64.npz'))
JOINT = dict(eye_net_frozen=False, loss_coeff_PoG_cm_initial=0.002, loss_coeff_g_ang_initial=1.0, loss_coeff_pupil_size=1.0,
             loss_coeff_heatmap_mse_final=0.5, loss_coeff_PoG_cm_final=0.01)
CASES = {'c3': (dict(refine_net_rnn_type='CGRU'), (2, 4, 0, 0.25), 0), 'joint': (dict(refine_net_rnn_type='CGRU', **JOINT), (2, 4, 0, 0.25), 0),
         'clstm': (dict(refine_net_rnn_type='CLSTM'), (2, 3, 23, 0.2), 2), 'crnn': (dict(refine_net_rnn_type='CRNN'), (2, 3, 23, 0.2), 2),
         'noskip': (dict(refine_net_rnn_type='CGRU', refine_net_use_skip_connections=False), (2, 3, 23, 0.2), 2),
         'noaug': (dict(refine_net_rnn_type='CGRU', refine_net_do_offset_augmentation=False), (2, 3, 23, 0.2), 2)}
for tag, (over, (B, T, seed, inv), npseed) in CASES.items():
    cfg = eve_amd.reset_standalone_config()
    cfg.import_json(os.path.join(REPO, 'configs', 'refine_net.json'))
    cfg.import_dict(dict(eye_net_load_pretrained=False, **over))
    model = eve_amd.EVE(output_predictions=True)
    model.eye_net.compute_dtype = model.refine_net.compute_dtype = torch.float32
    detweights.fill_module(model.eye_net, 0); detweights.fill_module(model.refine_net, 1)
    model = model.cuda().train()
    batch = {k: v.cuda() for k, v in detweights.eve_batch(B, T, seed=seed, invalid_fraction=inv).items()}
    np.random.seed(npseed)
    out = model({'s': batch}, current_epoch=0.0)
    out['full_loss'].backward()
    print(tag, 'loss', float(out['full_loss']), 'ref f32', float(fx[tag + '_full_loss_f32']), 'ref f64', float(fx[tag + '_full_loss_f64']))
    for net in ('eye_net', 'refine_net'):
        key = '%s_%s_names' % (tag, net)
        if key not in fx.files: continue
        params = dict(getattr(model, net).named_parameters())
        names, norms, dev = fx[key], fx['%s_%s_norms' % (tag, net)], fx['%s_%s_ref_f32_dev' % (tag, net)]
        rows = []
        for n, want, d in zip(names, norms, dev):
            if want < 1e-9 * norms.max(): continue
            g = params[str(n)].grad
            full = '%s_%s_grad_%s' % (tag, net, n)
            if full in fx.files:
                err = float((g.detach().cpu().double() - torch.from_numpy(fx[full]).double()).norm()) / want
                rows.append((str(n), err, d, 'full'))
            else:
                rows.append((str(n), abs(float(g.double().norm()) - want) / want, d, 'norm'))
        fe = [r for r in rows if r[3] == 'full']
        ne = np.array([r[1] for r in rows if r[3] == 'norm']); nd = np.array([r[2] for r in rows if r[3] == 'norm'])
        print('  %s: FULL tensors: %s' % (net, ['%s hip %.2e ref %.2e' % (r[0].split('.')[-3] + '.' + r[0].split('.')[-2] if r[0].count('.') > 2 else r[0], r[1], r[2]) for r in fe]))
        print('  %s: norms: hip median %.2e max %.2e | ref dev (full-tensor L2) median %.2e max %.2e | worst ratio hip/(2 ref) %.2f' % (
            net, np.median(ne), ne.max(), np.median(nd), nd.max(), (ne / np.maximum(2 * nd, 1e-4)).max()))
eve_amd.reset_standalone_config()
