#!/usr/bin/env python
"""Executes the reference-side binding INTEGRATION.md describes, in the build container (needs /root/reference):

  * imports the reference's own `models.eve` (logging / IO stubs and the torchvision stand-in of
    tests/golden/make_golden.py, so that `models/eye_net.py:26` imports),
  * rebinds the two names `src/models/eve.py:36-37` imports -- `EyeNet`, `RefineNet` -- to `eve_amd.EyeNet` /
    `eve_amd.RefineNet` (what changing those two import lines does),
  * runs the REFERENCE's `EVE.__init__` / `EVE.forward` (per-frame calls `eve.py:108-111,146-147`, its own geometry,
    heat-maps, losses) on the deterministic batch, train mode, refine_net.json + CGRU,
  * and compares every scalar and prediction with tests/golden/eve_harness.npz (the reference's own run).

The drop-ins read the reference's config singleton here (`eve_amd.config.get_config()` finds `core.DefaultConfig`).
Kernels: tests/fake_kernels.py (CPU) -- this checks the binding, constructor / forward contracts and state_dict, not
the HIP arithmetic.  Run by tests/test_reference_binding.py in a fresh process; prints `binding ok`.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, 'golden'))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from make_golden import REF_SRC, import_reference
    config = import_reference()
    import eve_amd
    from eve_amd import config as eve_config
    from eve_amd import kernels
    from fake_kernels import FakeKernels
    from oracle import detweights
    kernels.set_default_kernels(FakeKernels())
    assert eve_config.get_config() is config, 'the drop-ins must read the reference singleton when `core` is loaded'
    config.import_json(os.path.join(REF_SRC, 'configs', 'refine_net.json'))
    config.override('refine_net_rnn_type', 'CGRU')
    config.override('eye_net_load_pretrained', False)
    import models.eve as ref_eve
    ref_eye_keys = list(ref_eve.EyeNet().state_dict().keys())
    ref_refine_keys = list(ref_eve.RefineNet().state_dict().keys())
    ref_eve.EyeNet, ref_eve.RefineNet = eve_amd.EyeNet, eve_amd.RefineNet         # eve.py:36-37
    model = ref_eve.EVE(output_predictions=True)
    assert type(model.eye_net) is eve_amd.EyeNet and type(model.refine_net) is eve_amd.RefineNet
    assert list(model.eye_net.state_dict().keys()) == ref_eye_keys
    assert list(model.refine_net.state_dict().keys()) == ref_refine_keys
    assert not any(p.requires_grad for p in model.eye_net.parameters())            # eve.py:58-60 froze the drop-in
    detweights.fill_module(model.eye_net, seed=0)
    detweights.fill_module(model.refine_net, seed=1)
    model.train()
    fx = np.load(os.path.join(HERE, 'golden', 'eve_harness.npz'))
    B, T = int(fx['B']), int(fx['T'])
    batch = detweights.eve_batch(B, T, seed=0, invalid_fraction=float(fx['invalid_fraction']))
    np.random.seed(0)
    out = model({'synthetic': batch}, current_epoch=0.0)
    checked = 0
    for k, v in out.items():
        key = 'c3_' + k
        if key not in fx.files or not isinstance(v, torch.Tensor):
            continue
        got, want = v.detach().numpy(), fx[key]
        if v.dim() == 0:
            np.testing.assert_allclose(got, want, rtol=5e-4, atol=1e-6, err_msg=k)
        elif 'PoG_px' in k:
            np.testing.assert_allclose(got, want, atol=0.05, err_msg=k)              # pixels on a 1920 x 1080 screen
        else:
            np.testing.assert_allclose(got, want, atol=2e-4, err_msg=k)
        checked += 1
    assert checked >= 30, checked
    out['full_loss'].backward()
    names, norms = fx['c3_refine_net_grad_names'], fx['c3_refine_net_grad_norms']
    params = dict(model.refine_net.named_parameters())
    for n, want in zip(names, norms):
        g = params[str(n)].grad
        if want < 0:
            assert g is None, n
        else:
            assert abs(float(g.double().norm()) - want) <= 2e-2 * want + 3e-5, (n, float(g.double().norm()), want)
    assert all(p.grad is None for p in model.eye_net.parameters())
    print('binding ok: %d outputs of the reference EVE.forward over the drop-in modules match the reference run' % checked)


if __name__ == '__main__':
    main()
