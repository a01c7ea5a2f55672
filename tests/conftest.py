import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _kernel_selection_is_the_default(request):
    """Every GPU test runs on the library's DEFAULT kernel selection (include/eve_hip.h eve_dispatch_config): a stray EVE_*
    variable in the environment, or an override a previous test leaked, would silently change which kernel a parity claim is
    about.  Printed once per session; tests that compare two kernels force the variant inside HipKernels.dispatch_override()."""
    if request.node.get_closest_marker('gpu') is None:
        yield
        return
    from eve_amd.kernels import default_kernels
    k = default_kernels()
    cur, dflt = k.dispatch_config().as_dict(), k.default_dispatch_config().as_dict()
    if not getattr(_kernel_selection_is_the_default, 'printed', False):
        _kernel_selection_is_the_default.printed = True
        print('\neve_dispatch_config: %s' % cur)
    assert cur == dflt, 'kernel selection differs from the defaults: %s' % {n: (cur[n], dflt[n]) for n in cur if cur[n] != dflt[n]}
    yield
    after = k.dispatch_config().as_dict()
    assert after == dflt, 'the test leaked a kernel-selection override: %s' % {n: after[n] for n in after if after[n] != dflt[n]}
