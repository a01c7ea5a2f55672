"""CPU, build container only: the reference-side binding of INTEGRATION.md is EXECUTED -- the reference's own
`models.eve.EVE` with `EyeNet` / `RefineNet` rebound to the drop-ins reproduces the reference's own run
(tests/golden/eve_harness.npz).  Runs tests/reference_binding_check.py in a fresh process (it stubs modules and loads the
reference's `core` singleton, which must not leak into this test process).  Skipped where /root/reference is absent."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir('/root/reference/src'), reason='needs the reference checkout (build container)')
def test_reference_eve_runs_on_the_drop_in_modules():
    p = subprocess.run([sys.executable, os.path.join(HERE, 'reference_binding_check.py')], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0 and 'binding ok' in p.stdout, p.stdout[-4000:]
