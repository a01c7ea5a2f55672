"""CPU oracle for the EVE hot path (EyeNet + RefineNet) -- TEST INFRASTRUCTURE ONLY.

This package is a plain-``torch`` fp32 restatement of the reference's
``src/models/{eye_net,refine_net,common}.py`` (and of the un-vendored
``torchvision==0.6.1`` ``models/resnet.py`` ResNet-18 it constructs at
``src/models/eye_net.py:48-50``).  It exists so that the HIP path can be
checked for parity; it is never the thing measured or shipped.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  Nothing under ``eve_amd/`` imports it, and the
product path raises when the HIP library is missing instead of falling back
to anything here.

Pinning status (see DESIGN.md "Oracle"):
  * RefineNet, CRNN/CLSTM/CGRU cells, the EyeNet head (fc_common, GRUCell,
    gaze/pupil heads) and the loss terms are pinned against the reference
    classes imported in the build container -- fixtures under
    ``tests/golden/`` made by ``tests/golden/make_golden.py``.
  * The ResNet-18(InstanceNorm) trunk arithmetic lives in torchvision 0.6.1,
    which is absent from /root/reference and from this image; the reference
    has no tests for it.  The trunk restatement (``oracle/resnet_in.py``)
    follows the published torchvision algorithm; it is exercised through the
    reference ``EyeNet`` class with the restated trunk injected as
    ``torchvision.models.resnet`` (that pins the plumbing around it) and pinned
    numerically by an INDEPENDENT implementation of the same architecture that
    the image does ship -- ``transformers.models.resnet`` with InstanceNorm2d,
    ``tests/golden/make_golden_trunk.py`` -> ``trunk_independent.npz``.  Not
    reference-held (nothing is, for this dependency), but not oracle-vs-oracle.
  * ``oracle/bf16_faithful.py`` is the same oracle with bfloat16 rounding at the
    tensors the HIP bf16 instantiation stores in bfloat16; with rounding off it
    reproduces the float32 oracle (tests/test_oracle_golden.py).
"""
