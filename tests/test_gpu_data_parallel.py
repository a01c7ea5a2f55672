"""GPU: data parallelism on the HIP path.  Two ranks share the one GPU of the test box (EVE_AMD_FORCE_DEVICE=0, gloo for
the collective: RCCL needs one device per rank), each running the product trainer with the real kernels on its clip of a
global batch: the ranks stay bit-identical, every gradient bucket is all-reduced exactly once per step (launched from the
in-place weight-gradient notifications and the autograd hooks while backward is still running), and the averaged gradient
/ the update equal a single process on the concatenated batch.  BASELINE configs[3]'s code path short of the RCCL
transport itself, which bench.py --gpus N exercises when the driver has a multi-GPU node."""
import pytest

import dp_common

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_two_ranks_one_gpu_eyenet_trainer(tmp_path, dtype):
    a = dp_common.run_and_compare(str(tmp_path), 'eyenet', 'cuda', dtype, grad_tol=1e-3 if dtype == 'fp32' else 5e-3)
    assert a['buckets'] >= 3


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_two_ranks_one_gpu_eve_pipeline_trainer(tmp_path, dtype):
    """configs[3]: EyeNet (frozen, forward only) + RefineNet/CGRU trained through eve_amd.EVE on both ranks."""
    dp_common.run_and_compare(str(tmp_path), 'eve', 'cuda', dtype, grad_tol=1e-3 if dtype == 'fp32' else 5e-3)


def test_two_ranks_one_gpu_per_frame_contract(tmp_path):
    dp_common.run_and_compare(str(tmp_path), 'eyenet_per_frame', 'cuda', 'fp32')


def test_two_ranks_one_gpu_hipgraph_replay_with_eager_collectives(tmp_path):
    """use_graph with several ranks: forward + backward are captured, the bucket all-reduces, the clip and Adam run
    eagerly behind every replay; three steps (capture + two replays) equal three single-process steps."""
    dp_common.run_and_compare(str(tmp_path), 'eyenet', 'cuda', 'bf16', use_graph=True)
