"""CPU, world_size 2 over gloo: the bucketed gradient all-reduce of eve_amd.parallel / the flat trainer gives the same
update as one process on the concatenated batch (clips are independent units, losses are per-clip means, equal local
batches).  The same workers run on the GPU with the real kernels in tests/test_gpu_data_parallel.py."""
import dp_common


def test_two_rank_step_equals_single_process_on_the_global_batch(tmp_path):
    a = dp_common.run_and_compare(str(tmp_path), 'eyenet', 'cpu', 'fp32')
    assert a['buckets'] >= 3


def test_two_rank_per_frame_contract_ships_complete_gradients(tmp_path):
    """The reference's per-frame contract uses every trunk weight 2 T times per step; its gradient is written in place
    into the flat buffer by every one of those backward passes, and the bucket all-reduce must not leave before the last
    (ops._note_use / _notify_grad_ready): the averaged gradient equals the single-process one."""
    dp_common.run_and_compare(str(tmp_path), 'eyenet_per_frame', 'cpu', 'fp32')


# ---- BASELINE configs[3]: the whole EyeNet + RefineNet pipeline (eve_amd.EVE) data-parallel -------------------------
def test_two_rank_eve_pipeline_step_equals_single_process(tmp_path):
    dp_common.run_and_compare(str(tmp_path), 'eve', 'cpu', 'fp32')
