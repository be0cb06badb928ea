"""GPU: the data-parallel path with 2 ranks on the one card of the GPU box (gloo on device tensors) - the real HIP
Predictor step on every rank, GradSink listener -> GradSync hook, bucket all-reduces overlapped with backward on the side
stream, SyncBatchNorm, train -> eval -> train, random-context batches - against one process on the whole batch
(tools/dp_check.py).  The two ranks are started by tests/conftest.py at session start, BEFORE this process touches the GPU
(a process that has initialised the GPU must not spawn GPU programs on this pool); this test only collects the result."""
import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_equal_single_process(request):
    job = getattr(request.config, "_npvp_dp_job", None)
    if job is None:
        pytest.skip("the DP rehearsal is only started for `-m gpu` sessions on a box with a GPU")
    proc, log_path = job
    try:
        rc = proc.wait(timeout=600)
    except Exception:
        proc.kill()
        raise
    log = open(log_path).read()
    assert rc == 0 and "[dp_check] OK" in log, f"tools/dp_check.py failed (rc={rc}):\n{log[-3000:]}"
    assert log.count("grad rel-L2") == 2, log[-2000:]
