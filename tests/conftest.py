import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import npvp_amd  # noqa: E402,F401  (before anything initialises the HIP runtime: the package picks the runtime's graph-replay mode at import)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """`-m gpu` sessions: start the 2-rank jobs (tools/dp_jobs.py: the data-parallel rehearsal tools/dp_check.py, then
    bench.py --gpus 2) NOW, before this process initialises the GPU (torch.cuda.device_count() does not), and let
    tests/test_dp_gpu.py collect them.  3 GPU processes in all."""
    import subprocess
    import tempfile
    config = session.config
    config._npvp_dp_job = None
    config._npvp_larger_oracle = None
    mark = config.getoption("-m") or ""
    if "gpu" not in mark or "not gpu" in mark or os.environ.get("NPVP_SKIP_DP_TEST"):
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
    except Exception:
        return
    # the CPU-oracle side of test_against_oracle_larger (230 s of CPU work) runs BESIDE the GPU tests in a CPU-only child: it imports
    # torch and the oracle, never touches the GPU, and leaves one .pt per case for the test to pick up (tests/larger_oracle.py)
    if not os.environ.get("NPVP_SKIP_LARGER_WORKER"):
        odir = tempfile.mkdtemp(prefix="npvp_larger_oracle_")
        olog = open(os.path.join(odir, "worker.log"), "w")
        oproc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "larger_oracle.py"), odir], stdout=olog,
                                 stderr=subprocess.STDOUT, cwd=ROOT)
        config._npvp_larger_oracle = (oproc, odir)
    log = tempfile.NamedTemporaryFile(prefix="npvp_dp_jobs_", suffix=".log", delete=False)
    # tools/dp_jobs.py runs the 2-rank jobs one after the other (dp_check, then bench.py --gpus 2): at most 2 + 1 GPU processes
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "dp_jobs.py"), log.name], stdout=log,
                            stderr=subprocess.STDOUT, cwd=ROOT)
    config._npvp_dp_job = (proc, log.name)


def pytest_sessionfinish(session, exitstatus):
    job = getattr(session.config, "_npvp_larger_oracle", None)
    if job is not None and job[0].poll() is None:
        job[0].kill()              # (the exact child this session started)


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible and -m gpu was not asked for.  The tests that only COLLECT the
    multi-process jobs started at session start (tests/test_dp_gpu.py) run last: the jobs share the card with the session's own
    tests, and a collector that runs early just waits for them."""
    late = [it for it in items if it.fspath.basename == "test_dp_gpu.py"]
    if late:
        items[:] = [it for it in items if it.fspath.basename != "test_dp_gpu.py"] + late
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
