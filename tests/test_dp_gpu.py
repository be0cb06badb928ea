"""GPU: the data-parallel path with 2 ranks on the one card of the GPU box (gloo on device tensors) - the real HIP
Predictor step on every rank, GradSink listener -> GradSync hook, bucket all-reduces overlapped with backward on the side
stream, SyncBatchNorm, train -> eval -> train, random-context batches - against one process on the whole batch
(tools/dp_check.py).  The two ranks are started by tests/conftest.py at session start, BEFORE this process touches the GPU
(a process that has initialised the GPU must not spawn GPU programs on this pool); this test only collects the result."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def _jobs(request):
    job = getattr(request.config, "_npvp_dp_job", None)
    if job is None:
        pytest.skip("the 2-rank jobs are only started for `-m gpu` sessions on a box with a GPU")
    proc, log_path = job
    try:
        rc = proc.wait(timeout=900)
    except Exception:
        proc.kill()
        raise
    read = lambda name: open(f"{log_path}.{name}").read() if os.path.exists(f"{log_path}.{name}") else ""
    if not getattr(request.config, "_npvp_dp_times_shown", False):
        request.config._npvp_dp_times_shown = True
        print("\n[dp jobs] " + read("times").replace("\n", "; "))
    return rc, read


def test_two_ranks_equal_single_process(request):
    rc, read = _jobs(request)
    log = read("dp_check")
    assert "[dp_check] OK" in log, f"tools/dp_check.py failed (rc={rc}):\n{log[-3000:]}"
    assert log.count("grad rel-L2") == 2, log[-2000:]


def test_four_ranks_equal_single_process(request):
    """the same check with four ranks sharing the card (an 8-sample global batch, 2 per rank)"""
    rc, read = _jobs(request)
    log = read("dp_check4")
    assert "[dp_check] OK" in log, f"tools/dp_check.py with 4 ranks failed (rc={rc}):\n{log[-3000:]}"
    assert log.count("grad rel-L2") == 2 and "world=4" in log, log[-2000:]


def test_bench_two_ranks(request):
    """bench.py's own N > 1 branch (the one the driver's scaling runs take) with 2 ranks on this card over gloo: the BASELINE
    multi-GPU shard workload c4 (8 clips per rank), rank 0's JSON line."""
    rc, read = _jobs(request)
    log = read("bench2")
    lines = [l for l in log.splitlines() if l.startswith("{")]
    assert lines, f"bench.py --gpus 2 printed no JSON line (rc={rc}):\n{log[-3000:]}"
    r = json.loads(lines[-1])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["value"] > 0 and r["scaling"] == "weak"
    assert r["config"]["global_batch"] == 16 and "c4" not in r["config"]["parallelism"] and r["config"]["parallelism"] == "dp2"
    assert r["roofline"] is not None and r["roofline"]["achieved"] > 0
    assert abs(r["value"] - 2 * 8 * 20 / (r["ms_per_step"] * 1e-3)) < 1e-2 * r["value"]
    # the N > 1 record says how many ranks really reduce together, how much of the all-reduce is exposed, and - beside
    # the weak-scaling `value` - the strong-scaling figure of BASELINE's data-parallel configuration for this GPU count, measured in the
    # same run against the same global batch on ONE GPU
    d = r["dp"]
    assert d["world_size"] == 2 and d["ranks_in_allreduce"] == 2.0 and d["exposed_allreduce_ms_per_step"] >= 0.0
    ss = r["strong_scaling"]
    assert ss["n_gpus"] == 2 and ss["ideal"] == 2 and "16 clips over 2 GPUs" in ss["config"]
    assert ss["frames_per_s"] > 0 and ss["whole_batch_on_1_gpu_frames_per_s"] > 0
    assert abs(ss["speedup_vs_1_gpu"] - ss["frames_per_s"] / ss["whole_batch_on_1_gpu_frames_per_s"]) < 2e-3
    assert r["scaling_dp"]["strong_ratio"] == ss["speedup_vs_1_gpu"]
    # the step of this job was recorded as graph segments (--dp-graph always): the mode trial timed all three candidates
    assert "'segments'" in log and "'eager_fused'" in log and "graph_segments" in r["config"]["workload"] or "HIP-graph segments" in r["config"]["workload"]


def test_rccl_one_rank(request):
    """RCCL as far as one GPU allows: bench.py on ONE rank with backend nccl and NPVP_DP_FORCE=1 - ProcessGroupNCCL
    initialised, the model broadcast, SyncBatchNorm2d converted and reducing on its own communicator, GradSync's buckets reduced by
    all_reduce(async_op=True) on the side stream while backward runs, work.wait() + finish() before the clip - all on real RCCL.
    A group of one leaves every value unchanged, so the step must equal the plain one-process step's loss scale and finish."""
    rc, read = _jobs(request)
    log = read("rccl1")
    lines = [l for l in log.splitlines() if l.startswith("{")]
    assert lines, f"bench.py on one RCCL rank printed no JSON line (rc={rc}):\n{log[-3000:]}"
    r = json.loads(lines[-1])
    assert r["n_gpus"] == 1 and r["value"] > 0 and r["config"]["final_loss"] == r["config"]["final_loss"]
    d = r["dp"]
    assert d["backend"] == "nccl" and d["buckets"] >= 6 and d["last_bucket_mb"] <= 16.0
    # step 1 learns the contribution counts and reduces everything in finish(); from step 2 on every bucket is launched from a hook
    # (eager trial, the recording's warm-up, then replays: every bucket once per step whoever launches it)
    assert d["allreduces_launched"] % d["buckets"] == 0 and d["allreduces_launched"] >= d["buckets"] * 5, d
    assert d["exposed_allreduce_ms_per_step"] >= 0.0 and d["world_size"] == 1 and d["ranks_in_allreduce"] == 1.0
    # round 6: the timed steps of this job replay the data-parallel step as graph segments with the collectives (real RCCL calls)
    # issued eagerly between them - the host's share of a step is a few graph launches (a quarter of the step, not all of it)
    assert "HIP-graph segments" in r["config"]["workload"], r["config"]["workload"]
    assert r["config"]["host_enqueue_ms_per_step"] <= 12.0, r["config"]["host_enqueue_ms_per_step"]     # (eager: 25 - 35 ms)


def test_segmented_step_equals_eager_bit_for_bit(request):
    """tools/dp_segments_check.py: the data-parallel step recorded as a chain of HIP-graph segments cut at every collective
    (trainer.StepTape; bucket all-reduces on the side stream, SyncBatchNorm's statistics forward and backward, the closing wait) and
    replayed for K - 2 steps == K eager data-parallel steps: parameters, Adam moments, the last gradient, the loss and the step count
    are EQUAL (torch.equal) - on one real RCCL rank, and on two gloo ranks that share the card (there the collectives really
    exchange data between the segments)."""
    rc, read = _jobs(request)
    for name, ranks in (("seg1", 1), ("seg2", 2)):
        log = read(name)
        assert "[dp_segments_check] OK" in log, f"tools/dp_segments_check.py ({name}) failed (rc={rc}):\n{log[-3000:]}"
        assert log.count("'params': True, 'adam_m': True, 'adam_v': True, 'grad': True, 'loss': True, 'step_count': True") == ranks, log[-2000:]
    one = read("seg1")
    assert "backend=nccl" in one
    host = float(one.split("host_ms_min=")[1].split()[0])
    # (the runtime's safe replay mode enqueues a graph's ~1 000 packets from the host: ~3 ms per step + ~0.15 ms per segment; with the
    #  packet-capture path this was 3 ms in all - and wrong results.  Either way a fraction of the 31 ms the device needs)
    assert host <= 10.0, f"a replayed data-parallel step costs the host {host} ms"
    # the same job with the runtime's prepared-packet replay mode on (NPVP_GRAPH_PACKET_CAPTURE=1): the segments contain no memset node
    # (which that mode does not execute reliably on ROCm 7.2), it is bit-identical too, and the host pays less per replay
    fastlog = read("seg1f")
    assert "[dp_segments_check] OK" in fastlog and "'memset'" not in fastlog.split("nodes of the segments:")[1].split("\n")[0], fastlog[-3000:]
    assert fastlog.count("'params': True, 'adam_m': True, 'adam_v': True, 'grad': True, 'loss': True, 'step_count': True") == 1, fastlog[-2000:]
    print(f"\n[dp segments] host ms per replayed step (best): node-by-node mode {host}, prepared packets {float(fastlog.split('host_ms_min=')[1].split()[0])}")


def test_replayed_step_survives_caller_allocations(request):
    """tools/graph_alloc_hazard.py: a caller that allocates device memory after the capture and writes it between two replays (a
    16-float tensor, a fresh 1 MiB buffer, `loss.clone()` per step, a new batch tensor copied into the static inputs) leaves the
    replayed run ON the eager trajectory - losses and parameters, bit for bit - in the package's default runtime mode (hazard0-2) and
    with the runtime's prepared-packet replay switched on (fast0-2: NPVP_GRAPH_PACKET_CAPTURE=1; fast2 = the stochastic NPVP-S predictor).  The second holds because the step
    has no memset node: on ROCm 7.2 that replay mode does not execute memset nodes reliably (until round 6 the step had two
    kinds of them and computed wrong steps there: DESIGN 7, profiles/r06_graph_alloc_hazard.txt)."""
    rc, read = _jobs(request)
    for i in range(3):
        log = read(f"hazard{i}")
        assert "[graph_alloc_hazard] packet capture off" in log and "bit-equal True" in log and log.rstrip().endswith("OK"), log[-1500:]
    for i in range(3):
        log = read(f"fast{i}")
        assert "[graph_alloc_hazard] packet capture ON" in log and "bit-equal True" in log and log.rstrip().endswith("OK"), log[-1500:]
        assert "'memset'" not in log.split("nodes of the captured step:")[1].split("\n")[0], log[-1500:]


def test_library_exchange_one_rank(request):
    """include/npvp_hip.h npvp_dp_*: the same one-rank RCCL step with NPVP_DP_COMM=c - rank 0 draws the communicator id
    (npvp_dp_unique_id), the process group carries it, npvp_dp_init joins, every gradient bucket is reduced by
    npvp_dp_allreduce_async on GradSync's side stream and the compute stream is ordered behind them by npvp_dp_wait."""
    rc, read = _jobs(request)
    log = read("rccl1c")
    lines = [l for l in log.splitlines() if l.startswith("{")]
    assert lines, f"bench.py on one rank with NPVP_DP_COMM=c printed no JSON line (rc={rc}):\n{log[-3000:]}"
    r = json.loads(lines[-1])
    d = r["dp"]
    assert d["comm"] == "c" and d["backend"] == "nccl" and d["buckets"] >= 6
    assert "'eager_fused'" in log, "the forced second leg of the mode trial (one launch per layer backward) did not run"
    assert d["allreduces_launched"] % d["buckets"] == 0 and d["allreduces_launched"] >= d["buckets"] * 5, d
    assert r["value"] > 0 and r["config"]["final_loss"] == r["config"]["final_loss"]
