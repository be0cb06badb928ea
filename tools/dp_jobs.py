"""The multi-process GPU jobs of `pytest -m gpu`, run one after the other by ONE child of the test session (started by
tests/conftest.py before the test process touches the GPU; this wrapper itself never does):
  1. tools/dp_check.py        - 2 ranks on one card (gloo on device tensors): the real data-parallel step == one process
  2. bench.py --gpus 2        - the N > 1 branch of the benchmark itself (rank-strided model build, GradSync, SyncBatchNorm,
                                MAX-over-ranks timing, the JSON line) on the BASELINE multi-GPU shard workload (c4: 8 clips)
  3. bench.py --gpus 1 on RCCL - one rank, backend nccl, NPVP_DP_FORCE=1: the data-parallel path on real RCCL (see below)
  4. the same with NPVP_DP_COMM=c - the gradient buckets through the library's own npvp_dp_* exchange
  5. tools/dp_check.py with 4 ranks on the card
  7. tools/graph_alloc_hazard.py - replayed steps with a caller allocating / copying batches between the replays, in both replay modes
                                of the runtime
  6. tools/dp_segments_check.py - the data-parallel step replayed as HIP-graph segments with the collectives issued eagerly between
                                them (trainer.StepTape) against the eager data-parallel step, bit for bit: on one RCCL rank and on two
                                gloo ranks sharing the card
Each job's output goes to <log>.<name>; the wrapper's exit code is the first failure's."""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
log = sys.argv[1]
env = dict(os.environ, NPVP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
run = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1"]
jobs = [("dp_check", run + ["--master-port", "29531", os.path.join(ROOT, "tools", "dp_check.py")]),
        ("bench2", run + ["--master-port", "29532", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c4", "--steps", "3",
                          "--warmup", "1", "--dp-fused-trial", "always", "--dp-graph", "always", "--trial-steps", "2", "--no-cpu-baseline"])]
# 3. RCCL itself, as far as one GPU allows: ONE rank, backend nccl, NPVP_DP_FORCE=1 = the whole data-parallel code path on a group
#    of one (ProcessGroupNCCL init, model broadcast, SyncBatchNorm2d's all-reduces on their own communicator, GradSync's bucket
#    all_reduce(async_op=True) on the side stream + work.wait() + finish()) inside the benchmark's own step
run1 = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1"]
jobs.append(("rccl1", run1 + ["--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "c4", "--steps", "3",
                              "--warmup", "2", "--no-secondary", "--no-cpu-baseline", "--dp-graph", "always", "--trial-steps", "3"]))
# 4. the same one-rank RCCL step with the gradient buckets on the LIBRARY's exchange (NPVP_DP_COMM=c: npvp_dp_unique_id / npvp_dp_init /
#    npvp_dp_allreduce_async / npvp_dp_wait of include/npvp_hip.h) instead of ProcessGroupNCCL's all_reduce
#    (and with the mode trial's second leg forced - one launch per layer backward beside the gradient stream: GradSync must see the
#    same per-parameter contribution counts in both legs)
jobs.append(("rccl1c", run1 + ["--master-port", "29534", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "c4", "--steps", "3",
                               "--warmup", "2", "--no-secondary", "--no-cpu-baseline", "--dp-fused-trial", "always", "--dp-graph", "never",
                               "--trial-steps", "3"]))
# 5. tools/dp_check.py once more with FOUR ranks on the card (gloo on device tensors; 8-sample global batch, 2 per rank)
run4 = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1"]
jobs.append(("dp_check4", run4 + ["--master-port", "29535", os.path.join(ROOT, "tools", "dp_check.py")]))
# 6. the segmented replay of the data-parallel step == the eager data-parallel step, bit for bit (parameters, Adam state, gradients, loss)
jobs.append(("seg1", run1 + ["--master-port", "29536", os.path.join(ROOT, "tools", "dp_segments_check.py")]))
jobs.append(("seg2", run + ["--master-port", "29537", os.path.join(ROOT, "tools", "dp_segments_check.py")]))
#    ... and once more on the RCCL rank with the runtime's prepared-packet replay switched on: the segments carry no memset node
#    (SyncBatchNorm's statistics are the library's column sums), so that mode is exact too and the host pays less per replay
jobs.append(("seg1f", run1 + ["--master-port", "29538", os.path.join(ROOT, "tools", "dp_segments_check.py")]))
# 7. replayed steps and what a caller does between replays (tools/graph_alloc_hazard.py): in the package's default runtime mode
#    (hazard0-2) and with the runtime's prepared-packet replay switched on (fast0-2: the step has no memset node, which that mode
#    does not execute reliably on ROCm 7.2 - npvp_amd/__init__.py) losses and parameters must stay on the eager trajectory, bit for bit
for i in range(3):
    jobs.append((f"hazard{i}", [sys.executable, os.path.join(ROOT, "tools", "graph_alloc_hazard.py")]))
for i in range(3):
    jobs.append((f"fast{i}", [sys.executable, os.path.join(ROOT, "tools", "graph_alloc_hazard.py")]))
fast = {k: v for k, v in env.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}         # (inherited from a parent that imported npvp_amd)
envs = {"hazard0": dict(env, BETWEEN="tiny"), "hazard1": dict(env, BETWEEN="fill:0.001"), "hazard2": dict(env, BETWEEN="clone"),
        "fast0": dict(fast, NPVP_GRAPH_PACKET_CAPTURE="1", BETWEEN="inputs"), "fast1": dict(fast, NPVP_GRAPH_PACKET_CAPTURE="1", BETWEEN="tiny"),
        "fast2": dict(fast, NPVP_GRAPH_PACKET_CAPTURE="1", BETWEEN="inputs", STOCHASTIC="1"),     # (NPVP-S: one memcpy node, same noise in both runs)
        "seg1": dict(env, NPVP_DIST_BACKEND="nccl", NPVP_DP_FORCE="1"),
        "seg1f": dict(fast, NPVP_DIST_BACKEND="nccl", NPVP_DP_FORCE="1", NPVP_GRAPH_PACKET_CAPTURE="1"),
        "seg2": dict(env, SEG_CHECK_STEPS="4", SEG_CHECK_LAYERS="4"),
        "dp_check4": dict(env, DP_CHECK_SEED="12"),      # (clip seed 11 puts ONE unit of the 8-clip batch on a ReLU kink, counted: profiles/r06_dp_check_relu_kink.txt)
        "rccl1": dict(env, NPVP_DIST_BACKEND="nccl", NPVP_DP_FORCE="1"),
        "rccl1c": dict(env, NPVP_DIST_BACKEND="nccl", NPVP_DP_FORCE="1", NPVP_DP_COMM="c")}
import time
rc = 0
with open(f"{log}.times", "w") as tf:
    for name, cmd in jobs:
        t0 = time.time()
        with open(f"{log}.{name}", "w") as f:
            r = subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, env=envs.get(name, env), cwd=ROOT)
        tf.write(f"{name} {time.time() - t0:.1f} s rc={r.returncode}\n"); tf.flush()
        if r.returncode != 0 and rc == 0:
            rc = r.returncode
sys.exit(rc)
