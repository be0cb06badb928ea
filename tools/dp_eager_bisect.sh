#!/bin/bash
# Why was the EAGER data-parallel step on one RCCL rank twice as slow as the plain step (c4 shard: 64 - 73 ms against 29, c2: 266
# against 233)?  One knob at a time, no profiler (the tracer's host cost hides it).  Run from the repo root on the GPU box:
#   bash tools/dp_eager_bisect.sh stages | env | queues          (W=c2 for the large workload)
# Round 6 finding (profiles/r06_dp_eager_bisect.txt): not the all-reduce, not the hooks, not the progress flushes - the step doubles
# as soon as ONE collective has made ProcessGroupNCCL create its communicator, IF the process also owns a low-priority HIP stream
# (the gradient stream): every dispatch of the eager two-stream step then takes ~50 us longer on the device.  A normal-priority
# gradient stream (sched.WgradStreamState.use_normal_priority, taken by dp.GradSync) or GPU_MAX_HW_QUEUES <= 3 restores the step.
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 NPVP_DP_FORCE=1 NPVP_DIST_BACKEND=nccl
Q="--gpus 1 --workload ${W:-c4} --steps 8 --warmup 3 --no-secondary --no-cpu-baseline --no-probe --mode eager --dp-graph never --dp-fused-trial never"
P=29560
run() { P=$((P+1)); echo "== $1"; shift; env MASTER_PORT=$P "$@" python3 bench.py $Q > /tmp/bisect_run.log 2>&1; grep -a "timed steps" /tmp/bisect_run.log | cut -c1-230; grep -aq "timed steps" /tmp/bisect_run.log || tail -5 /tmp/bisect_run.log; }
case "${1:-stages}" in
stages)    # NPVP_DP_STAGE: how far bench.py switches the data-parallel machinery on (init / convert / one / bcast / model)
  run "process group initialised, nothing else (low-priority gradient stream)" NPVP_DP_STAGE=init NPVP_WGRAD_PRIORITY=low
  run "+ SyncBatchNorm conversion" NPVP_DP_STAGE=convert NPVP_WGRAD_PRIORITY=low
  run "+ ONE small broadcast (the communicator exists)" NPVP_DP_STAGE=one NPVP_WGRAD_PRIORITY=low
  run "+ model broadcast" NPVP_DP_STAGE=model NPVP_WGRAD_PRIORITY=low
  run "the whole data-parallel step, low-priority gradient stream" NPVP_WGRAD_PRIORITY=low
  run "the whole data-parallel step (default: normal priority under data parallelism)" ;;
env)
  run "one small broadcast, low priority (baseline)" NPVP_DP_STAGE=one NPVP_WGRAD_PRIORITY=low
  run "one small broadcast, low priority, watchdog + monitoring off" NPVP_DP_STAGE=one NPVP_WGRAD_PRIORITY=low TORCH_NCCL_ASYNC_ERROR_HANDLING=0 TORCH_NCCL_ENABLE_MONITORING=0
  run "one small broadcast, low priority, HSA_ENABLE_INTERRUPT=0" NPVP_DP_STAGE=one NPVP_WGRAD_PRIORITY=low HSA_ENABLE_INTERRUPT=0
  run "one small broadcast, low priority, NCCL_MAX_NCHANNELS=1" NPVP_DP_STAGE=one NPVP_WGRAD_PRIORITY=low NCCL_MAX_NCHANNELS=1 NCCL_MIN_NCHANNELS=1
  run "one small broadcast, gradient stream at normal priority" NPVP_DP_STAGE=one NPVP_WGRAD_PRIORITY=normal
  run "one small broadcast, no gradient stream" NPVP_DP_STAGE=one NPVP_WGRAD_STREAM=0 ;;
queues)
  for N in 1 2 3 4 6 8 16; do run "one small broadcast, low priority, GPU_MAX_HW_QUEUES=$N" NPVP_DP_STAGE=one NPVP_WGRAD_PRIORITY=low GPU_MAX_HW_QUEUES=$N; done ;;
esac
