#!/bin/bash
# per-kernel event-pair times of the c4 shard step with and without an RCCL communicator alive in the process
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 NPVP_DP_FORCE=1 NPVP_DIST_BACKEND=nccl
Q="--gpus 1 --workload c4 --steps 8 --warmup 3 --no-secondary --no-cpu-baseline --probe-all --dp-graph never --dp-fused-trial never"
for S in convert one; do
  echo "== stage $S"
  MASTER_PORT=$((29600 + RANDOM % 300)) NPVP_DP_STAGE=$S python3 bench.py $Q 2>&1 | grep -a "timed steps\|probe " | cut -c1-220
done
