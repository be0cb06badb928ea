#!/bin/bash
# the data-parallel step on ONE RCCL rank (NPVP_DP_FORCE=1), eager and replayed as graph segments, per workload; run on the GPU box
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 NPVP_DP_FORCE=1 NPVP_DIST_BACKEND=nccl
P=29700
for W in "$@"; do
  for M in never always; do
    P=$((P+1))
    echo "== $W, --dp-graph $M"
    MASTER_PORT=$P python3 bench.py --gpus 1 --workload $W --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-probe --dp-graph $M --dp-fused-trial never > /tmp/dpm.log 2>&1
    grep -a "mode trial\|timed steps\|recorded" /tmp/dpm.log | cut -c1-260
    grep -aq "timed steps" /tmp/dpm.log || tail -8 /tmp/dpm.log
  done
done
