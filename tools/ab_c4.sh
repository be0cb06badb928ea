#!/bin/bash
# shard benches without the per-GEMM event probe (run on the GPU box)
cd "$(dirname "$0")/.."
for w in c4 c3 c1 c0; do
  python bench.py --steps 10 --warmup 3 --workload $w --no-secondary --no-cpu-baseline --no-probe 2>&1 >/dev/null | grep "timed steps"
done
python bench.py --steps 6 --warmup 3 --workload c2 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "timed steps"
