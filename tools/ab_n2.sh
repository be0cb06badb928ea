#!/bin/bash
cd "$(dirname "$0")/.."
for v in 1 0 1 0; do
  NPVP_MID_BWD_N2=$v python bench.py --steps 6 --warmup 3 --workload c2 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "timed steps" | sed "s/^/n2=$v /"
done
