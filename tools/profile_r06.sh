#!/bin/bash
# Round-6 profile set, run on the GPU box from the repo root:  bash tools/profile_r06.sh <step> [<step> ...]
# Writes under gpurun_out/r06/ (summaries copied to profiles/r06_* afterwards).  Counter passes carry no other tracing domain and
# the profiled program stands directly after `--` (one-rank RCCL jobs get their rendezvous from the environment: no launcher hop).
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/r06"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
QUIET="--no-secondary --no-cpu-baseline --no-probe"
DP1="env RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29551 NPVP_DP_FORCE=1 NPVP_DIST_BACKEND=nccl"

for STEP in "$@"; do
case "$STEP" in
bench)      # the default command, as the driver runs it
  python3 $ROOT/bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || exit 1
  tail -c 1500 "$OUT/bench_default.json" ;;
dptrace)    # the data-parallel c4 shard step on ONE RCCL rank: eager (two streams + side stream) and replayed as graph segments
  export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29551 NPVP_DP_FORCE=1 NPVP_DIST_BACKEND=nccl
  for M in never always; do
    rocprofv3 --kernel-trace -d "$OUT/kdp_$M" -o c4 -- python3 $ROOT/bench.py --steps 7 --warmup 3 --workload c4 $QUIET --dp-graph $M --dp-fused-trial never > "$OUT/bench_c4_dp_$M.json" 2> "$OUT/kdp_$M.err"
    python3 $ROOT/tools/rocpd_stats.py $(ls "$OUT"/kdp_$M/*.db | head -1) "$OUT/kernel_stats_c4_dp_$M.csv" > /dev/null
    python3 $ROOT/tools/rocpd_streams.py $(ls "$OUT"/kdp_$M/*.db | head -1) 5 "$OUT/streams_c4_dp_$M.md" > /dev/null
    rm -rf "$OUT/kdp_$M"
    grep -a "timed steps" "$OUT/kdp_$M.err" | cut -c1-200
    head -8 "$OUT/streams_c4_dp_$M.md"
  done
  unset RANK WORLD_SIZE LOCAL_RANK NPVP_DP_FORCE NPVP_DIST_BACKEND ;;
*) echo "unknown step $STEP" ;;
esac
done
