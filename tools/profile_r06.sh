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
c2trace)    # kernel trace + stream view of the primary workload as the default command runs it (rocprofv3 --kernel-trace --stats)
  rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o c2 -- python3 $ROOT/bench.py --steps 3 --warmup 2 --mode eager $QUIET > "$OUT/bench_c2_profiled.json" 2> "$OUT/kt.err"
  python3 $ROOT/tools/rocpd_stats.py $(ls "$OUT"/kt/*.db | head -1) "$OUT/kernel_stats_c2.csv" > /dev/null
  python3 $ROOT/tools/rocpd_streams.py $(ls "$OUT"/kt/*.db | head -1) 2 "$OUT/streams_c2.md" > /dev/null
  rm -rf "$OUT/kt"
  head -12 "$OUT/kernel_stats_c2.csv" ;;
traffic)    # HBM-side traffic of the primary workload (separate FETCH_SIZE / WRITE_SIZE passes); the JSON records the library's sha256
  BENCH="python3 $ROOT/bench.py --steps 3 --warmup 2 --mode eager $QUIET"
  rocprofv3 --pmc FETCH_SIZE -d "$OUT/pf" -o c2 -- $BENCH > /dev/null 2> "$OUT/pf.err"
  rocprofv3 --pmc WRITE_SIZE -d "$OUT/pw" -o c2 -- $BENCH > /dev/null 2> "$OUT/pw.err"
  python3 $ROOT/tools/rocpd_traffic.py $(ls "$OUT"/pf/*.db | head -1) $(ls "$OUT"/pw/*.db | head -1) "$OUT/hbm_traffic_c2.md" "$OUT/hbm_traffic_c2.json" > /dev/null
  rm -rf "$OUT/pf" "$OUT/pw"
  tail -8 "$OUT/hbm_traffic_c2.md" ;;
mfma)       # BASELINE's "MFMA util %": one --pmc pass per workload, the bench record beside it for the unprofiled step time
  for W in c2p c2; do
    python3 $ROOT/bench.py --workload $W --steps 6 --warmup 3 --mode eager $QUIET > "$OUT/bench_$W.json" 2> "$OUT/bench_$W.err" || exit 1
    MS=$(python3 -c "import json,sys; print(json.load(open('$OUT/bench_$W.json'))['ms_per_step'])")
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/mu_$W" -o $W -- python3 $ROOT/bench.py --workload $W --steps 3 --warmup 2 --mode eager $QUIET > /dev/null 2> "$OUT/mu_$W.err" || exit 1
    python3 $ROOT/tools/rocpd_mfma_util.py $(ls "$OUT"/mu_$W/*.db | head -1) --steps 3 --ms $MS --workload $W --out "$OUT/mfma_util_$W.md" --json "$OUT/mfma_util_$W.json" | head -12
    rm -rf "$OUT/mu_$W"
  done ;;
gemmpmc)    # SQ counter passes over the stand-alone GEMMs at the c2 decoder shapes (two passes of 8 counters, no other tracing domain)
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE \
    -d "$OUT/g1" -o gemm -- python3 $ROOT/tools/gemm_pmc.py > /dev/null 2> "$OUT/g1.err"
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES \
    -d "$OUT/g2" -o gemm -- python3 $ROOT/tools/gemm_pmc.py > /dev/null 2> "$OUT/g2.err"
  python3 $ROOT/tools/rocpd_pmc.py $(ls "$OUT"/g1/*.db | head -1) $(ls "$OUT"/g2/*.db | head -1) --filter npvp::gemm --out "$OUT/pmc_gemm_table.md" > /dev/null
  rm -rf "$OUT/g1" "$OUT/g2"
  cat "$OUT/pmc_gemm_table.md" | cut -c1-400 ;;
*) echo "unknown step $STEP" ;;
esac
done
