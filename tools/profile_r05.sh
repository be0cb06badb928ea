#!/bin/bash
# Round-5 profile set, run on the GPU box from the repo root:  bash tools/profile_r05.sh <step> [<step> ...]
# Writes under gpurun_out/r05/ (summaries copied to profiles/r05_* afterwards).  Counter passes carry no other tracing domain and
# the profiled program stands directly after `--`.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/r05"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
QUIET="--no-secondary --no-cpu-baseline --no-probe"

for STEP in "$@"; do
case "$STEP" in
bench)      # the default command, as the driver runs it
  python3 $ROOT/bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || exit 1
  tail -c 1500 "$OUT/bench_default.json" ;;
mfma)       # BASELINE's "MFMA util %": one --pmc pass per workload, the bench record beside it for the unprofiled step time
  for W in c2p c2; do
    python3 $ROOT/bench.py --workload $W --steps 6 --warmup 3 $QUIET > "$OUT/bench_$W.json" 2> "$OUT/bench_$W.err" || exit 1
    MS=$(python3 -c "import json,sys; print(json.load(open('$OUT/bench_$W.json'))['ms_per_step'])")
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/mu_$W" -o $W -- python3 $ROOT/bench.py --workload $W --steps 3 --warmup 2 $QUIET > /dev/null 2> "$OUT/mu_$W.err" || exit 1
    python3 $ROOT/tools/rocpd_mfma_util.py $(ls "$OUT"/mu_$W/*.db | head -1) --steps 3 --ms $MS --workload $W --out "$OUT/mfma_util_$W.md" --json "$OUT/mfma_util_$W.json" | head -12
    rm -rf "$OUT/mu_$W"
  done ;;
graphs)     # the 8-clip shards and c0: eager / single-stream graph / two-stream graph (and the runtime's graph knobs)
  for W in c4 c3 c0; do
    python3 $ROOT/bench.py --workload $W --steps 10 --warmup 3 $QUIET 2>&1 | grep -E "timed steps|metric" | cut -c1-400 > "$OUT/g_${W}_eager.txt"
    python3 $ROOT/bench.py --workload $W --steps 10 --warmup 3 $QUIET --graph 2>&1 | grep -E "timed steps|captured" | cut -c1-300 > "$OUT/g_${W}_graph1.txt"
    cat "$OUT/g_${W}_eager.txt" "$OUT/g_${W}_graph1.txt" | grep "timed steps"
  done
  W=c4
  python3 $ROOT/bench.py --workload $W --steps 10 --warmup 3 $QUIET --graph --graph-streams 2 2>&1 | grep -E "timed steps|captured" | cut -c1-300 > "$OUT/g_${W}_graph2.txt"
  DEBUG_HIP_FORCE_GRAPH_QUEUES=2 python3 $ROOT/bench.py --workload $W --steps 10 --warmup 3 $QUIET --graph --graph-streams 2 2>&1 | grep -E "timed steps|captured" | cut -c1-300 > "$OUT/g_${W}_graph2_q2.txt"
  DEBUG_HIP_FORCE_GRAPH_QUEUES=1 python3 $ROOT/bench.py --workload $W --steps 10 --warmup 3 $QUIET --graph --graph-streams 2 2>&1 | grep -E "timed steps|captured" | cut -c1-300 > "$OUT/g_${W}_graph2_q1.txt"
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python3 $ROOT/bench.py --workload $W --steps 10 --warmup 3 $QUIET --graph 2>&1 | grep -E "timed steps|captured" | cut -c1-300 > "$OUT/g_${W}_graph1_nocap.txt"
  grep -H "timed steps" "$OUT"/g_${W}_graph*.txt ;;
shard)      # kernel trace of the 8-clip shard (c4): eager and graph replay
  rocprofv3 --kernel-trace -d "$OUT/kt4" -o c4 -- python3 $ROOT/bench.py --steps 7 --warmup 3 --workload c4 $QUIET > "$OUT/bench_c4_profiled.json" 2> "$OUT/kt4.err"
  python3 $ROOT/tools/rocpd_stats.py $(ls "$OUT"/kt4/*.db | head -1) "$OUT/kernel_stats_c4shard.csv" > /dev/null
  python3 $ROOT/tools/rocpd_streams.py $(ls "$OUT"/kt4/*.db | head -1) 5 "$OUT/streams_c4shard.md" > /dev/null
  rm -rf "$OUT/kt4"
  rocprofv3 --kernel-trace -d "$OUT/kt4g" -o c4 -- python3 $ROOT/bench.py --steps 7 --warmup 3 --workload c4 $QUIET --graph > "$OUT/bench_c4_graph_profiled.json" 2> "$OUT/kt4g.err"
  python3 $ROOT/tools/rocpd_stats.py $(ls "$OUT"/kt4g/*.db | head -1) "$OUT/kernel_stats_c4shard_graph.csv" > /dev/null
  python3 $ROOT/tools/rocpd_streams.py $(ls "$OUT"/kt4g/*.db | head -1) 5 "$OUT/streams_c4shard_graph.md" > /dev/null
  rm -rf "$OUT/kt4g"
  head -12 "$OUT/streams_c4shard_graph.md" ;;
c2trace)    # kernel trace + stream view of the primary workload
  rocprofv3 --kernel-trace -d "$OUT/kt" -o c2 -- python3 $ROOT/bench.py --steps 3 --warmup 2 $QUIET > "$OUT/bench_c2_profiled.json" 2> "$OUT/kt.err"
  python3 $ROOT/tools/rocpd_stats.py $(ls "$OUT"/kt/*.db | head -1) "$OUT/kernel_stats_c2.csv" > /dev/null
  python3 $ROOT/tools/rocpd_stats.py $(ls "$OUT"/kt/*.db | head -1) "$OUT/kernel_stats_c2_by_grid.csv" --by-grid > /dev/null
  python3 $ROOT/tools/rocpd_streams.py $(ls "$OUT"/kt/*.db | head -1) 2 "$OUT/streams_c2.md" > /dev/null
  rm -rf "$OUT/kt"
  head -30 "$OUT/streams_c2.md" ;;
traffic)    # HBM-side traffic of the primary workload (separate FETCH_SIZE / WRITE_SIZE passes)
  BENCH="python3 $ROOT/bench.py --steps 3 --warmup 2 $QUIET"
  rocprofv3 --pmc FETCH_SIZE -d "$OUT/pf" -o c2 -- $BENCH > /dev/null 2> "$OUT/pf.err"
  rocprofv3 --pmc WRITE_SIZE -d "$OUT/pw" -o c2 -- $BENCH > /dev/null 2> "$OUT/pw.err"
  python3 $ROOT/tools/rocpd_traffic.py $(ls "$OUT"/pf/*.db | head -1) $(ls "$OUT"/pw/*.db | head -1) "$OUT/hbm_traffic_c2.md" "$OUT/hbm_traffic_c2.json" > /dev/null
  rm -rf "$OUT/pf" "$OUT/pw" ;;
gemm)       # GEMM micro-benchmarks
  python3 $ROOT/tools/gemm_bench.py --mode f16x3 --rows 114688 20480 8192 --check > "$OUT/gemm_bench_f16x3.txt" 2>/dev/null
  tail -8 "$OUT/gemm_bench_f16x3.txt" ;;
tiles)      # shard-sized GEMMs: the forward / dgrad tiles and the fused dgrad + weight-gradient launch
  python3 $ROOT/tools/gemm_bench.py --mode f16x3 --rows 8192 4096 2048 --iters 50 > "$OUT/gemm_bench_small.txt" 2>/dev/null
  grep -E "all shapes" "$OUT/gemm_bench_small.txt"
  python3 $ROOT/tools/linear_bwd_bench.py > "$OUT/linear_bwd_bench.txt" 2>&1
  cat "$OUT/linear_bwd_bench.txt" ;;
mid)        # MlpDWBN layer backward at the c2 decoder size and at an 8-clip shard's
  for F in 1792 160; do python3 $ROOT/tools/mlpdw_bench.py $F >> "$OUT/mlpdw_bench.txt" 2>/dev/null; done
  cat "$OUT/mlpdw_bench.txt" ;;
midpmc)     # SQ counters of the fused MlpDWBN middle's backward at the c2 decoder size (what is it bound by?)
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_BUSY_CYCLES \
    -d "$OUT/m1" -o mid -- python3 $ROOT/tools/mlpdw_bench.py 1792 > /dev/null 2> "$OUT/m1.err" || exit 1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC \
    -d "$OUT/m2" -o mid -- python3 $ROOT/tools/mlpdw_bench.py 1792 > /dev/null 2> "$OUT/m2.err" || echo "pass 2 failed"
  rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_WAIT_INST_ANY TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum \
    -d "$OUT/m3" -o mid -- python3 $ROOT/tools/mlpdw_bench.py 1792 > /dev/null 2> "$OUT/m3.err" || echo "pass 3 failed"
  python3 $ROOT/tools/rocpd_pmc.py $(ls "$OUT"/m1/*.db "$OUT"/m2/*.db "$OUT"/m3/*.db 2>/dev/null) --filter npvp::mlpdw_mid --out "$OUT/pmc_mid_table.md"
  rm -rf "$OUT/m1" "$OUT/m2" "$OUT/m3" ;;
posfuse)    # positional-fuse backward at the c2 decoder / encoder size and at an 8-clip shard's
  python3 $ROOT/tools/posfuse_bench.py 64 28 > "$OUT/posfuse_bench.txt" 2>/dev/null
  python3 $ROOT/tools/posfuse_bench.py 64 2 >> "$OUT/posfuse_bench.txt" 2>/dev/null
  python3 $ROOT/tools/posfuse_bench.py 8 16 >> "$OUT/posfuse_bench.txt" 2>/dev/null
  cat "$OUT/posfuse_bench.txt" ;;
ab_droppath) # DropPath masks inside the dgrad / weight-gradient GEMMs at every size (round 4 limited it to <= 32 768 rows)
  python3 $ROOT/tools/ab_bench.py c2 ops.DROP_PATH_IN_GEMM_ROWS 32768 1000000000 8 2>/dev/null | grep -v "^\[bench" > "$OUT/ab_droppath_c2.txt"
  cat "$OUT/ab_droppath_c2.txt" ;;
ab)         # in-process A/Bs of scheduling knobs (tools/ab_bench.py: A, B, A, B on one box)
  : > "$OUT/ab_knobs.txt"
  for SPEC in "c2 ops.FusedLinearBwd.enabled True False" "c1 ops.FusedLinearBwd.MAX_ROWS 16384 32768" "c2 sched.WgradStreamState.enabled True False" \
              "c1 sched.WgradStreamState.enabled True False" "c2 sched.ReduceQueueState.enabled True False" "c4 ops.FusedLinearBwd.enabled True False" \
              "c4 ops.ActSink.enabled True False" "c2 ops.ActSink.enabled True False"; do
    echo "== $SPEC" >> "$OUT/ab_knobs.txt"
    python3 $ROOT/tools/ab_bench.py $SPEC 8 2>/dev/null | grep "^{" >> "$OUT/ab_knobs.txt"
  done
  cat "$OUT/ab_knobs.txt" ;;
ablib)      # A/B of two BUILDS on one box: the tree against a copy of the package built from other sources under .ab/old (made by hand:
            # stash the change, build, cp -r bench.py npvp_amd configs .ab/old/, pop, build); new, old, new, old per workload
  for W in ${ABLIB_WORKLOADS:-c2 c4}; do
    for T in new old new old; do
      B=$ROOT/bench.py; [ $T = old ] && B=$ROOT/.ab/old/bench.py
      MS=$(python3 $B --workload $W --steps 12 --warmup 4 --mode ${ABLIB_MODE:-eager} --no-cpu-baseline --no-probe --no-secondary 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])") || exit 1
      echo "$W ${ABLIB_MODE:-eager} $T $MS" | tee -a "$OUT/ab_lib.txt"
    done
  done ;;
ab2)        # second sweep: the deferred reductions in the eager two-stream step of the small workloads (fused launches off)
  : > "$OUT/ab_knobs2.txt"
  for SPEC in "c4 sched.ReduceQueueState.enabled True False" "c3 sched.ReduceQueueState.enabled True False" "c0 sched.ReduceQueueState.enabled True False" \
              "c1 sched.ReduceQueueState.enabled True False" "c0 ops.ActSink.enabled True False"; do
    echo "== $SPEC (FusedLinearBwd off)" >> "$OUT/ab_knobs2.txt"
    python3 $ROOT/tools/ab_bench.py $SPEC 8 --set=ops.FusedLinearBwd.enabled=False 2>/dev/null | grep "^{" >> "$OUT/ab_knobs2.txt"
  done
  cat "$OUT/ab_knobs2.txt" ;;
*) echo "unknown step $STEP"; exit 2 ;;
esac
done
