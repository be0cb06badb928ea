#!/bin/bash
# Round-4 profile set, run on the GPU box from the repo root:  bash tools/profile_r04.sh
# Writes summaries under gpurun_out/prof_r04/ (copied to profiles/r04_* afterwards).  Counter passes are separate from the
# kernel-trace pass and from each other (FETCH_SIZE / WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md), and carry no other
# tracing domain.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_r04"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 2 --no-secondary --no-cpu-baseline --no-probe"

echo "[1] un-profiled bench (the record the kernel stats are read beside)"
python3 $ROOT/bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
tail -2 "$OUT/bench_c2.err"

echo "[2] kernel trace"
rocprofv3 --kernel-trace -d "$OUT/kt" -o c2 -- $BENCH > "$OUT/bench_c2_profiled.json" 2> "$OUT/kt.err"
python3 $ROOT/tools/rocpd_stats.py $(ls "$OUT"/kt/*.db | head -1) "$OUT/kernel_stats_c2.csv" > /dev/null
python3 $ROOT/tools/rocpd_stats.py $(ls "$OUT"/kt/*.db | head -1) "$OUT/kernel_stats_c2_by_grid.csv" --by-grid > /dev/null
python3 $ROOT/tools/rocpd_streams.py $(ls "$OUT"/kt/*.db | head -1) 2 "$OUT/streams_c2.md" > /dev/null
rm -rf "$OUT/kt"

echo "[3] FETCH_SIZE pass"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pf" -o c2 -- $BENCH > /dev/null 2> "$OUT/pf.err"
echo "[4] WRITE_SIZE pass"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pw" -o c2 -- $BENCH > /dev/null 2> "$OUT/pw.err"
python3 $ROOT/tools/rocpd_traffic.py $(ls "$OUT"/pf/*.db | head -1) $(ls "$OUT"/pw/*.db | head -1) "$OUT/hbm_traffic_c2.md" "$OUT/hbm_traffic_c2.json" > /dev/null
rm -rf "$OUT/pf" "$OUT/pw"

echo "[5] SQ counters of the fp16 GEMM kernels (two passes)"
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  -d "$OUT/g1" -o gemm -- python3 $ROOT/tools/gemm_pmc.py > /dev/null 2> "$OUT/g1.err"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES \
  -d "$OUT/g2" -o gemm -- python3 $ROOT/tools/gemm_pmc.py > /dev/null 2> "$OUT/g2.err"
python3 $ROOT/tools/rocpd_pmc.py $(ls "$OUT"/g1/*.db | head -1) $(ls "$OUT"/g2/*.db | head -1) --filter npvp::gemm --out "$OUT/pmc_gemm_table.md" > /dev/null
rm -rf "$OUT/g1" "$OUT/g2"
echo "[6] kernel trace of the 8-clip shard (c4)"
rocprofv3 --kernel-trace -d "$OUT/kt4" -o c4 -- python3 $ROOT/bench.py --steps 7 --warmup 3 --workload c4 --no-secondary --no-cpu-baseline --no-probe > "$OUT/bench_c4_profiled.json" 2> "$OUT/kt4.err"
python3 $ROOT/tools/rocpd_stats.py $(ls "$OUT"/kt4/*.db | head -1) "$OUT/kernel_stats_c4shard.csv" > /dev/null
python3 $ROOT/tools/rocpd_streams.py $(ls "$OUT"/kt4/*.db | head -1) 5 "$OUT/streams_c4shard.md" > /dev/null
rm -rf "$OUT/kt4"
echo "[7] fp16 range audit after 200 optimiser steps"
python3 $ROOT/tools/f16_range_audit.py --clips 8 --steps-before 200 > "$OUT/f16_range_audit.txt" 2> "$OUT/f16_range_audit.err"
tail -3 "$OUT/f16_range_audit.txt"
echo "[8] GEMM and attention micro-benchmarks"
python3 $ROOT/tools/gemm_bench.py --mode f16x3 --rows 114688 20480 8192 --check > "$OUT/gemm_bench_f16x3.txt" 2>/dev/null
python3 $ROOT/tools/gemm_bench.py --mode f16x3 --rows 16384 --check --heavy --grad-scale 1e-8 > "$OUT/gemm_bench_f16x3_heavy_tail_grads.txt" 2>/dev/null
python3 $ROOT/tools/attn_bench.py > "$OUT/attn_bench.txt" 2>/dev/null
ls -la "$OUT"
