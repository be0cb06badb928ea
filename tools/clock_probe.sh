#!/bin/bash
# Is the fp16 GEMM power bound?  Effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and matrix-pipe busy fraction of the
# shipped kernel and of measurement builds with parts of the K loop removed (results invalid, timing only).  Run on the GPU box.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/clk"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for abl in 0 7 15; do
  NPVP_HIPCC_EXTRA="-DNPVP_H_ABL=$abl" python3 $ROOT/npvp_amd/build.py --force > /dev/null 2>&1
  rm -rf "$OUT/p$abl"
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES -d "$OUT/p$abl" -o g -- python3 $ROOT/tools/gemm_pmc.py > /dev/null 2> "$OUT/p$abl.err"
  echo "== NPVP_H_ABL=$abl"
  python3 $ROOT/tools/rocpd_pmc.py $(ls "$OUT"/p$abl/*.db | head -1) --filter "npvp::gemm_f16_kernel" | grep "gemm_f16_kernel" | cut -c1-400
  rm -rf "$OUT/p$abl"
done
python3 $ROOT/npvp_amd/build.py --force > /dev/null 2>&1
