#!/bin/bash
# Ablation ladder of the fp16 GEMM kernel (measurement builds: results INVALID, timing only).  Run on the GPU box.
set -e
cd "$(dirname "$0")/.."
for abl in 0 1 2 4 8 3 7 15; do
  NPVP_HIPCC_EXTRA="-DNPVP_H_ABL=$abl" python npvp_amd/build.py --force > /dev/null 2>&1
  echo "== NPVP_H_ABL=$abl"
  python tools/gemm_bench.py --mode f16x3 --rows 114688 --iters 20 2>/dev/null | grep "^R=" | awk '{print $2,$3,$4,$5,$6,$7,$8,$9,$10,$11,$12,$13}'
done
python npvp_amd/build.py --force > /dev/null 2>&1
