#!/bin/bash
# r04ak: same-box A/B of the selection changes (variants/before_select.so = the tree at 9a1e6ac): headline kernel at 8192 latents, small and mid-size calls
set -o pipefail
mkdir -p gpurun_out/r04ak
R=$PWD
for v in main before_select main before_select; do
  [ $v = main ] && unset IREC_LIB_PATH || export IREC_LIB_PATH=$R/relative-entropy-coding_amd/csrc/variants/$v.so
  echo "== $v" >> gpurun_out/r04ak/ab_select.log
  LATENTS=8192 REPS=5 python scripts/run_variant.py 2>&1 | grep "latents/s" | tail -3 >> gpurun_out/r04ak/ab_select.log
  python scripts/table_build_time.py 2>&1 | grep "blocks" | grep "tables kept" >> gpurun_out/r04ak/ab_select.log
done
cat gpurun_out/r04ak/ab_select.log
