#!/bin/bash
# r04al: which of the selection changes costs the three-team build?  same-box A/B at 8192 and 256 latents per call
set -o pipefail
mkdir -p gpurun_out/r04al
R=$PWD
for v in main before_select noassume norank32 neither main before_select; do
  [ $v = main ] && unset IREC_LIB_PATH || export IREC_LIB_PATH=$R/relative-entropy-coding_amd/csrc/variants/$v.so
  echo "== $v" >> gpurun_out/r04al/ab.log
  LATENTS=8192 REPS=4 python scripts/run_variant.py 2>&1 | grep "latents/s" | tail -2 >> gpurun_out/r04al/ab.log
  LATENTS=256 REPS=12 python scripts/run_variant.py 2>&1 | grep "latents/s" | sort -t' ' -k5 -n | head -2 >> gpurun_out/r04al/ab.log
done
cat gpurun_out/r04al/ab.log
