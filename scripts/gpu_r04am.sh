#!/bin/bash
# r04am: the quick selection only where it pays (one-table / split encoders, two-team builds): full suite, same-box A/B against the tree before the selection work
set -o pipefail
mkdir -p gpurun_out/r04am
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04am/pytest_gpu.log 2>&1
rc=$?; tail -4 gpurun_out/r04am/pytest_gpu.log; [ $rc = 0 ] || exit $rc
R=$PWD
for v in main before_select main before_select; do
  [ $v = main ] && unset IREC_LIB_PATH || export IREC_LIB_PATH=$R/relative-entropy-coding_amd/csrc/variants/$v.so
  echo "== $v" >> gpurun_out/r04am/ab_select.log
  LATENTS=8192 REPS=4 python scripts/run_variant.py 2>&1 | grep "latents/s" | tail -2 >> gpurun_out/r04am/ab_select.log
  LATENTS=256 REPS=12 python scripts/run_variant.py 2>&1 | grep "latents/s" | sort -t' ' -k5 -n | head -2 >> gpurun_out/r04am/ab_select.log
  python scripts/table_build_time.py 2>&1 | grep "blocks" | grep "tables kept" >> gpurun_out/r04am/ab_select.log
done
cat gpurun_out/r04am/ab_select.log
