#!/bin/bash
# r04ao: split encoder with 8 waves per workgroup (IREC_SPLIT_NW=8) on the current kernels, same-box A/B on the 9- and 13-block calls
set -o pipefail
mkdir -p gpurun_out/r04ao
R=$PWD
for v in main nw8 main nw8; do
  [ $v = main ] && unset IREC_LIB_PATH || export IREC_LIB_PATH=$R/relative-entropy-coding_amd/csrc/variants/$v.so
  echo "== $v" >> gpurun_out/r04ao/ab_nw8.log
  python scripts/table_build_time.py 2>&1 | grep "^9 blocks\|^13 blocks" >> gpurun_out/r04ao/ab_nw8.log
done
cat gpurun_out/r04ao/ab_nw8.log
