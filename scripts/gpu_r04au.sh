#!/bin/bash
# r04au: 16-byte loads in the one-table / split encoder's LDS table fill, same-box A/B on the small calls + small-call tests
set -o pipefail
mkdir -p gpurun_out/r04au
R=$PWD
for v in main before_fill main before_fill; do
  [ $v = main ] && unset IREC_LIB_PATH || export IREC_LIB_PATH=$R/relative-entropy-coding_amd/csrc/variants/$v.so
  echo "== $v" >> gpurun_out/r04au/ab_fill.log
  python scripts/table_build_time.py 2>&1 | grep "^9 blocks\|^13 blocks" >> gpurun_out/r04au/ab_fill.log
done
unset IREC_LIB_PATH
cat gpurun_out/r04au/ab_fill.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split or small or golden or one_table or variant or graph" > gpurun_out/r04au/pytest_sel.log 2>&1
rc=$?; tail -3 gpurun_out/r04au/pytest_sel.log; exit $rc
