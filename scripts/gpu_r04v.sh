#!/bin/bash
# r04v: one-launch proposal-table build (LDS-atomic ranks): timings + the tests that look at the tables
set -o pipefail
mkdir -p gpurun_out/r04v
python scripts/table_build_time.py > gpurun_out/r04v/table_build.log 2>&1 || { tail -20 gpurun_out/r04v/table_build.log; exit 1; }
cat gpurun_out/r04v/table_build.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "proposal_table or reused or golden or shared or chunk or lut" > gpurun_out/r04v/pytest_sel.log 2>&1
rc=$?; tail -5 gpurun_out/r04v/pytest_sel.log; exit $rc
