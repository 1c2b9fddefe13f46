#!/bin/bash
# r04x: one preparation kernel per call (books + granules + row costs + tables), stamps committed by the encode kernel:
# full GPU suite, call timings, kernel traces
set -o pipefail
mkdir -p gpurun_out/r04x
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04x/pytest_gpu.log 2>&1
rc=$?; tail -5 gpurun_out/r04x/pytest_gpu.log; [ $rc = 0 ] || exit $rc
python scripts/table_build_time.py > gpurun_out/r04x/table_build.log 2>&1 || { tail -20 gpurun_out/r04x/table_build.log; exit 1; }
cat gpurun_out/r04x/table_build.log
