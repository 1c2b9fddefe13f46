#!/bin/bash
# r04aa: split encoder (beam mode): proposal rows fetched one sub-batch ahead -- parity of the small-call tests, call timings
set -o pipefail
mkdir -p gpurun_out/r04aa
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split or small or golden or give_up or graph" > gpurun_out/r04aa/pytest_split.log 2>&1
rc=$?; tail -4 gpurun_out/r04aa/pytest_split.log; [ $rc = 0 ] || exit $rc
python scripts/table_build_time.py > gpurun_out/r04aa/call_timings.log 2>&1 || { tail -20 gpurun_out/r04aa/call_timings.log; exit 1; }
grep "blocks" gpurun_out/r04aa/call_timings.log
