#!/bin/bash
# r04as: leftover team slots go to the costliest shared rows (W + 1 teams): team-encoder tests, soak of the mid-size policies, call timings
set -o pipefail
mkdir -p gpurun_out/r04as
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "share or cost or mid_size or give_up or golden or reused" > gpurun_out/r04as/pytest_sel.log 2>&1
rc=$?; tail -4 gpurun_out/r04as/pytest_sel.log; [ $rc = 0 ] || exit $rc
SOAK_CASES=300 SOAK_SEED=31 timeout -k 10 300 python scripts/soak_midsize.py > gpurun_out/r04as/soak_midsize_300.log 2>&1; rc=$?; tail -2 gpurun_out/r04as/soak_midsize_300.log; [ $rc = 0 ] || exit $rc
python scripts/table_build_time.py 2>&1 | grep "blocks" > gpurun_out/r04as/call_timings.log; cat gpurun_out/r04as/call_timings.log
python scripts/share_all_probe.py 2>&1 | cut -c1-330 > gpurun_out/r04as/share_all_probe.log; cut -c1-200 gpurun_out/r04as/share_all_probe.log
