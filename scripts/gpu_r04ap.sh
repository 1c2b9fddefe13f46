#!/bin/bash
# r04ap: faster top-B selection (threshold by count probing, rank by constant-lane broadcasts + one 64-bit compare): full GPU suite, call timings
set -o pipefail
mkdir -p gpurun_out/r04ap
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04ap/pytest_gpu.log 2>&1
rc=$?; tail -4 gpurun_out/r04ap/pytest_gpu.log; [ $rc = 0 ] || exit $rc
python scripts/table_build_time.py > gpurun_out/r04ap/call_timings.log 2>&1 || { tail -20 gpurun_out/r04ap/call_timings.log; exit 1; }
grep "blocks" gpurun_out/r04ap/call_timings.log
LATENTS=8192 REPS=4 python scripts/run_variant.py 2>&1 | grep "latents/s" | tail -2 | tee gpurun_out/r04ap/headline_8192.log
