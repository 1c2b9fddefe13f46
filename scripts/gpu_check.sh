#!/bin/bash
# One GPU-box session: parity tests, smoke, the default bench line, the config-3 harness.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench (default line, CPU baselines included)"
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/bench_default.err | tail -1 | tee gpurun_out/bench_default.log
tail -12 gpurun_out/bench_default.err
echo "== config-3 harness"
timeout 600 python scripts/config3_harness.py 2>&1 | tail -3 | tee gpurun_out/config3.log
