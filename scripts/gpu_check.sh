#!/bin/bash
# One GPU-box session: parity tests, soak, smoke, the default bench line, the config-3 harness.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -q -m gpu --durations=5 2>&1 | grep -v amdgpu.ids | tail -14 | tee gpurun_out/pytest_gpu.log
echo "== soak"; SOAK_CASES=${SOAK_CASES:-400} timeout 1500 python scripts/soak_parity.py 2>&1 | tail -1 | tee gpurun_out/soak.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/smoke.log
echo "== bench (default line, CPU baselines included)"
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/bench_default.err | tail -1 | tee gpurun_out/bench_default.log
grep -v amdgpu.ids gpurun_out/bench_default.err | tail -6
echo "== config-3 harness (300 images, then one GPU's share of 38)"
timeout 600 python scripts/config3_harness.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/config3.log
timeout 600 python scripts/config3_harness.py --images 38 --no-graph 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/config3_38.log
