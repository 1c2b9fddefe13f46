#!/bin/bash
# One GPU-box session: parity tests, smoke, a short bench and a kernel-trace profile.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" 
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"
timeout 900 python bench.py --steps 3 --warmup 1 --latents ${LATENTS:-1024} 2>&1 | tail -3 | tee gpurun_out/bench.log
