#!/bin/bash
# r04w: cost-ordered hand-out of mid-size calls: probe timings + the tests of the team encoder's hand-out
set -o pipefail
mkdir -p gpurun_out/r04w
python scripts/fold_order_probe.py > gpurun_out/r04w/fold_probe.log 2>&1 || { tail -20 gpurun_out/r04w/fold_probe.log; exit 1; }
cat gpurun_out/r04w/fold_probe.log | cut -c1-200
SKEW=1 python scripts/fold_order_probe.py > gpurun_out/r04w/fold_probe_skew.log 2>&1 || { tail -20 gpurun_out/r04w/fold_probe_skew.log; exit 1; }
cat gpurun_out/r04w/fold_probe_skew.log | cut -c1-200
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cost_ordered or share_rows or give_up or reused or golden" > gpurun_out/r04w/pytest_sel.log 2>&1
rc=$?; tail -8 gpurun_out/r04w/pytest_sel.log; exit $rc
