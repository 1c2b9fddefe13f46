#!/bin/bash
# Round 4, step g: beam stripes of 16 / 18 for 32 < B <= 54 (B = 50 of the reference's sweep without ten phantom beams):
# parity, then the B = 50 cells of the sweep grid on the new build and on the 60-beam build (shape 3) on one box.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r04g}
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sweep_grid or wide_beam or soak or plan_names or select or diagnostic" > $OUT/pytest_b50.log 2>&1; rc=$?
tail -5 $OUT/pytest_b50.log
[ $rc -ne 0 ] && { echo "parity failed rc=$rc"; exit $rc; }
timeout 500 python scripts/grid_bench.py --beams 50 --check 0 --shape default 2>&1 | grep -v amdgpu.ids | tee $OUT/grid_b50_54.log
timeout 500 python scripts/grid_bench.py --beams 50 --check 0 --shape 3 2>&1 | grep -v amdgpu.ids | tee $OUT/grid_b50_60.log
timeout 300 python scripts/grid_bench.py --beams 40 --check 0 --shape default --omegas 3,5 2>&1 | grep -v amdgpu.ids | tee $OUT/grid_b40_48.log
timeout 300 python scripts/grid_bench.py --beams 40 --check 0 --shape 3 --omegas 3,5 2>&1 | grep -v amdgpu.ids | tee $OUT/grid_b40_60.log
