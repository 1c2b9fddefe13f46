#!/bin/bash
# r04ad: four 4-wave teams per CU for 10 beams (128 VGPRs, diagnostic shape "4") against the default, the sweep's B = 10 / B = 1 cells
set -o pipefail
mkdir -p gpurun_out/r04ad
for sh in default 4 default 4; do
  echo "== shape $sh" >> gpurun_out/r04ad/ab_four_teams.log
  python scripts/grid_bench.py --omegas 2,3,4 --eps 1.0,1.2,1.5 --beams 10 --check 1 --reps 5 --shape $sh 2>&1 | grep "encode_" | cut -c1-118 >> gpurun_out/r04ad/ab_four_teams.log
done
cat gpurun_out/r04ad/ab_four_teams.log
