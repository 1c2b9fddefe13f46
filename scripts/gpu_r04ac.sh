#!/bin/bash
# r04ac: wave roles rotated per team (IREC_TEAM_ROTATE): same-box A/B against rot0 -- headline kernel at 8192 latents, sweep cells with
# few samples, mid-size calls; then the team-encoder parity tests
set -o pipefail
mkdir -p gpurun_out/r04ac
R=$PWD
for v in main rot0 main rot0; do
  [ $v = main ] && unset IREC_LIB_PATH || export IREC_LIB_PATH=$R/relative-entropy-coding_amd/csrc/variants/$v.so
  echo "== $v" >> gpurun_out/r04ac/ab.log
  LATENTS=8192 REPS=4 python scripts/run_variant.py 2>&1 | grep "latents/s" | tail -2 >> gpurun_out/r04ac/ab.log
  python scripts/grid_bench.py --omegas 2,3 --eps 1.0,1.5 --beams 10,50 --check 0 --reps 5 2>&1 | grep "encode_" | cut -c1-110 >> gpurun_out/r04ac/ab.log
done
unset IREC_LIB_PATH
cat gpurun_out/r04ac/ab.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "team or golden or share or cost or mid_size or shape" > gpurun_out/r04ac/pytest_team.log 2>&1
rc=$?; tail -3 gpurun_out/r04ac/pytest_team.log; exit $rc
