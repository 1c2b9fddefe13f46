#!/bin/bash
# r04ar: randomised parity soaks on the final kernels (after the selection / split-encoder changes)
set -o pipefail
mkdir -p gpurun_out/r04ar
SOAK_CASES=600 SOAK_SEED=21 timeout -k 10 500 python scripts/soak_parity.py > gpurun_out/r04ar/soak_600.log 2>&1; r1=$?; tail -2 gpurun_out/r04ar/soak_600.log
SOAK_BIG=1 SOAK_CASES=120 SOAK_SEED=22 timeout -k 10 300 python scripts/soak_parity.py > gpurun_out/r04ar/soak_big_120.log 2>&1; r2=$?; tail -2 gpurun_out/r04ar/soak_big_120.log
SOAK_CASES=400 SOAK_SEED=23 timeout -k 10 300 python scripts/soak_midsize.py > gpurun_out/r04ar/soak_midsize_400.log 2>&1; r3=$?; tail -2 gpurun_out/r04ar/soak_midsize_400.log
timeout -k 10 200 python scripts/soak_extreme.py > gpurun_out/r04ar/soak_extreme.log 2>&1; r4=$?; tail -2 gpurun_out/r04ar/soak_extreme.log
exit $((r1 + r2 + r3 + r4))
