#!/bin/bash
# Round 4, step q: randomised parity soaks on the final kernels: the standard mix, candidate sets beyond 1024, and (new) blocks of
# more than 1024 dims with up to 256 beams.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r04q}
mkdir -p $OUT
SOAK_CASES=1500 SOAK_SEED=41 timeout -k 10 420 python scripts/soak_parity.py 2>&1 | grep -v amdgpu.ids | tee $OUT/soak_1500.log
SOAK_LARGE=1 SOAK_CASES=400 SOAK_SEED=42 timeout -k 10 420 python scripts/soak_parity.py 2>&1 | grep -v amdgpu.ids | tee $OUT/soak_large_400.log
SOAK_BIG=1 SOAK_CASES=250 SOAK_SEED=43 timeout -k 10 300 python scripts/soak_parity.py 2>&1 | grep -v amdgpu.ids | tee $OUT/soak_big_250.log
