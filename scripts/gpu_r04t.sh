#!/bin/bash
# Round 4, step t: float32-edge soak on the final kernels; phase stamps of the mid-size calls AFTER this round's changes (shared rows).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r04t}
mkdir -p $OUT
C=$PWD/relative-entropy-coding_amd/csrc
timeout -k 10 400 python scripts/soak_extreme.py 400 2>&1 | grep -v amdgpu.ids | tee $OUT/soak_extreme.log | tail -3
{
for cfg in "LATENTS=38 BEAMS=20" "LATENTS=38 BEAMS=20 NO_SPLIT=1" "LATENTS=34 BEAMS=10 EPS1=1.0" "LATENTS=1 BEAMS=20"; do
  echo "== stamps: $cfg"
  env $cfg IREC_VARIANT=auto REPS=2 IREC_LIB_PATH=$C/variants/stamps.so IREC_STAMPS=1 timeout 120 python scripts/run_variant.py 2>&1 | grep -v amdgpu.ids | tail -20
done
} 2>&1 | tee $OUT/stamps_after.log
