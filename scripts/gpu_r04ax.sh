#!/bin/bash
# r04ax: phase stamps of the mid-size and small calls on the FINAL kernels (before: r04a, r04t)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r04ax
mkdir -p $OUT
C=$PWD/relative-entropy-coding_amd/csrc
{
for cfg in "LATENTS=38 BEAMS=20 MAXK=32" "LATENTS=14 BEAMS=20 MAXK=32" "LATENTS=28 BEAMS=20 MAXK=32" "LATENTS=34 BEAMS=10 EPS1=1.0 MAXK=32"; do
  echo "== stamps: $cfg"
  env $cfg IREC_VARIANT=auto REPS=2 IREC_LIB_PATH=$C/variants/stamps.so IREC_STAMPS=1 timeout 120 python scripts/run_variant.py 2>&1 | grep -v amdgpu.ids | tail -20
done
} 2>&1 | tee $OUT/stamps_final.log | cut -c1-150 | tail -70
