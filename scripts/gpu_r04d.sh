#!/bin/bash
# Round 4, step d: same-box A/B of the split encoder -- next step's constants in the update (late) / between key publish and sweep (early).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r04d}
mkdir -p $OUT
C=$PWD/relative-entropy-coding_amd/csrc
{
for round in 1 2; do
for V in consts_late consts_early; do
  for cfg in "LATENTS=1 BEAMS=20" "LATENTS=2 BEAMS=20" "LATENTS=4 BEAMS=20" "LATENTS=1 BEAMS=10 EPS1=1.0"; do
    echo "== $V: $cfg"
    env $cfg IREC_VARIANT=auto REPS=10 IREC_LIB_PATH=$C/variants/$V.so timeout 120 python scripts/run_variant.py 2>&1 | tail -3
  done
done; done
for V in consts_late consts_early; do
  for cfg in "LATENTS=1 BEAMS=20" "LATENTS=2 BEAMS=20"; do
    echo "== stamps $V: $cfg"
    env $cfg IREC_VARIANT=auto REPS=2 IREC_STAMPS=1 IREC_LIB_PATH=$C/variants/$V.so timeout 120 python scripts/run_variant.py 2>&1 | tail -6
  done
done
} 2>&1 | tee $OUT/ab_split_consts.log
