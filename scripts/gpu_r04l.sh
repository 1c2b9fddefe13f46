#!/bin/bash
# Round 4, step l: alpha_choice_kernel (the per-call proposal tables of the team encoder) compiled for 1 / 2 / 4 waves per SIMD:
# what a mid-size call pays for its tables when they are rebuilt (as issued), same box.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r04l}
mkdir -p $OUT
C=$PWD/relative-entropy-coding_amd/csrc
{
for round in 1 2; do
for V in libirec_hip variants/choice_wpe2 variants/choice_wpe4; do
  for cfg in "LATENTS=38 BEAMS=20" "LATENTS=34 BEAMS=10 EPS1=1.0" "LATENTS=1024 BEAMS=20"; do
    echo "== $V: $cfg"
    env $cfg IREC_VARIANT=auto REPS=10 IREC_LIB_PATH=$C/$V.so timeout 120 python scripts/run_variant.py 2>&1 | tail -2
  done
done; done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_choice_wpe.log
