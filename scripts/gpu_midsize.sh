#!/bin/bash
# Diagnostic: team-encoder workgroup shapes on calls of one to three blocks per CU (config 3's per-GPU share: 38 images = 342 blocks).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for L in ${SIZES:-28 38 57 76 114}; do
for sh in default 1x2 2x2 2; do
  echo "== latents=$L ($((L*9)) blocks) shape=$sh"; SHAPE=$sh LATENTS=$L REPS=6 timeout 120 python scripts/run_variant.py 2>&1 | tail -2
done; done
} | tee gpurun_out/midsize.log
