#!/bin/bash
# A/B of the product library against one diagnostic build on the same box in one session (alternating runs).
# usage: VARIANT="name [name ...]" [LATENTS=8192] scripts/gpu_ab.sh
# (build the diagnostic libraries first: make -C relative-entropy-coding_amd/csrc variant NAME=<name> DEFS=-DIREC_...=0)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
C=$PWD/relative-entropy-coding_amd/csrc
V=${VARIANT:?name of csrc/variants/<name>.so}
{
for rep in 1 2; do
  echo "== product"; LATENTS=${LATENTS:-8192} REPS=4 timeout 300 python scripts/run_variant.py 2>&1 | tail -2
  for v in $V; do echo "== $v"; IREC_LIB_PATH=$C/variants/$v.so LATENTS=${LATENTS:-8192} REPS=4 timeout 300 python scripts/run_variant.py 2>&1 | tail -2; done
done
} | tee gpurun_out/ab_$(echo $V | tr ' ' '_').log
