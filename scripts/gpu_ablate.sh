#!/bin/bash
# Diagnostic: time the team encoder with one phase removed (results are garbage; timing only).  Needs csrc/variants/*.so.
# The variants are loaded by path (IREC_LIB_PATH); the product library is never touched.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
C=$PWD/relative-entropy-coding_amd/csrc
run() { timeout 300 python scripts/run_variant.py 2>&1 | tail -1; }
{
echo "== full"; LATENTS=2048 run
for v in SCORING SELECT UPDATE ${EXTRA_VARIANTS:-}; do
  [ -f $C/variants/ablate_$v.so ] || continue
  echo "== without $v"; IREC_LIB_PATH=$C/variants/ablate_$v.so LATENTS=2048 run
done
} | tee gpurun_out/ablate.log
