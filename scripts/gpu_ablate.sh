#!/bin/bash
# Diagnostic: time the team encoder with one phase removed (results are garbage; timing only).  Needs csrc/variants/*.so.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
C=relative-entropy-coding_amd/csrc
cp $C/libirec_hip.so /tmp/full.so
run() { timeout 300 python scripts/run_variant.py 2>&1 | tail -1; }
{
echo "== full"; LATENTS=2048 run
for v in SCORING SELECT UPDATE; do
  cp $C/variants/ablate_$v.so $C/libirec_hip.so
  echo "== without $v"; LATENTS=2048 run
done
cp /tmp/full.so $C/libirec_hip.so
} | tee gpurun_out/ablate.log
