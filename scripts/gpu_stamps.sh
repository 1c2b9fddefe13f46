#!/bin/bash
# Diagnostic: per-wave phase shares of the team encoder (csrc/variants/stamps.so = -DIREC_TEAM_STAMPS build; not a timing run).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
C=relative-entropy-coding_amd/csrc
cp $C/libirec_hip.so /tmp/full.so
cp $C/variants/stamps.so $C/libirec_hip.so
IREC_STAMPS=1 LATENTS=${LATENTS:-2048} timeout 300 python scripts/run_variant.py 2>&1 | tail -16 | tee gpurun_out/stamps.log
cp /tmp/full.so $C/libirec_hip.so
