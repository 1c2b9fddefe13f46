#!/bin/bash
# Diagnostic: per-wave phase shares of the team encoder (csrc/variants/stamps.so = -DIREC_TEAM_STAMPS build; not a timing run).
# Loaded by path (IREC_LIB_PATH); the product library is never touched.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
C=$PWD/relative-entropy-coding_amd/csrc
IREC_LIB_PATH=$C/variants/stamps.so IREC_STAMPS=1 LATENTS=${LATENTS:-2048} timeout 300 python scripts/run_variant.py 2>&1 | tail -16 | tee gpurun_out/stamps.log
