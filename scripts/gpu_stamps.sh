#!/bin/bash
# Diagnostic: per-phase cycle shares of the fast encoders (in-kernel s_memtime stamps; not a timing run).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in table fused; do
  echo "== $v"
  IREC_STAMPS=1 IREC_VARIANT=$v timeout 300 python scripts/run_variant.py 2>&1 | grep -E "stamps|ms" | tail -6
done | tee gpurun_out/stamps.log
