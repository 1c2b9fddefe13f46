#!/bin/bash
# Diagnostic: block order of the persistent kernel (largest first = product) against natural / interleaved orders.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for rep in 1 2; do
for o in "" natural natural_tail spread; do
  echo "== order=${o:-largest_first}"; ORDER=$o LATENTS=${LATENTS:-8192} REPS=3 timeout 300 python scripts/run_variant.py 2>&1 | tail -1
done; done
} | tee gpurun_out/order.log
