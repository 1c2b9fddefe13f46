#!/bin/bash
# Round 4, step e: same-box A/B of the beam-split encoder with 4 / 8 waves per workgroup; parity of the 8-wave build first.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r04e}
mkdir -p $OUT
C=$PWD/relative-entropy-coding_amd/csrc
IREC_LIB_PATH=$C/variants/split_nw8.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split or golden or small" > $OUT/pytest_nw8.log 2>&1; rc=$?
tail -3 $OUT/pytest_nw8.log
[ $rc -ne 0 ] && { echo "parity of the 8-wave build failed rc=$rc"; exit $rc; }
{
for round in 1 2; do
for V in split_nw4 split_nw8; do
  for cfg in "LATENTS=1 BEAMS=20" "LATENTS=1 BEAMS=10 EPS1=1.0"; do
    echo "== $V: $cfg"
    env $cfg IREC_VARIANT=auto REPS=10 IREC_LIB_PATH=$C/variants/$V.so timeout 120 python scripts/run_variant.py 2>&1 | tail -3
  done
done; done
for V in split_nw4 split_nw8; do
  echo "== stamps $V: LATENTS=1 BEAMS=20"
  env LATENTS=1 BEAMS=20 IREC_VARIANT=auto REPS=2 IREC_STAMPS=1 IREC_LIB_PATH=$C/variants/$V.so timeout 120 python scripts/run_variant.py 2>&1 | tail -6
done
} 2>&1 | tee $OUT/ab_split_nw.log
