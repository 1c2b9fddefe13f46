#!/bin/bash
# Round 4, step o: half-slot look-up granules (ten look-ups per issue instead of twenty) in the two-team 20-beam build, same box.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r04o}
mkdir -p $OUT
C=$PWD/relative-entropy-coding_amd/csrc
IREC_LIB_PATH=$C/variants/nh2.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or one_to_two or share_rows or diagnostic" > $OUT/pytest_nh2.log 2>&1; rc=$?
tail -3 $OUT/pytest_nh2.log
[ $rc -ne 0 ] && { echo "parity failed rc=$rc"; exit $rc; }
{
for round in 1 2; do
for V in libirec_hip variants/nh2; do
  for cfg in "LATENTS=38 BEAMS=20" "LATENTS=38 BEAMS=20 NO_SPLIT=1" "LATENTS=50 BEAMS=20" "LATENTS=28 BEAMS=20 SHAPE=2" "LATENTS=512 BEAMS=20 SHAPE=2" "LATENTS=4096 BEAMS=20 SHAPE=2"; do
    echo "== $V: $cfg"
    env $cfg IREC_VARIANT=auto REPS=10 IREC_LIB_PATH=$C/$V.so timeout 120 python scripts/run_variant.py 2>&1 | tail -2
  done
done; done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_nh2.log
