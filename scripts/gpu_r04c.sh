#!/bin/bash
# Round 4, step c: GPU suite on the r04 sources (decoder index checks, LUT injection, skewed K, dlog load out of the selection,
# step constants under the split encoder's wait), the mid-size / small-call timings again, and the hand-out under skewed K.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r04c}
mkdir -p $OUT
C=$PWD/relative-entropy-coding_amd/csrc
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -5 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && { echo "pytest failed rc=$rc"; exit $rc; }
run() { local name=$1; shift; echo "== $name: $*"; env "$@" REPS=8 timeout 120 python scripts/run_variant.py 2>&1 | tail -3; }
{
run mid342_default   LATENTS=38 BEAMS=20 IREC_VARIANT=auto
run mid252_default   LATENTS=28 BEAMS=20 IREC_VARIANT=auto
run kodak306_default LATENTS=34 BEAMS=10 EPS1=1.0 IREC_VARIANT=auto
run split9_b20       LATENTS=1 BEAMS=20 IREC_VARIANT=auto
run split9_b10       LATENTS=1 BEAMS=10 EPS1=1.0 IREC_VARIANT=auto
run split18_b20      LATENTS=2 BEAMS=20 IREC_VARIANT=auto
run batch8192        LATENTS=8192 BEAMS=20 IREC_VARIANT=auto
} 2>&1 | tee $OUT/timings.log
{
run skew8192_listed  LATENTS=8192 BEAMS=20 IREC_VARIANT=auto SKEW=1 MAXK=128 TABLE_STEPS=128
run skew8192_byK     LATENTS=8192 BEAMS=20 IREC_VARIANT=auto SKEW=1 MAXK=128 TABLE_STEPS=128 ORDER_BY_K=1
for cfg in "LATENTS=8192 BEAMS=20" "LATENTS=8192 BEAMS=20 SKEW=1 MAXK=128 TABLE_STEPS=128" "LATENTS=8192 BEAMS=20 SKEW=1 MAXK=128 TABLE_STEPS=128 ORDER_BY_K=1"; do
  echo "== stamps: $cfg"
  env $cfg IREC_VARIANT=auto REPS=2 IREC_LIB_PATH=$C/variants/stamps.so IREC_STAMPS=1 timeout 300 python scripts/run_variant.py 2>&1 | tail -19
done
} 2>&1 | tee $OUT/skew.log
