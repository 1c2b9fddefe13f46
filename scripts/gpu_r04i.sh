#!/bin/bash
# Round 4, step i: rows shared between teams (calls of one to two blocks per CU): parity, then timings against IREC_FLAG_NO_SPLIT.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r04i}
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "share or shared or mid_size or one_to_two or hand_out or golden or plan_names or two_streams" > $OUT/pytest_share.log 2>&1; rc=$?
tail -12 $OUT/pytest_share.log
[ $rc -ne 0 ] && { echo "parity failed rc=$rc"; exit $rc; }
run() { local name=$1; shift; echo "== $name: $*"; env "$@" REPS=8 timeout 120 python scripts/run_variant.py 2>&1 | tail -3; }
{
run mid342_share2     LATENTS=38 BEAMS=20 IREC_VARIANT=auto
run mid342_whole      LATENTS=38 BEAMS=20 IREC_VARIANT=auto NO_SPLIT=1
run mid342_share5_3t  LATENTS=38 BEAMS=20 IREC_VARIANT=auto SHAPE=3
run mid342_whole_3t   LATENTS=38 BEAMS=20 IREC_VARIANT=auto SHAPE=3 NO_SPLIT=1
run kodak306_share    LATENTS=34 BEAMS=10 EPS1=1.0 IREC_VARIANT=auto
run kodak306_whole    LATENTS=34 BEAMS=10 EPS1=1.0 IREC_VARIANT=auto NO_SPLIT=1
run kodak306_share_3t LATENTS=34 BEAMS=10 EPS1=1.0 IREC_VARIANT=auto SHAPE=3
run mid297_share      LATENTS=33 BEAMS=20 IREC_VARIANT=auto
run mid297_whole      LATENTS=33 BEAMS=20 IREC_VARIANT=auto NO_SPLIT=1
run mid450_share_3t   LATENTS=50 BEAMS=20 IREC_VARIANT=auto SHAPE=3
run mid450_whole      LATENTS=50 BEAMS=20 IREC_VARIANT=auto NO_SPLIT=1
run mid450_default    LATENTS=50 BEAMS=20 IREC_VARIANT=auto
} 2>&1 | tee $OUT/share_timings.log
