#!/bin/bash
# Round 4, step j: every row of a mid-size call shared between teams (default) against whole rows (NO_SPLIT) and excess-only sharing.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r04j}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "share_rows or shared_rows or golden or hand_out" > $OUT/pytest_share.log 2>&1; rc=$?
tail -6 $OUT/pytest_share.log
run() { local name=$1; shift; echo "== $name: $*"; env "$@" REPS=8 timeout 120 python scripts/run_variant.py 2>&1 | tail -3; }
{
for L in 8 14 28 38 42; do
run b20_L${L}_share   LATENTS=$L BEAMS=20 IREC_VARIANT=auto
run b20_L${L}_whole   LATENTS=$L BEAMS=20 IREC_VARIANT=auto NO_SPLIT=1
done
run b20_L38_excess    LATENTS=38 BEAMS=20 IREC_VARIANT=auto SHARE_EXCESS=1
for L in 14 34; do
run b10_L${L}_share   LATENTS=$L BEAMS=10 EPS1=1.0 IREC_VARIANT=auto
run b10_L${L}_whole   LATENTS=$L BEAMS=10 EPS1=1.0 IREC_VARIANT=auto NO_SPLIT=1
done
run b20_L38_W3        LATENTS=38 BEAMS=20 IREC_VARIANT=auto SPLIT_W=2
run b20_L28_W2        LATENTS=28 BEAMS=20 IREC_VARIANT=auto SPLIT_W=2
run b20_L14_W3        LATENTS=14 BEAMS=20 IREC_VARIANT=auto SPLIT_W=3
} 2>&1 | tee $OUT/share_all_timings.log
