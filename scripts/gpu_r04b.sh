#!/bin/bash
# Round 4, step b: what a pair-mode call could reach -- 684 blocks of TEN beams (S = 36) on the three-team build do the scoring work of
# the 684 half-blocks of a 342-block, 20-beam call.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r04b}
mkdir -p $OUT
run() { local name=$1; shift; echo "== $name: $*"; env "$@" REPS=8 timeout 120 python scripts/run_variant.py 2>&1 | tail -3; }
{
run b10_684_auto LATENTS=76 BEAMS=10 EPS1=1.2 IREC_VARIANT=auto
run b10_684_3    LATENTS=76 BEAMS=10 EPS1=1.2 IREC_VARIANT=auto SHAPE=3
run b10_504_3    LATENTS=56 BEAMS=10 EPS1=1.2 IREC_VARIANT=auto SHAPE=3
run b10_342_auto LATENTS=38 BEAMS=10 EPS1=1.2 IREC_VARIANT=auto
run b20_252_auto LATENTS=28 BEAMS=20 IREC_VARIANT=auto
} 2>&1 | tee $OUT/pair_estimate.log
