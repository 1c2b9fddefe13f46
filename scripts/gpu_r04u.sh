#!/bin/bash
# Round 4, step u: every row shared two ways on the TWO-team build (calls of up to one block per CU), against the 8-wave striped team.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r04u}
mkdir -p $OUT
run() { local name=$1; shift; echo "== $name: $*"; env "$@" REPS=8 timeout 120 python scripts/run_variant.py 2>&1 | tail -3; }
{
for L in 8 14 20 28; do
run L${L}_default       LATENTS=$L BEAMS=20 IREC_VARIANT=auto
run L${L}_shareall_2t   LATENTS=$L BEAMS=20 IREC_VARIANT=auto SHAPE=2 SHARE_ALL=1
run L${L}_shareall_2t_w2 LATENTS=$L BEAMS=20 IREC_VARIANT=auto SHAPE=2 SHARE_ALL=1 SPLIT_W=2
done
run L28_b10_default     LATENTS=28 BEAMS=10 EPS1=1.0 IREC_VARIANT=auto
run L28_b10_shareall_2t LATENTS=28 BEAMS=10 EPS1=1.0 IREC_VARIANT=auto SHAPE=2 SHARE_ALL=1
} 2>&1 | grep -v amdgpu.ids | tee $OUT/share_all_two_teams.log
