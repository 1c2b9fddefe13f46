#!/bin/bash
# Round 4, step a: where a mid-size call's time goes (VERDICT r03 "Next" #1).  Timings on the product library, then the phase
# stamps of the diagnostic build (csrc/variants/stamps.so) for the same calls: 342 blocks (38 latents, B = 20, S = 36:
# one GPU's share of config 3), 306 blocks (34 latents, B = 10, S = 20: a Kodak image's first level), 9 blocks (split encoder).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r04a}
mkdir -p $OUT
C=$PWD/relative-entropy-coding_amd/csrc
run() { # name, env...
  local name=$1; shift
  echo "== $name: $*"
  env "$@" REPS=8 timeout 120 python scripts/run_variant.py 2>&1 | tail -4
}
{
run mid342_default   LATENTS=38 BEAMS=20 IREC_VARIANT=auto
run mid342_1x2       LATENTS=38 BEAMS=20 IREC_VARIANT=auto SHAPE=1x2
run mid342_3         LATENTS=38 BEAMS=20 IREC_VARIANT=auto SHAPE=3
run mid252_default   LATENTS=28 BEAMS=20 IREC_VARIANT=auto
run kodak306_default LATENTS=34 BEAMS=10 EPS1=1.0 IREC_VARIANT=auto
run kodak306_3       LATENTS=34 BEAMS=10 EPS1=1.0 IREC_VARIANT=auto SHAPE=3
run split9_b20       LATENTS=1 BEAMS=20 IREC_VARIANT=auto
run split9_b10       LATENTS=1 BEAMS=10 EPS1=1.0 IREC_VARIANT=auto
run split18_b20      LATENTS=2 BEAMS=20 IREC_VARIANT=auto
} 2>&1 | tee $OUT/timings.log
{
for cfg in "LATENTS=38 BEAMS=20" "LATENTS=38 BEAMS=20 SHAPE=1x2" "LATENTS=38 BEAMS=20 SHAPE=3" "LATENTS=34 BEAMS=10 EPS1=1.0" "LATENTS=1 BEAMS=20" "LATENTS=1 BEAMS=10 EPS1=1.0"; do
  echo "== stamps: $cfg"
  env $cfg IREC_VARIANT=auto REPS=2 IREC_LIB_PATH=$C/variants/stamps.so IREC_STAMPS=1 timeout 120 python scripts/run_variant.py 2>&1 | tail -22
done
} 2>&1 | tee $OUT/stamps.log
