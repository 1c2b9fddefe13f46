#!/bin/bash
# The ONE A/B script (round 5; it replaces the per-experiment scripts/gpu_r04*.sh of round 4): the product library against
# diagnostic builds on the same box in one session, alternating runs of scripts/run_variant.py.
#
#   usage: scripts/gpu_ab.sh NAME [-DIREC_...=v ...]     one variant: built from DEFS if csrc/variants/NAME.so is missing
#          VARIANTS="a b c" scripts/gpu_ab.sh            several prebuilt variants (make -C .../csrc variant NAME=a DEFS=...)
#   env:   KIND=team|k|lone|dec|gang|ten   which translation unit the DEFS go to (Makefile targets variant, variant_k, variant_lone, variant_dec, variant_gang, variant_ten)
#          LATENTS BEAMS OMEGA EPS1 MAXK SHAPE IREC_VARIANT ...   passed on to run_variant.py;  REPS (default 4), ROUNDS (default 2)
#          STAMPS=1               run NAME = stamps (make stamps) once with IREC_STAMPS=1 instead of an A/B: phase shares
#          TESTS="-k expr"        first run the GPU parity tests selected by expr against every variant (bit-exactness of an A/B build)
# Variants are loaded through scripts/with_lib.py (an explicit irec._lib.load(path)); the product library is never touched
# and the product loader reads no environment variable.  Log: gpurun_out/ab_<names>.log
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD; C=$R/relative-entropy-coding_amd/csrc
V=${VARIANTS:-${1:?variant name}}
[ $# -ge 1 ] && shift
if [ -z "${VARIANTS:-}" ] && [ ! -f $C/variants/$V.so ]; then
  case ${KIND:-team} in team) T=variant;; k) T=variant_k;; lone) T=variant_lone;; dec) T=variant_dec;; gang) T=variant_gang;; ten) T=variant_ten;; *) echo "unknown KIND=$KIND"; exit 1;; esac
  if [ "$V" = stamps ]; then make -s -C $C stamps; else make -s -C $C $T NAME=$V DEFS="$*"; fi || exit 1
fi
run() { timeout -k 10 300 python scripts/with_lib.py $1 scripts/run_variant.py 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-2}; }
{
if [ -n "${STAMPS:-}" ]; then echo "== $V (phase stamps)"; IREC_STAMPS=1 TAIL=24 REPS=${REPS:-2} run $V; exit 0; fi
if [ -n "${TESTS:-}" ]; then
  for v in $V; do echo "== parity of $v: pytest $TESTS"; timeout -k 10 900 python scripts/with_lib.py $v -m pytest tests/test_gpu_parity.py -m gpu -x -q $TESTS 2>&1 | tail -2; done
fi
for rep in $(seq ${ROUNDS:-2}); do
  echo "== product"; REPS=${REPS:-4} run main
  for v in $V; do echo "== $v"; REPS=${REPS:-4} run $v; done
done
} | tee gpurun_out/ab_$(echo $V | tr ' ' '_').log
