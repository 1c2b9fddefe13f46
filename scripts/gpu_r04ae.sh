#!/bin/bash
# r04ae: are the sweep's small-S cells bound by the beam slab's traffic?  IREC_ABLATE_SLAB build (no beam reads / writes: wrong outputs, timing only)
set -o pipefail
mkdir -p gpurun_out/r04ae
R=$PWD
for v in main noslab; do
  [ $v = main ] && unset IREC_LIB_PATH || export IREC_LIB_PATH=$R/relative-entropy-coding_amd/csrc/variants/$v.so
  echo "== $v" >> gpurun_out/r04ae/ablate_slab.log
  python scripts/grid_bench.py --omegas 2,3 --eps 1.0,1.5 --beams 10,50 --check 0 --reps 5 2>&1 | grep "encode_" | cut -c1-110 >> gpurun_out/r04ae/ablate_slab.log
done
cat gpurun_out/r04ae/ablate_slab.log
