#!/bin/bash
# PMC passes over scripts/run_variant.py (one encode call shape, set by LATENTS / BEAMS / EPS1 / NO_TEN ... in the environment).
#   usage: TAG=r06e KERNEL=encode_ten scripts/gpu_pmc_variant.sh        -> gpurun_out/pmc_$TAG/*.summary
set -u
export TMPDIR=/tmp
TAG=${TAG:-pmc}; OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  REPS=3 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$name -- python3 scripts/run_variant.py > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 scripts/pmc_summary.py "$f" ${KERNEL:-encode_} | tee $OUT/$name.summary
  rm -rf $OUT/$name
done
