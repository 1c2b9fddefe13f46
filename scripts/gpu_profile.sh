#!/bin/bash
# Kernel-trace + PMC profile of bench.py on the GPU box.  Summaries land in gpurun_out/prof_*; copy into profiles/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
L=${LATENTS:-8192}
TAG=${TAG:-r02}
echo "== quick parity"; timeout 600 python -m pytest tests -x -q -m gpu -k "golden or full_size or reduce_scatter" 2>&1 | tail -3
echo "== bench"; timeout 600 python bench.py --steps 5 --warmup 2 --latents $L --no-cpu-baseline 2>&1 | tail -4 | tee gpurun_out/bench_$TAG.log
echo "== kernel trace"
rm -rf gpurun_out/prof_$TAG; mkdir -p gpurun_out/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG/trace -- python bench.py --steps 5 --warmup 2 --latents $L --no-cpu-baseline > gpurun_out/prof_$TAG/trace.log 2>&1
f=$(find gpurun_out/prof_$TAG/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python - "$f" gpurun_out/prof_$TAG/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
out = [rows[0]] + [[r[0][:100]] + r[1:] for r in rows[1:]]
csv.writer(open(sys.argv[2], "w")).writerows(out)
for r in out[:6]: print(r)
PY
if [ "${PMC:-1}" = "1" ]; then
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE GRBM_COUNT"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/prof_$TAG/pmc_$name -- python bench.py --steps 2 --warmup 1 --latents $L --no-cpu-baseline > gpurun_out/prof_$TAG/pmc_$name.log 2>&1
  f=$(find gpurun_out/prof_$TAG/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scripts/pmc_summary.py "$f" ${KERNEL:-encode_team} | tee gpurun_out/prof_$TAG/pmc_$name.summary
done
fi
# keep only the small summaries
find gpurun_out/prof_$TAG -name "*.csv" -size +2M -delete
du -sh gpurun_out/prof_$TAG
python scripts/make_traffic_json.py gpurun_out/prof_$TAG $L ${KERNEL:-encode_team} > gpurun_out/prof_$TAG/traffic.json 2>&1 || true
