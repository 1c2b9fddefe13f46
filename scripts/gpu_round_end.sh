#!/bin/bash
# Round-end evidence on the GPU box: full GPU suite, the default bench line, rocprofv3 --kernel-trace --stats of the same bench
# command, separate --pmc passes (LDS / VALU / HBM-side traffic) of the encoder, the 60-cell sweep grid.  Everything is
# written under gpurun_out/$TAG/ as it goes (no output held back behind a pipe).
set -u
export TMPDIR=/tmp
TAG=${TAG:-r06end}
STAGES=${STAGES:-ab}    # a: GPU suite, smoke;  b: PMC passes -> traffic.json, bench line, kernel trace of the same command;  g: sweep grid, decoder bench;  c: end-to-end shim
O=gpurun_out/$TAG
mkdir -p $O
if [[ $STAGES == *a* ]]; then
echo "== pytest -m gpu"; timeout -k 10 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.full.log 2>&1; grep -v amdgpu.ids $O/pytest_gpu.full.log | tail -8 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
fi
if [[ $STAGES == *t* ]]; then   # the reference's default settings (configs[3]: B = 10, S = 20) on encode_ten_kernel: PMC passes of one 1024-latent call
  LATENTS=1024 BEAMS=10 EPS1=1.0 MAXK=32 IREC_VARIANT=auto TAG=${TAG}_ten KERNEL=encode_ten scripts/gpu_pmc_variant.sh > /dev/null 2>&1
  mkdir -p $O/pmc_ten; cp gpurun_out/pmc_${TAG}_ten/*.summary $O/pmc_ten/ 2>/dev/null; cat $O/pmc_ten/*.summary | cut -c1-160
fi
if [[ $STAGES == *b* ]]; then
L=8192
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE GRBM_COUNT"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  echo "== pmc $name"
  timeout -k 10 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$name -- python3 bench.py --steps 2 --warmup 1 --latents $L --no-cpu-baseline --no-secondary > $O/pmc_$name.log 2>&1
  f=$(find $O/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 scripts/pmc_summary.py "$f" encode_team | tee $O/pmc_$name.summary
  [ -n "$f" ] && python3 scripts/pmc_summary.py "$f" decode_tensor | tee -a $O/pmc_$name.summary
done
find $O -name "*.csv" -size +2M -delete
python3 scripts/make_traffic_json.py $O $L encode_team > $O/traffic.json 2>&1 || true
# the bench line is taken AFTER the PMC passes so that it carries roofline.traffic of exactly these sources (on the box's scratch copy)
grep -q source_sha16 $O/traffic.json && cp $O/traffic.json profiles/traffic.json
echo "== bench (default line)"; timeout -k 10 900 python bench.py --steps 20 --warmup 5 2> $O/bench_default.err > $O/bench_default.out; tail -1 $O/bench_default.out > $O/bench_default.log; grep -v amdgpu.ids $O/bench_default.err | tail -12
echo "== kernel trace of the same command (no CPU baselines / secondary configs: the timed region is the same)"
rm -rf $O/trace; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" $O/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
out = [rows[0]] + [[r[0][:100]] + r[1:] for r in rows[1:]]
csv.writer(open(sys.argv[2], "w")).writerows(out)
for r in out[:8]: print(r)
PY
fi
if [[ $STAGES == *g* ]]; then   # g: the reference's 60-cell sweep grid (one oracle-checked latent per cell), decoder bench, large-block rates
echo "== sweep grid"; timeout -k 10 1000 python scripts/grid_bench.py > $O/grid.full.log 2>&1; grep -v amdgpu.ids $O/grid.full.log > $O/grid.log; tail -3 $O/grid.log
echo "== decode bench"; timeout -k 10 300 python scripts/decode_bench.py 2>&1 | grep -v amdgpu.ids > $O/decode_bench.log; cat $O/decode_bench.log
echo "== blocks of more than 1024 dims"; LATENTS=3072 SKIP_GENERIC=1 timeout -k 10 300 python scripts/generic_rate.py 2>&1 | grep -v amdgpu.ids > $O/chunk_rate.log; cat $O/chunk_rate.log
fi
if [[ $STAGES == *c* ]]; then   # c: the end-to-end shim: 38-image share and single image (eager / one HIP graph), kernel trace of the single-image pass
echo "== config 3 harness (38 images, 12 singles)"; timeout -k 10 300 python scripts/config3_harness.py --images 38 --singles 12 2>&1 | grep -v amdgpu.ids > $O/config3.log; tail -1 $O/config3.log | cut -c1-300
echo "== single image: compress / decompress eager"; timeout -k 10 300 python scripts/single_image_decompress.py 2>&1 | grep -v amdgpu.ids > $O/single_image_decompress.log; tail -1 $O/single_image_decompress.log
echo "== kernel trace of the single-image pass"
rm -rf /tmp/si_trace; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/si_trace -- python3 scripts/single_image_profile.py > $O/single_image_trace.log 2>&1
f=$(find /tmp/si_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/single_image_kernel_stats.csv && grep -h "encode_fast_kernel\|decode_tensor_kernel" $O/single_image_kernel_stats.csv | cut -c1-160
fi
du -sh $O
