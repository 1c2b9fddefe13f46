#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04w
for m in listed cost; do
  MODE=$m timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04w/trace_$m -o t -- python3 $R/scripts/midsize_trace.py > $R/gpurun_out/r04w/trace_$m.log 2>&1 || { tail -5 $R/gpurun_out/r04w/trace_$m.log; exit 1; }
  f=$(find $R/gpurun_out/r04w/trace_$m -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] || { echo "no kernel_stats.csv"; ls -R $R/gpurun_out/r04w/trace_$m; exit 1; }
  echo "== $m"; cut -c1-160 "$f" < /dev/null | sed -n 1,8p
  cp $f $R/gpurun_out/r04w/kernel_stats_$m.csv
  rm -rf $R/gpurun_out/r04w/trace_$m
done
