#!/bin/bash
# kernel traces of the small and mid-size calls, as issued (tables built per call)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04w
run() {  # name, env...
  name=$1; shift
  env_line="$*"
  ( export AS_ISSUED=1 "$@"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04w/trace_$name -o t -- python3 $R/scripts/midsize_trace.py > $R/gpurun_out/r04w/trace_$name.log 2>&1 ) || { tail -5 $R/gpurun_out/r04w/trace_$name.log; exit 1; }
  f=$(find $R/gpurun_out/r04w/trace_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] || { echo "no kernel_stats.csv"; exit 1; }
  echo "== $name ($env_line)"; cut -c1-150 "$f" < /dev/null | sed -n 2,6p
  cp "$f" $R/gpurun_out/r04w/kernel_stats_$name.csv
  rm -rf $R/gpurun_out/r04w/trace_$name
}
run b9 LATENTS=1
run b342 LATENTS=38
run kodak2 LATENTS=1 N_DIMS=12288 BEAMS=10 EPS1=1.0
run kodak1 LATENTS=1 N_DIMS=301056 BEAMS=10 EPS1=1.0
