#!/bin/bash
# kernel trace of a mid-size call issued back to back (scripts/host_overhead.py): average duration of the preparation kernel and of the
# block kernel.  LIB = main | a build under csrc/variants/; CASE / DIMS / LISTED / KEEP as host_overhead.py takes them; OUT = directory under gpurun_out/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${OUT:-trace_mid}; rm -rf $O/trace; mkdir -p $O
REPS=${REPS:-60} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 scripts/with_lib.py ${LIB:-main} scripts/host_overhead.py > $O/run.log 2>&1
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "irec" in r["Name"]: print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:8.2f} us  min {float(r["MinNs"]) / 1e3:8.2f}')
PY
grep "back to back" $O/run.log | tail -1
rm -rf $O/trace
