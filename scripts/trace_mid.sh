cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r06t2; mkdir -p gpurun_out/r06t2
LATENTS=34 BEAMS=10 EPS1=1.0 IREC_VARIANT=auto REPS=40 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06t2/trace -- python3 scripts/run_variant.py > gpurun_out/r06t2/run.log 2>&1
f=$(find gpurun_out/r06t2/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"][:60], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "irec" in r["Kernel_Name"]]
# last 20 calls: prep + encode pairs
import statistics
prep = [k for k in ks if "prep_kernel" in k[0]][-20:]
enc = [k for k in ks if "encode_" in k[0]][-20:]
print("prep us", statistics.mean((e - s) / 1e3 for _, s, e in prep))
print("encode us", statistics.mean((e - s) / 1e3 for _, s, e in enc), enc[0][0])
print("gap prep end -> encode start us", statistics.mean((en[1] - pr[2]) / 1e3 for pr, en in zip(prep, enc)))
print("span prep start -> encode end us", statistics.mean((en[2] - pr[1]) / 1e3 for pr, en in zip(prep, enc)))
others = set(k[0] for k in ks)
print(others)
PY
tail -3 gpurun_out/r06t2/run.log
