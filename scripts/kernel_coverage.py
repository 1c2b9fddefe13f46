#!/usr/bin/env python3
"""Which compiled kernels did a traced run dispatch?  Reads the kernel_stats.csv files of `rocprofv3 --kernel-trace --stats` (one per
traced process) and compares the kernel names with the __global__ functions `nm` finds in libirec_hip.so.
   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/suite_trace -- python3 -m pytest tests -m gpu -q
   python3 scripts/kernel_coverage.py gpurun_out/suite_trace > profiles/<tag>/suite_kernel_coverage.txt
Exit code 1 when a compiled kernel was never dispatched."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kernel_names as kn  # noqa: E402

calls = {}
files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)
for f in files:
    for row in csv.DictReader(open(f)):
        name = kn.normalise(row["Name"])
        calls[name] = calls.get(name, 0) + int(row["Calls"])
compiled = sorted(kn.compiled_kernels())
print(f"# kernels of libirec_hip.so dispatched by the traced run ({len(files)} traced process(es)): {sum(1 for k in compiled if calls.get(k))} of {len(compiled)}")
print("# dispatches | kernel")
for k in compiled:
    print(f"{calls.get(k, 0):10d} | {k}")
missing = [k for k in compiled if not calls.get(k)]
print(f"# never dispatched: {missing if missing else 'none'}")
sys.exit(1 if missing else 0)
