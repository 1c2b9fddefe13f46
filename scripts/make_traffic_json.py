#!/usr/bin/env python3
"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE PMC summaries of one scripts/gpu_profile.sh run, tagged with the
hash of the kernel sources it was measured on (bench.py reports roofline.traffic only when that hash is the build's).
Usage: python scripts/make_traffic_json.py gpurun_out/prof_<tag> <latents_per_launch> [kernel-substring]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

d, latents = sys.argv[1], int(sys.argv[2])
pat = sys.argv[3] if len(sys.argv) > 3 else "encode_team"
vals, kernel = {}, None
for f in os.listdir(d):
    if f.endswith(".summary"):
        for line in open(os.path.join(d, f)):
            m = re.match(r"(.*?)\s+(\w+)\s+mean/dispatch\s+([\d.]+)", line)
            if m and pat in m.group(1):
                kernel = m.group(1).strip().replace("void ", "")
                vals[m.group(2)] = float(m.group(3))
fetch, write = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
total = (2 * fetch + write) * 1024
out = {"source": f"{d}: pmc FETCH_SIZE / WRITE_SIZE summaries (rocprofv3 --pmc, separate passes)", "kernel": kernel,
       "source_sha16": bench.kernel_source_hash(), "latents_per_launch": latents, "FETCH_SIZE_KiB": fetch,
       "WRITE_SIZE_KiB": write,
       "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (MI355X_MICROARCH.md: KiB units; FETCH_SIZE reads half of 16 B/lane streams on gfx950)",
       "hbm_bytes_per_launch": total, "hbm_bytes_per_latent": total / latents}
# the binding pipes (same run's SQ pass, when it is there): share of LDS-active cycles that are bank conflicts; LDS array and VALU busy
# (SQ_LDS_IDX_ACTIVE is per CU, SQ_ACTIVE_INST_VALU in quad-cycles per SIMD; GRBM_GUI_ACTIVE is summed over the 8 XCDs)
if "SQ_LDS_BANK_CONFLICT" in vals and vals.get("SQ_LDS_IDX_ACTIVE"):
    out["lds_conflict_frac"] = vals["SQ_LDS_BANK_CONFLICT"] / vals["SQ_LDS_IDX_ACTIVE"]
if "GRBM_GUI_ACTIVE" in vals and vals["GRBM_GUI_ACTIVE"] > 0:
    cyc = vals["GRBM_GUI_ACTIVE"] / 8.0
    n_cu = int(os.environ.get("N_CU", "256"))
    if "SQ_LDS_IDX_ACTIVE" in vals:
        out["lds_busy"] = vals["SQ_LDS_IDX_ACTIVE"] / (n_cu * cyc)
    if "SQ_ACTIVE_INST_VALU" in vals:
        out["valu_busy"] = 4.0 * vals["SQ_ACTIVE_INST_VALU"] / (4 * n_cu * cyc)
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
