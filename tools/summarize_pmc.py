"""Summarises gpurun_out/prof_final/ (tools/collect_profiles.sh) into profiles/:
   <tag>_kernel_stats.csv (copied), <tag>_pmc_traffic_raw.json (per-kernel FETCH_SIZE / WRITE_SIZE, KB per launch),
   r01_traffic.json (what bench.py reads for roofline.traffic).

FETCH_SIZE under-reports on gfx950 by a factor that depends on the bytes per lane of the loads; it is calibrated on
kernels whose read volume is known exactly (k_zsolve256 / k_project: 16 B per lane -> x1.99; 8 B per lane -> x1.54),
as measured in round 1 (profiles/r01_pmc_notes.md).  WRITE_SIZE needs no correction.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_final")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"


def per_kernel(counter):
    f = glob.glob(os.path.join(src, f"pmc_{counter}_counter_collection.csv"))
    tot, cnt = defaultdict(float), defaultdict(int)
    seen = set()
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"].split("(")[0]
        tot[name] += float(row["Counter_Value"])
        key = (name, row.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key)
            cnt[name] += 1
    return {k: tot[k] / max(cnt[k], 1) for k in tot}


fetch, write = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
raw = {k: {"FETCH_SIZE_KB": fetch.get(k, 0.0), "WRITE_SIZE_KB": write.get(k, 0.0)} for k in sorted(set(fetch) | set(write))}
json.dump(raw, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic_raw.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "trace_kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))

CAL8 = 1.5403810697802829      # 8 B per lane loads (round-1 calibration)
tend = [k for k in raw if "k_tend_step3" in k][0]
traffic = raw[tend]["FETCH_SIZE_KB"] * 1024 * CAL8 + raw[tend]["WRITE_SIZE_KB"] * 1024
out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 3, build {tag}",
       "read_calibration": {"8B_per_lane": CAL8, "note": "see profiles/r01_pmc_notes.md"},
       "kernels": {"fused_tendency_step": {"kernel": tend, "FETCH_SIZE_KB": raw[tend]["FETCH_SIZE_KB"],
                                           "WRITE_SIZE_KB": raw[tend]["WRITE_SIZE_KB"],
                                           "traffic_bytes_per_launch": traffic,
                                           "algorithmic_bytes_per_launch": 96 * 256 ** 3}}}
json.dump(out, open(os.path.join(ROOT, "profiles", "r01_traffic.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
