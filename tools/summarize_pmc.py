"""Summarises the separate `rocprofv3 --pmc` passes of tools/gpu_run.sh (`pmc:<name>`) per kernel:

    python tools/summarize_pmc.py gpurun_out/<tag> <name> [cells_per_launch]  > profiles/rNN_<name>_pmc.json

For every kernel: launches seen, the average per launch of each counter, and for kernels with both traffic counters the
HBM-side bytes per launch as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE come in KB from
separate passes; on gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane streaming reads, so the read side is doubled
(an upper bound for the 8-B-per-lane loads mixed in: round 1 calibrated x1.54 for those, `traffic_bytes_8B_calibration`);
WRITE_SIZE is exact.  SQ counters count quad-cycles; only their ratios are used.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

CAL8 = 1.5403810697802829      # 8 B per lane loads (profiles/r01_pmc_notes.md)


def main():
    out_dir, name = sys.argv[1], sys.argv[2]
    cells = float(sys.argv[3]) if len(sys.argv) > 3 else 256.0 ** 3
    tot = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(set))
    for f in glob.glob(os.path.join(out_dir, f"{name}_*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            c = row["Counter_Name"]
            tot[k][c] += float(row["Counter_Value"])
            cnt[k][c].add(row.get("Dispatch_Id"))
    res = {}
    for k in sorted(tot):
        e = {"launches": max(len(s) for s in cnt[k].values())}
        for c in sorted(tot[k]):
            e[c] = tot[k][c] / max(len(cnt[k][c]), 1)
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["traffic_bytes_per_launch"] = 2.0 * e["FETCH_SIZE"] * 1024 + e["WRITE_SIZE"] * 1024
            e["traffic_bytes_8B_calibration"] = CAL8 * e["FETCH_SIZE"] * 1024 + e["WRITE_SIZE"] * 1024
        if "SQ_INSTS_VALU" in e:
            e["SQ_INSTS_VALU_per_64_cells"] = e["SQ_INSTS_VALU"] / (cells / 64.0)
        if "SQ_ACTIVE_INST_VALU" in e and e.get("SQ_WAVE_CYCLES"):
            # a SIMD issues one VALU instruction at a time: with 4 resident waves per SIMD (k_tend4: 16 waves per CU) its VALU
            # is busy 4 x (VALU-active wave cycles / wave cycles) of the time -- the figure round 2 reported as "VALU busy"
            e["valu_busy_frac_at_4_waves_per_simd"] = 4.0 * e["SQ_ACTIVE_INST_VALU"] / e["SQ_WAVE_CYCLES"]
        if "SQ_WAIT_ANY" in e and e.get("SQ_WAVE_CYCLES"):
            e["wait_any_frac_of_wave_cycles"] = e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"]
        res[k] = e
    json.dump({"source": f"rocprofv3 --pmc (separate passes) of bench.py --steps 3 --warmup 1, {out_dir} {name}",
               "kernels": res}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
