#!/bin/bash
# round-2 GPU call P: coalesced-load variant of the fused rhs + x transform (A/B), transform-size tests, reference dynamics tests, config-3 counters
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2p
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_fft_sizes.py tests/test_reference_dynamics.py tests/test_parity_gpu.py -m gpu -x -q -k "transform or dynamics or diffusion or wave or headline or full_size or medium or bitwise" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
run() { # name, env..., -- args
  local nm=$1; shift
  env "$@" > $O/bench_$nm.json 2> $O/bench_$nm.err || { echo "bench $nm failed"; tail -5 $O/bench_$nm.err; return 1; }
  python - <<PY
import json
d=json.load(open("$O/bench_$nm.json")); p=d["phases_ms_warmup"]; print("$nm ms/step %.4f" % d["ms_per_step"], "fft_forward", p.get("fft_forward"), "tend", p.get("fused_tendency_step"), d["max_abs_divergence"])
PY
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20"
run co $B &&
run team OCNHIP_XFFT_TEAM=1 $B &&
run co_128 $B --size 128 128 128 &&
run team_128 OCNHIP_XFFT_TEAM=1 $B --size 128 128 128 &&
run co_512 $B --size 512 512 128 --steps 40 &&
run team_512 OCNHIP_XFFT_TEAM=1 $B --size 512 512 128 --steps 40 &&
run co_b $B &&
run team_b OCNHIP_XFFT_TEAM=1 $B
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c -d $O/pmc3_$c -o pmc --output-format csv -- python3 $R/bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline --graph off > $O/pmc3_$c.log 2>&1
done
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/pmc3_sq -o pmc --output-format csv -- python3 $R/bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline --graph off > $O/pmc3_sq.log 2>&1
python3 - <<PY
import csv, glob, json
from collections import defaultdict
out = defaultdict(dict)
for d in ("pmc3_FETCH_SIZE", "pmc3_WRITE_SIZE", "pmc3_sq"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        tot, cnt = defaultdict(float), defaultdict(set)
        for row in csv.DictReader(open(f)):
            nm = row["Kernel_Name"].split("(")[0]
            tot[(nm, row["Counter_Name"])] += float(row["Counter_Value"])
            cnt[(nm, row["Counter_Name"])].add(row.get("Dispatch_Id"))
        for (nm, c), v in tot.items():
            out[nm][c] = v / max(len(cnt[(nm, c)]), 1)
json.dump(out, open("$O/pmc3_raw.json", "w"), indent=1)
for nm, v in out.items():
    if "FETCH_SIZE" in v: print(nm[:50], "fetchKB %.0f writeKB %.0f valu/wave %.0f waves %.0f" % (v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0), v.get("SQ_INSTS_VALU", 0) / max(v.get("SQ_WAVES", 1), 1), v.get("SQ_WAVES", 0)))
PY
