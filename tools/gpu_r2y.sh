#!/bin/bash
# round-2 GPU call Y: kernel trace of one config-4 slab (512x512x32 through the slab code path on one GPU) and of 256x256x128 slabs
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2y
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
OCNHIP_FORCE_DIST=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace_slab -o trace --output-format csv -- python3 $R/bench.py --size 512 512 32 --steps 200 --warmup 20 --no-cpu-baseline > $O/trace_slab.json 2> $O/trace_slab.err
python3 - <<PY
import csv, json
rows=list(csv.DictReader(open("$O/trace_slab/trace_kernel_stats.csv")))
steps=280
for r in rows[:22]:
    print(r["Name"][:70].ljust(72), r["Calls"].rjust(6), "%8.1f us  per step %7.1f" % (float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/steps/1e3))
print(open("$O/trace_slab.json").read()[-600:])
PY
OCNHIP_FORCE_DIST=1 timeout -k 10 300 python3 $R/bench.py --size 256 256 128 --steps 200 --warmup 20 --no-cpu-baseline > $O/slab256x128.json 2> $O/slab256x128.err
timeout -k 10 300 python3 $R/bench.py --size 256 256 128 --steps 200 --warmup 20 --no-cpu-baseline > $O/plain256x128.json 2> $O/plain256x128.err
timeout -k 10 300 python3 $R/bench.py --size 512 512 32 --steps 200 --warmup 20 --no-cpu-baseline > $O/plain512x32.json 2> $O/plain512x32.err
python3 - <<PY
import json
for nm in ("slab256x128","plain256x128","plain512x32"):
    d=json.load(open("$O/%s.json" % nm)); print(nm, "%.4f" % d["ms_per_step"], d["phases_ms_warmup"])
PY
