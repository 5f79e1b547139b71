#!/bin/bash
# round-2 GPU call W: full GPU suite on the final tree, smoke, default bench line, bench lines of configs 1 and 3, kernel trace of the default run
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2w
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_default.json")); print("ms/step %.4f value %.4g tend %.4f frac %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]), d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["step_roofline"])
PY
timeout -k 10 300 python bench.py --no-cpu-baseline --config 1 --steps 2000 --warmup 20 > $O/bench_c1.json 2> $O/bench_c1.err
timeout -k 10 300 python bench.py --no-cpu-baseline --config 3 --steps 30 --warmup 6 > $O/bench_c3.json 2> $O/bench_c3.err
timeout -k 10 300 python bench.py --no-cpu-baseline --topology PPB --steps 100 --warmup 10 > $O/bench_ppb256.json 2> $O/bench_ppb256.err
python - <<PY
import json
for nm in ("c1", "c3", "ppb256"):
    d=json.load(open("$O/bench_%s.json" % nm)); print(nm, "ms/step %.4f" % d["ms_per_step"], d["step_graphs"], d["step_roofline"]["frac_of_hbm_peak"])
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o trace --output-format csv -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/trace.log 2>&1
head -9 $O/trace/trace_kernel_stats.csv | cut -c1-150
