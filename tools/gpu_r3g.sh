#!/bin/bash
# round-2 GPU call AG: final tree (ABI 3, diffusivity boundary conditions, new reference tests) -- full GPU suite, smoke, default bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
SECONDS=0
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? in ${SECONDS}s"
python - <<PY
import json
d=json.load(open("$O/bench_default.json")); print("ms/step %.4f value %.4g tend %.4f frac %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]), d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["step_roofline"])
PY
timeout -k 10 300 python bench.py --no-cpu-baseline --config 3 --steps 30 --warmup 6 > $O/bench_c3.json 2> $O/bench_c3.err
python - <<PY
import json
d=json.load(open("$O/bench_c3.json")); print("c3 ms/step %.4f" % d["ms_per_step"])
PY
