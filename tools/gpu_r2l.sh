#!/bin/bash
# round-2 GPU call L: full GPU test pass on the current tree (k_rest4, hydrostatic batching, U1 / U3, RCCL fallback), smoke, default bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2l
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
timeout -k 10 300 python bench.py --no-cpu-baseline --config 3 --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err
python - <<PY
import json
d=json.load(open("$O/bench_c3.json")); print("c3 ms/step %.4f" % d["ms_per_step"], d["phases_ms_warmup"])
PY
