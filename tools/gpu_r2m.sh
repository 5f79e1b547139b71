#!/bin/bash
# round-2 GPU call M: whole-step hipGraphs of the general path -- bitwise test, config 1 / config 3 with and without graphs
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2m
mkdir -p $O
cd $R
OCNHIP_DEBUG=1 timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "step_graph or regr_ or bitwise" > $O/pytest_graph.log 2>&1; echo "pytest rc=$?" >> $O/pytest_graph.log
tail -5 $O/pytest_graph.log
grep -c "step graphs off" $O/pytest_graph.log
run() { # name, args...
  local nm=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $O/bench_$nm.json 2> $O/bench_$nm.err || { echo "bench $nm failed"; tail -5 $O/bench_$nm.err; return 1; }
  python - <<PY
import json
d=json.load(open("$O/bench_$nm.json")); print("$nm ms/step %.4f" % d["ms_per_step"], d["step_graphs"], d["max_abs_divergence"])
PY
}
run c1_auto --config 1 --steps 2000 --warmup 20 &&
run c1_off --config 1 --steps 2000 --warmup 20 --graph off &&
run c1_256_auto --config 1 --size 256 256 1 --steps 1000 --warmup 20 &&
run c1_256_off --config 1 --size 256 256 1 --steps 1000 --warmup 20 --graph off &&
run c3_on --config 3 --steps 40 --warmup 10 --graph on &&
run c3_off --config 3 --steps 40 --warmup 10 --graph off &&
run c3_64_on --config 3 --size 64 64 64 --steps 300 --warmup 10 --graph on &&
run c3_64_off --config 3 --size 64 64 64 --steps 300 --warmup 10 --graph off &&
run ppb64_on --topology PPB --size 64 64 64 --steps 500 --warmup 10 --graph on &&
run ppb64_off --topology PPB --size 64 64 64 --steps 500 --warmup 10 --graph off
