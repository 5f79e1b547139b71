#!/bin/bash
# round-2 GPU call Q: six-level unroll of k_tend4 with circular register windows -- A/B against the same sources without it
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2q
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "headline or medium or bitwise or full_size or ppp_weno" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
run() { # name, args
  local nm=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 "$@" > $O/bench_$nm.json 2> $O/bench_$nm.err || { echo "bench $nm failed"; tail -5 $O/bench_$nm.err; return 1; }
  python - <<PY
import json
d=json.load(open("$O/bench_$nm.json")); print("$nm ms/step %.4f tend %.4f frac %.3f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]), d["max_abs_divergence"])
PY
}
run unr && run nounr --lib tools/ab/libocnhip_nounr.so && run unr_b && run nounr_b --lib tools/ab/libocnhip_nounr.so &&
run unr_128 --size 128 128 128 && run nounr_128 --size 128 128 128 --lib tools/ab/libocnhip_nounr.so &&
run unr_rk3 --stepper RK3 --steps 60 && run nounr_rk3 --stepper RK3 --steps 60 --lib tools/ab/libocnhip_nounr.so
