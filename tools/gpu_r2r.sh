#!/bin/bash
# round-2 GPU call R: k_tend4 six-level unroll A/B (clean), config 3 with the pipelined hydrostatic kernel and k_amd_all occupancy variants
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2r
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "headline or medium or bitwise or ppp_weno or amd or ppb_ or regr_ or config3" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
run() { # name, env/args
  local nm=$1; shift
  env "$@" > $O/bench_$nm.json 2> $O/bench_$nm.err || { echo "bench $nm failed"; tail -5 $O/bench_$nm.err; return 1; }
  python - <<PY
import json
d=json.load(open("$O/bench_$nm.json")); p=d["phases_ms_warmup"]; print("$nm ms/step %.4f dom %.4f frac %.3f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]), "amd", p.get("amd_diffusivities"), "hydro", p.get("hydrostatic"), d["max_abs_divergence"])
PY
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20"
C="timeout -k 10 300 python bench.py --no-cpu-baseline --config 3 --steps 30 --warmup 6"
run unr $B && run nounr $B --lib tools/ab/libocnhip_nounr.so && run unr_b $B && run nounr_b $B --lib tools/ab/libocnhip_nounr.so &&
run unr_rk3 $B --stepper RK3 --steps 60 && run nounr_rk3 $B --stepper RK3 --steps 60 --lib tools/ab/libocnhip_nounr.so &&
run c3 $C && run c3_w5 OCNHIP_AMD_WAVES=5 $C && run c3_w6 OCNHIP_AMD_WAVES=6 $C && run c3_b $C
