#!/bin/bash
# round-2 GPU call J: x-pass LDS image, U1 / U3 parity, transform sizes, bench lines
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2j
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_fft_sizes.py tests/test_parity_gpu.py -m gpu -x -q > $O/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $O/pytest_sel.log
tail -3 $O/pytest_sel.log
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); print("$name", "ms/step %.4f dom %s %.4f" % (d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"]), "step_frac %.3f" % d["step_roofline"]["frac_of_hbm_peak"], d["max_abs_divergence"], d["phases_ms_warmup"])
except Exception as e: print("$name ERR", e)
PY
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 100"
run default $B --steps 200
run s128 $B --size 128 128 128
run s512x32_forced OCNHIP_FORCE_DIST=1 $B --size 512 512 32
run s512x128 $B --size 512 512 128 --steps 40
run s512 $B --size 512 512 512 --steps 12 --warmup 4
