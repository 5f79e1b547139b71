#!/bin/bash
# round-2 GPU call D: unified single-barrier kernel (x-tiled variant, row-wise LDS DMA): all GPU tests, priorities, sizes
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2d
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); print("$name", "ms/step %.4f dom %s %.4f" % (d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"]), "step_frac %.3f" % d["step_roofline"]["frac_of_hbm_peak"], d["max_abs_divergence"], d["phases_ms_warmup"])
except Exception as e: print("$name ERR", e)
PY
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 100"
run prio2 $B
run prio0 OCNHIP_PRIO=0 $B
run prio4 OCNHIP_PRIO=4 $B
run nodma OCNHIP_NO_LDS_DMA=1 $B
run s128 $B --size 128 128 128
run s192 $B --size 192 192 192
run s512x128 $B --size 512 512 128 --steps 40
run s512x32_forced OCNHIP_FORCE_DIST=1 $B --size 512 512 32
run rk3 $B --stepper RK3 --steps 40
run tr1 $B --tracers 1
run c3 $B --config 3 --steps 20 --warmup 5
run c1 $B --config 1
ls $O | head -50
