#!/bin/bash
# round-2 GPU call X: size sweep and variants on the final tree (for profiles/r02_size_sweep.json)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2x
mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); print("$name", "ms/step %.4f dom %s %.4f" % (d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"]), "step_frac %.3f" % d["step_roofline"]["frac_of_hbm_peak"], d["max_abs_divergence"])
except Exception as e: print("$name ERR", e)
PY
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 100"
run s64 $B --size 64 64 64
run s128 $B --size 128 128 128
run s192 $B --size 192 192 192
run s256 $B --steps 200
run s320 $B --size 320 320 320 --steps 40
run s384 $B --size 384 384 384 --steps 30
run s512 $B --size 512 512 512 --steps 12 --warmup 4
run s512x128 $B --size 512 512 128 --steps 40
run s640x320 $B --size 640 640 320 --steps 15 --warmup 4
run s512x32_forced OCNHIP_FORCE_DIST=1 $B --size 512 512 32
run s256x512 $B --size 256 256 512 --steps 40
run rk3 $B --stepper RK3 --steps 60
run tr1 $B --tracers 1
run smooth $B --init smooth
run nu $B --nu 1e-4
