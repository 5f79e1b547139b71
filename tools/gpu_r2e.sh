#!/bin/bash
# round-2 GPU call E: unified kernel with scalar (wave-uniform) staging loops and priority switches
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2e
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
run prio2 $B
run prio0 OCNHIP_PRIO=0 $B
run prio1 OCNHIP_PRIO=1 $B
run prio3 OCNHIP_PRIO=3 $B
run nodma OCNHIP_NO_LDS_DMA=1 $B
run s128 $B --size 128 128 128
run s512x128 $B --size 512 512 128 --steps 40
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "pytest rc=$?" >> $O/pytest_parity.log
tail -3 $O/pytest_parity.log
