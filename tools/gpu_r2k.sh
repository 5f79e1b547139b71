#!/bin/bash
# round-2 GPU call K: tiled rest-terms kernel (k_rest4): parity, config-3 bench and trace, general-kernel block shapes
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2k
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "pytest rc=$?" >> $O/pytest_parity.log
tail -3 $O/pytest_parity.log
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); print("$name", "ms/step %.4f dom %s %.4f" % (d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"]), "step_frac %.3f" % d["step_roofline"]["frac_of_hbm_peak"], d["max_abs_divergence"], d["phases_ms_warmup"])
except Exception as e: print("$name ERR", e)
PY
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline"
run c3 $B --config 3 --steps 20 --warmup 5
run c3_norest4 OCNHIP_NO_LDS_DMA=1 $B --config 3 --steps 20 --warmup 5
run c3_amd256 OCNHIP_AMD_BLOCK=256,1,1 $B --config 3 --steps 20 --warmup 5
run c3_amd128 OCNHIP_AMD_BLOCK=128,2,1 $B --config 3 --steps 20 --warmup 5
run ppb256 $B --topology PPB --steps 50
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace_c3 -o trace --output-format csv -- python3 $R/bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > $O/trace_c3.log 2>&1
head -12 $O/trace_c3/trace_kernel_stats.csv | cut -c1-150
