#!/bin/bash
# round-2 GPU call C: single-barrier tendency kernel (k_tend4) vs the round-1 kernel, wave-priority schemes
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2c
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "pytest rc=$?" >> $O/pytest_parity.log
tail -3 $O/pytest_parity.log
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 > $O/bench_$name.json 2> $O/bench_$name.err; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); print("$name", "ms/step %.4f tend %.4f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"]), d["max_abs_divergence"])
except Exception as e: print("$name ERR", e)
PY
}
run v4 X=1
run v4_prio1 OCNHIP_PRIO=1
run v4_prio2 OCNHIP_PRIO=2
run v4_prio3 OCNHIP_PRIO=3
run v4_nodma OCNHIP_NO_LDS_DMA=1
run v4_nodma_prio1 OCNHIP_NO_LDS_DMA=1 OCNHIP_PRIO=1
run v3 OCNHIP_TEND3=1
run v3_prio1 OCNHIP_TEND3=1 OCNHIP_PRIO=1
run v3_prio2 OCNHIP_TEND3=1 OCNHIP_PRIO=2
run v3_prio3 OCNHIP_TEND3=1 OCNHIP_PRIO=3
ls $O | head -40
