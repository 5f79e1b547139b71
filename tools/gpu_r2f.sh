#!/bin/bash
# round-2 GPU call F: wave-priority code sweep (scalar s_setprio), one-chunk vs row-wise LDS DMA
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2f
mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); print("$name", "ms/step %.4f dom %s %.4f" % (d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"]), d["max_abs_divergence"])
except Exception as e: print("$name ERR", e)
PY
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 100"
run none $B
run rows OCNHIP_NO_LDS_DMA=2 $B
for c in 0xE4C6 0xD4D5 0x7955 0xC6E4 0xD515 0xECFF 0x20FF 0x9595 0x5455 0x6555; do
  run p$c OCNHIP_PRIO=$c $B
done
timeout -k 10 400 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "headline or bitwise or full_size or medium" > $O/pytest_parity.log 2>&1; echo "pytest rc=$?" >> $O/pytest_parity.log
tail -3 $O/pytest_parity.log
