#!/bin/bash
# round-2 GPU call AF: final tree -- full GPU suite, smoke, default bench (with cpu_baseline), long run, 4 ranks sharing the GPU (rehearsal), traces
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3f
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
/usr/bin/time -v timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; grep -E "Elapsed" $O/bench_default.err
python - <<PY
import json
d=json.load(open("$O/bench_default.json")); print("ms/step %.4f value %.4g tend %.4f frac %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]), d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["step_roofline"])
PY
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3000 --warmup 20 > $O/bench_long.json 2> $O/bench_long.err
python - <<PY
import json
d=json.load(open("$O/bench_long.json")); print("3000 steps: ms/step %.4f div %.3g" % (d["ms_per_step"], d["max_abs_divergence"]))
PY
OCNHIP_TRANSPORT=shm OCNHIP_BENCH_NDEV=1 OCNHIP_OVERLAP=1 timeout -k 10 500 python bench.py --gpus 4 --steps 10 --warmup 2 > $O/bench_4ranks_1gpu_shm.json 2> $O/bench_4ranks_1gpu_shm.err; echo "4-rank rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_4ranks_1gpu_shm.json")); print("4 ranks shm ms/step %.4f" % d["ms_per_step"], d["max_abs_divergence"], d["config"]["decomposition"], d["config"]["local_size"])
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o trace --output-format csv -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/trace.log 2>&1
head -8 $O/trace/trace_kernel_stats.csv | cut -c1-120
