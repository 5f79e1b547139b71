#!/bin/bash
# round-2 GPU call AB: distributed / slab tests on the final tree (overlap forced on in the two-rank cases), smoke, default bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_distributed_procs.py tests/test_parity_gpu.py tests/test_model_contracts.py -m gpu -x -q -k "two_ranks or forced or slab or rccl or config4 or bench_falls or bitwise or kernel_path" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_default.json")); print("ms/step %.4f value %.4g frac %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["frac"]), d["config"]["kernel_path"])
PY
OCNHIP_TRANSPORT=shm OCNHIP_BENCH_NDEV=1 OCNHIP_OVERLAP=1 timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 3 > $O/bench_2ranks_1gpu_shm.json 2> $O/bench_2ranks_1gpu_shm.err; echo "2-rank rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_2ranks_1gpu_shm.json")); print("2 ranks shm ms/step %.4f" % d["ms_per_step"], d["max_abs_divergence"], d["config"]["transport"], d["config"]["kernel_path"])
PY
