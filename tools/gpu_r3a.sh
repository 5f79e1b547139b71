#!/bin/bash
# round-2 GPU call AA: overlap default by rank count; slab tests with the overlap forced on; full GPU suite
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
run() { local nm=$1; shift
env "$@" > $O/bench_$nm.json 2> $O/bench_$nm.err || { echo "bench $nm failed"; tail -3 $O/bench_$nm.err; return 1; }
python - <<PY
import json
d=json.load(open("$O/bench_$nm.json")); print("$nm ms/step %.4f" % d["ms_per_step"], d["max_abs_divergence"], d["config"].get("transport"))
PY
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20"
run slab512 OCNHIP_FORCE_DIST=1 $B --size 512 512 32 && run slab512_ov OCNHIP_FORCE_DIST=1 OCNHIP_OVERLAP=1 $B --size 512 512 32 && run dflt $B
OCNHIP_TRANSPORT=shm OCNHIP_BENCH_NDEV=1 timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 3 > $O/bench_2ranks_1gpu_shm.json 2> $O/bench_2ranks_1gpu_shm.err; echo "2-rank rc=$?"
