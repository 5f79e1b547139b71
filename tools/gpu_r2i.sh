#!/bin/bash
# round-2 GPU call I: full GPU test pass on the current tree + the two-rank rehearsal on one GPU
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2i
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
OCNHIP_TRANSPORT=shm OCNHIP_BENCH_NDEV=1 timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 3 > $O/bench_2ranks_1gpu_shm.json 2> $O/bench_2ranks_1gpu_shm.err; echo "2-rank rc=$?"
head -c 400 $O/bench_2ranks_1gpu_shm.json
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
