#!/bin/bash
# round-2 GPU call B: where the tendency kernel's waves wait (stamp build), barrier-removal timing experiments, the new
# GPU tests (forced slabs, config-4 slab shape, config-3 full size, contracts, two ranks on one GPU), bench lines.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2b
mkdir -p $O
cd $R
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 5 --lib clima-oceananigans.jl_amd/libocnhip_diag.so > $O/bench_diag.json 2> $O/bench_diag.err
grep diag $O/bench_diag.err
for nb in 1 2 3; do
  OCNHIP_DBG_NOBAR=$nb timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 > $O/bench_nobar$nb.json 2> $O/bench_nobar$nb.err
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout -k 10 300 python bench.py --no-cpu-baseline --init smooth > $O/bench_smooth.json 2> $O/bench_smooth.err
OCNHIP_TRANSPORT=shm OCNHIP_BENCH_NDEV=1 timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 3 > $O/bench_2ranks_1gpu_shm.json 2> $O/bench_2ranks_1gpu_shm.err
echo "2-rank rc=$?"
ls $O
