#!/bin/bash
# round-2 GPU call Z: z-slab halo exchange overlapped with the interior levels of the next tendency launch -- slab tests, cost of the split
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2z
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests/test_parity_gpu.py tests/test_distributed_procs.py tests/test_model_contracts.py -m gpu -x -q -k "forced or slab or two_ranks or rccl or config4 or bitwise or bench_falls or checkpoint" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
run() { local nm=$1; shift
env "$@" > $O/bench_$nm.json 2> $O/bench_$nm.err || { echo "bench $nm failed"; tail -3 $O/bench_$nm.err; return 1; }
python - <<PY
import json
d=json.load(open("$O/bench_$nm.json")); p=d["phases_ms_warmup"]; print("$nm ms/step %.4f" % d["ms_per_step"], "tend", p.get("fused_tendency_step"), "halo", p.get("halo_exchange"), d["max_abs_divergence"])
PY
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20"
run slab512_ov OCNHIP_FORCE_DIST=1 OCNHIP_OVERLAP=1 $B --size 512 512 32 &&
run slab512_noov OCNHIP_FORCE_DIST=1 OCNHIP_OVERLAP=0 $B --size 512 512 32 &&
run slab256_ov OCNHIP_FORCE_DIST=1 OCNHIP_OVERLAP=1 $B --size 256 256 128 &&
run slab256_noov OCNHIP_FORCE_DIST=1 OCNHIP_OVERLAP=0 $B --size 256 256 128 &&
run slab256x64_ov OCNHIP_FORCE_DIST=1 OCNHIP_OVERLAP=1 $B --size 256 256 64 &&
run slab256x64_noov OCNHIP_FORCE_DIST=1 OCNHIP_OVERLAP=0 $B --size 256 256 64 &&
run slab512_rk3_ov OCNHIP_FORCE_DIST=1 OCNHIP_OVERLAP=1 $B --size 512 512 32 --stepper RK3 --steps 60 &&
run slab512_rk3_noov OCNHIP_FORCE_DIST=1 OCNHIP_OVERLAP=0 $B --size 512 512 32 --stepper RK3 --steps 60
OCNHIP_TRANSPORT=shm OCNHIP_BENCH_NDEV=1 timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 3 > $O/bench_2ranks_1gpu_shm.json 2> $O/bench_2ranks_1gpu_shm.err; echo "2-rank rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_2ranks_1gpu_shm.json")); print("2 ranks shm ms/step %.4f" % d["ms_per_step"], d["max_abs_divergence"], d["config"]["transport"])
PY
